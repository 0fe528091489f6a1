"""Generate tests/golden/* from the REAL reference (build container only: needs /root/reference).

    python -m oracle.make_golden            # everything (spawns a second process for inductor)
    python -m oracle.make_golden sched      # scheduler traces only

Writes
  tests/golden/sched_<scenario>.json.gz   step-by-step traces of the reference's Scheduler /
                                          BlockManager / Sequence on the seeded workloads of
                                          oracle/host_trace.py
  tests/golden/ops_eager.safetensors      inputs + outputs of the reference's RMSNorm,
  tests/golden/ops_compiled.safetensors   RotaryEmbedding, SiluAndMul modules run eagerly
                                          (TORCHDYNAMO_DISABLE=1) and as shipped (@torch.compile,
                                          inductor CPU backend)
  tests/golden/model_tiny.safetensors     logits of the reference's Qwen3ForCausalLM on the
                                          seeded qwen3-tiny checkpoint (prefill + 3 decode steps
                                          through the reference's own Attention/Context, eager),
                                          with the inputs needed to replay them
  tests/golden/sampler_eager.safetensors  the reference's Sampler module (layers/sampler.py:5-12, eager) on seeded
                                          logits / temperatures with torch.manual_seed(SAMPLER_SEED): the sampled ids
  tests/golden/engine_tiny.json.gz        a multi-step ENGINE run of the reference: its Scheduler + BlockManager +
                                          Sequence + Qwen3ForCausalLM (eager) driven by a CPU restatement of
                                          ModelRunner.prepare_prefill / prepare_decode / run (model_runner.py:123-220
                                          is hard-wired to CUDA) over a workload with chunked prefill, a shared
                                          512-token prefix served from the prefix cache and preemption by recompute:
                                          per step the batch, block tables, greedy token ids and top-1/top-2 margins
The reference ships no tests or golden vectors of its own; these files pin the
oracle (and through it the HIP kernels) to the reference's actual outputs.
"""
from __future__ import annotations

import gzip
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
BF16 = torch.bfloat16


def gen_sched():
    from . import host_trace, ref_import
    mods = ref_import.load_reference()
    Sched = mods["nanovllm.engine.scheduler"].Scheduler
    Seq = mods["nanovllm.engine.sequence"].Sequence
    SP = mods["nanovllm.sampling_params"].SamplingParams
    Seq.block_size = 256
    for name in host_trace.SCENARIOS:
        trace = host_trace.run_trace(name, lambda cfg: Sched(cfg),
                                     lambda p, mt, ie: Seq(p, SP(temperature=1.0, max_tokens=mt, ignore_eos=ie)))
        with gzip.open(os.path.join(GOLDEN, f"sched_{name}.json.gz"), "wt") as fh:
            json.dump(trace, fh, separators=(",", ":"))
        print(f"sched_{name}: {len(trace) - 1} steps")


def op_inputs():
    g = torch.Generator().manual_seed(1234)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(BF16)
    return {
        "rms_x": r(19, 1024, scale=3.0), "rms_w": (1 + 0.1 * torch.randn(1024, generator=g)).to(BF16),
        "head_x": r(11, 16, 128, scale=2.0), "head_w": (1 + 0.1 * torch.randn(128, generator=g)).to(BF16),
        "add_x": r(19, 1024), "add_r": r(19, 1024, scale=2.0),
        "rope_q": r(13, 16, 128), "rope_k": r(13, 8, 128),
        "rope_pos": torch.randint(0, 4096, (13,), generator=g),
        "silu_x": r(17, 2 * 3072, scale=2.0),
    }


def gen_ops(tag: str):
    """tag = 'eager' (TORCHDYNAMO_DISABLE=1 must be set by the caller) or 'compiled'."""
    from safetensors.torch import save_file
    from . import ref_import
    mods = ref_import.load_reference(eager=(tag == "eager"))
    RMSNorm = mods["nanovllm.layers.layernorm"].RMSNorm
    Rotary = mods["nanovllm.layers.rotary_embedding"].RotaryEmbedding
    Silu = mods["nanovllm.layers.activation"].SiluAndMul
    x = op_inputs()
    out = dict(x)
    torch.set_default_dtype(BF16)
    try:
        n = RMSNorm(1024, 1e-6)
        n.weight.data.copy_(x["rms_w"])
        out["rms_y"] = n(x["rms_x"].clone())
        y, res = n(x["add_x"].clone(), x["add_r"].clone())
        out["add_y"], out["add_res"] = y, res
        hn = RMSNorm(128, 1e-6)
        hn.weight.data.copy_(x["head_w"])
        out["head_y"] = hn(x["head_x"].clone())
        rope = Rotary(128, 128, 4096, 1e6)
        q, k = rope(x["rope_pos"], x["rope_q"].clone(), x["rope_k"].clone())
        out["rope_qo"], out["rope_ko"] = q, k
        out["silu_y"] = Silu()(x["silu_x"].clone())
    finally:
        torch.set_default_dtype(torch.float32)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, f"ops_{tag}.safetensors"))
    print(f"ops_{tag}: {len(out)} tensors")


def gen_model_tiny():
    """Reference Qwen3ForCausalLM (eager) on the qwen3-tiny synthetic checkpoint: one prefill of two
    ragged prompts then 3 greedy decode steps, through the reference's Attention + Context with the
    reference cache layout; K/V store by oracle.ops.store_kvcache (the Triton launcher needs a GPU)."""
    import tempfile
    from safetensors.torch import save_file
    from transformers import AutoConfig
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    from . import ops, ref_import
    mods = ref_import.load_reference(eager=True)
    attn_mod = mods["nanovllm.layers.attention"]
    attn_mod.store_kvcache = ops.store_kvcache          # CPU stand-in for the Triton launcher (same semantics)
    ctx = mods["nanovllm.utils.context"]
    path = tempfile.mkdtemp(prefix="qwen3tiny_")
    write_synthetic_checkpoint(path, "qwen3-tiny", seed=0, vocab_size=512, max_position_embeddings=2048)
    hf = AutoConfig.from_pretrained(path)
    torch.set_default_dtype(BF16)
    try:
        model = mods["nanovllm.models.qwen3"].Qwen3ForCausalLM(hf)
        mods["nanovllm.utils.loader"].load_model(model, path)
    finally:
        torch.set_default_dtype(torch.float32)
    B, nblk = 256, 4
    L, hkv = hf.num_hidden_layers, hf.num_key_value_heads
    kv = torch.zeros(2, L, nblk, B, hkv, 128, dtype=BF16)
    li = 0
    for m in model.modules():
        if hasattr(m, "k_cache") and hasattr(m, "v_cache"):
            m.k_cache, m.v_cache = kv[0, li], kv[1, li]
            li += 1
    g = torch.Generator().manual_seed(7)
    prompts = [torch.randint(0, 512, (37,), generator=g).tolist(), torch.randint(0, 512, (300,), generator=g).tolist()]
    tables = [[2], [0, 3]]
    out = {}
    with torch.inference_mode():
        ids = torch.tensor(prompts[0] + prompts[1])
        pos = torch.tensor(list(range(37)) + list(range(300)))
        slots = torch.tensor([2 * B + t for t in range(37)] + [0 * B + t for t in range(256)] +
                             [3 * B + t for t in range(44)], dtype=torch.int32)
        cu = torch.tensor([0, 37, 337], dtype=torch.int32)
        ctx.set_context(True, cu, cu, 300, 300, slots, None, None)
        logits = model.compute_logits(model(ids, pos))
        ctx.reset_context()
        out["prefill_ids"], out["prefill_pos"], out["prefill_slots"], out["prefill_cu"] = ids, pos, slots, cu
        out["prefill_logits"] = logits
        toks = [list(p) for p in prompts]
        nxt = logits.float().argmax(-1).tolist()
        for step in range(3):
            for s, t in zip(toks, nxt):
                s.append(t)
            lens = [len(s) for s in toks]
            ids = torch.tensor([s[-1] for s in toks])
            pos = torch.tensor([n - 1 for n in lens])
            slots = torch.tensor([tables[i][(n - 1) // B] * B + (n - 1) % B for i, n in enumerate(lens)],
                                 dtype=torch.int32)
            bt = torch.tensor([[2, -1], [0, 3]], dtype=torch.int32)
            ctx.set_context(False, slot_mapping=slots, context_lens=torch.tensor(lens, dtype=torch.int32),
                            block_tables=bt)
            logits = model.compute_logits(model(ids, pos))
            ctx.reset_context()
            out[f"decode{step}_ids"], out[f"decode{step}_logits"] = ids, logits
            nxt = logits.float().argmax(-1).tolist()
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "model_tiny.safetensors"))
    print("model_tiny:", {k: tuple(v.shape) for k, v in out.items() if "logits" in k})


SAMPLER_SEED = 4321


def sampler_inputs():
    g = torch.Generator().manual_seed(99)
    logits = (torch.randn(23, 3000, generator=g) * 2.5).to(BF16)
    logits[5, 77] = 40.0                                   # a row whose winner is certain at any temperature here
    temps = torch.tensor([1.0, 0.6, 0.3, 2.0, 0.05, 1.0, 0.8, 1.3] * 3, dtype=torch.float32)[:23]
    return logits, temps


def gen_sampler():
    """Reference Sampler.forward (layers/sampler.py:8-12), eager, global CPU generator seeded with SAMPLER_SEED.
    (Under inductor the module consumes the generator differently — SURVEY.md §8c(4) — so only the eager module is a
    stream-exact pin; three consecutive calls are recorded so that generator state carried across calls is pinned too.)"""
    from safetensors.torch import save_file
    from . import ref_import
    mods = ref_import.load_reference(eager=True)
    sampler = mods["nanovllm.layers.sampler"].Sampler()
    logits, temps = sampler_inputs()
    torch.manual_seed(SAMPLER_SEED)
    out = {"logits": logits, "temps": temps}
    with torch.inference_mode():
        for call in range(3):
            out[f"tokens{call}"] = sampler(logits.clone(), temps.clone())
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "sampler_eager.safetensors"))
    print("sampler_eager:", [out[f"tokens{c}"][:6].tolist() for c in range(3)])


# ---- engine golden -----------------------------------------------------------------------------------------------
ENGINE_CFG = dict(num_kvcache_blocks=10, max_num_seqs=4, max_num_batched_tokens=640, kvcache_block_size=256, eos=-1)


def engine_workload():
    """Shared 512-token prefix + ragged suffixes, one long unrelated prompt; sized (tools: a dry run of the
    scheduler) so that the run contains chunked prefill, prefix-cache hits and a preemption."""
    g = torch.Generator().manual_seed(11)
    shared = torch.randint(0, 512, (512,), generator=g).tolist()
    prompts = [shared + torch.randint(0, 512, (int(n),), generator=g).tolist() for n in (30, 200, 250, 5)]
    prompts.append(torch.randint(0, 512, (900,), generator=g).tolist())
    prompts.append(shared + torch.randint(0, 512, (100,), generator=g).tolist())
    return prompts, [20, 12, 30, 25, 9, 16]


def gen_engine_tiny():
    """The reference's own engine loop (llm_engine.py:49-55: schedule -> run -> postprocess) on CPU. Everything is the
    imported reference except (a) the runner's batch preparation, restated below from model_runner.py:123-188 without
    the CUDA / pinned-memory calls, (b) store_kvcache (Triton launcher) and flash-attn, replaced by oracle.ops as in
    gen_model_tiny, (c) sampling: T = 0 is outside the reference's SamplingParams (sampling_params.py:11), so the
    token is the argmax of the logits (first maximal index) — the T -> 0 limit of sampler.py:8-12."""
    import tempfile
    from types import SimpleNamespace
    from transformers import AutoConfig
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    from . import ops, ref_import
    mods = ref_import.load_reference(eager=True)
    mods["nanovllm.layers.attention"].store_kvcache = ops.store_kvcache
    ctx = mods["nanovllm.utils.context"]
    Sched = mods["nanovllm.engine.scheduler"].Scheduler
    Seq = mods["nanovllm.engine.sequence"].Sequence
    SP = mods["nanovllm.sampling_params"].SamplingParams
    path = tempfile.mkdtemp(prefix="qwen3tiny_")
    write_synthetic_checkpoint(path, "qwen3-tiny", seed=0, vocab_size=512, max_position_embeddings=2048)
    hf = AutoConfig.from_pretrained(path)
    torch.set_default_dtype(BF16)
    try:
        model = mods["nanovllm.models.qwen3"].Qwen3ForCausalLM(hf)
        mods["nanovllm.utils.loader"].load_model(model, path)
    finally:
        torch.set_default_dtype(torch.float32)
    cfg = SimpleNamespace(**ENGINE_CFG)
    B, nblk = cfg.kvcache_block_size, cfg.num_kvcache_blocks
    Seq.block_size = B
    kv = torch.zeros(2, hf.num_hidden_layers, nblk, B, hf.num_key_value_heads, 128, dtype=BF16)   # model_runner.py:115
    li = 0
    for m in model.modules():
        if hasattr(m, "k_cache") and hasattr(m, "v_cache"):
            m.k_cache, m.v_cache = kv[0, li], kv[1, li]
            li += 1

    def tables_of(seqs):                                   # model_runner.py:123-127
        width = max(len(s.block_table) for s in seqs)
        return torch.tensor([s.block_table + [-1] * (width - len(s.block_table)) for s in seqs], dtype=torch.int32)

    def stage_prefill(seqs):                               # model_runner.py:129-170
        ids, pos, slots, cq, ck = [], [], [], [0], [0]
        mq = mk = 0
        for s in seqs:
            lo = s.num_cached_tokens
            hi = lo + s.num_scheduled_tokens
            ids += s[lo:hi]
            pos += range(lo, hi)
            cq.append(cq[-1] + hi - lo)
            ck.append(ck[-1] + hi)
            mq, mk = max(mq, hi - lo), max(mk, hi)
            for blk in range(lo // B, (hi + B - 1) // B):   # :151-161, one block at a time
                first = s.block_table[blk] * B
                a = first + (lo % B if blk == lo // B else 0)
                b = first + (B if blk != (hi + B - 1) // B - 1 else hi - blk * B)
                slots += range(a, b)
        bt = tables_of(seqs) if ck[-1] > cq[-1] else None   # :162-163
        ctx.set_context(True, torch.tensor(cq, dtype=torch.int32), torch.tensor(ck, dtype=torch.int32), mq, mk,
                        torch.tensor(slots, dtype=torch.int32), None, bt)
        return torch.tensor(ids, dtype=torch.int64), torch.tensor(pos, dtype=torch.int64)

    def stage_decode(seqs):                                # model_runner.py:172-188
        ids = torch.tensor([s.last_token for s in seqs], dtype=torch.int64)
        pos = torch.tensor([len(s) - 1 for s in seqs], dtype=torch.int64)
        slots = torch.tensor([s.block_table[-1] * B + s.last_block_num_tokens - 1 for s in seqs], dtype=torch.int32)
        lens = torch.tensor([len(s) for s in seqs], dtype=torch.int32)
        ctx.set_context(False, slot_mapping=slots, context_lens=lens, block_tables=tables_of(seqs))
        return ids, pos

    sched = Sched(cfg)
    prompts, max_tokens = engine_workload()
    seqs_all = [Seq(p, SP(temperature=1.0, max_tokens=m, ignore_eos=True)) for p, m in zip(prompts, max_tokens)]
    index_of = {id(s): i for i, s in enumerate(seqs_all)}
    for s in seqs_all:
        sched.add(s)
    trace = []
    chunked = hits = preempted = False
    prefilled = set()
    with torch.inference_mode():
        while not sched.is_finished():
            batch, is_prefill = sched.schedule()                                        # llm_engine.py:50
            rec = dict(prefill=bool(is_prefill), seqs=[index_of[id(s)] for s in batch],
                       sched=[s.num_scheduled_tokens for s in batch], cached=[s.num_cached_tokens for s in batch],
                       tables=[list(s.block_table) for s in batch])
            if is_prefill:
                for s in batch:
                    i = index_of[id(s)]
                    done = s.num_cached_tokens + s.num_scheduled_tokens == len(s)
                    chunked |= not done
                    hits |= s.num_cached_tokens > 0 and s.num_cached_tokens % B == 0 and i not in prefilled and \
                        s.num_cached_tokens + s.num_scheduled_tokens == len(s) and s.num_scheduled_tokens < len(s)
                    preempted |= i in prefilled
                    if done:
                        prefilled.add(i)
            ids, pos = stage_prefill(batch) if is_prefill else stage_decode(batch)       # model_runner.py:214
            logits = model.compute_logits(model(ids, pos))                               # :216 (run_model, eager branch)
            ctx.reset_context()
            lf = logits.float()
            tokens = lf.argmax(-1).tolist()
            top2 = lf.topk(2, -1).values
            rec["tokens"] = tokens
            rec["margin"] = [round(float(x), 6) for x in (top2[:, 0] - top2[:, 1])]
            rec["logit_sum"] = [round(float(x), 4) for x in lf.sum(-1)]
            trace.append(rec)
            sched.postprocess(batch, tokens, is_prefill)                                 # llm_engine.py:53
    assert chunked and hits and preempted, (chunked, hits, preempted)
    trace.append(dict(final=[list(s.completion_token_ids) for s in seqs_all]))
    with gzip.open(os.path.join(GOLDEN, "engine_tiny.json.gz"), "wt") as fh:
        json.dump(trace, fh, separators=(",", ":"))
    print(f"engine_tiny: {len(trace) - 1} steps, chunked prefill / prefix hits / preemption all present")


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "ops_compiled":
        gen_ops("compiled")
        return
    if what in ("all", "sched"):
        gen_sched()
    if what in ("all", "ops"):
        os.environ["TORCHDYNAMO_DISABLE"] = "1"
        gen_ops("eager")
        env = dict(os.environ)
        env.pop("TORCHDYNAMO_DISABLE", None)
        subprocess.run([sys.executable, "-m", "oracle.make_golden", "ops_compiled"], check=True, env=env,
                       cwd=os.path.dirname(HERE))
    if what in ("all", "model"):
        gen_model_tiny()
    if what in ("all", "sampler"):
        os.environ["TORCHDYNAMO_DISABLE"] = "1"
        gen_sampler()
    if what in ("all", "engine"):
        os.environ["TORCHDYNAMO_DISABLE"] = "1"
        gen_engine_tiny()


if __name__ == "__main__":
    main()
