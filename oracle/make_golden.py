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


if __name__ == "__main__":
    main()
