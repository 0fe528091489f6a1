"""End-to-end parity on the MI355X: our engine (HIP kernels, hipGraph decode) vs the CPU oracle
engine on the same synthetic checkpoint and prompts.

Parity definition: greedy (T=0). bf16 noise makes free-running comparison of
random-weight models meaningless after the first near-tie, so the oracle is TEACHER-FORCED with
our tokens and every one of our decisions is judged against the oracle's logits for the same
history: the chosen token must be the oracle's argmax, or lie within `TOL` of the oracle's max
logit (TOL ~ 2x the measured reference-eager-vs-reference-compiled logits floor); and the large
majority of steps must be exact argmax matches. Scheduling (batch composition, block tables) must
be identical step for step.
"""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 0.15


def _run_ours(path, prompts, max_tokens, **kw):
    from nano_vllm_amd import LLM, SamplingParams
    llm = LLM(path, **kw)
    rec = []
    runner = llm.model_runner
    orig = runner.call

    def snap(seqs, is_prefill):
        return dict(prefill=is_prefill, seq_ids=[s.seq_id for s in seqs], tables=[list(s.block_table) for s in seqs],
                    sched=[s.num_scheduled_tokens for s in seqs])

    open_step = []

    def spy(method, *args):
        # a step is either one "run" call or a "decode_begin" ... "decode_end" pair (decode lookahead);
        # batch composition and block tables are snapshotted when the step is issued
        if method == "decode_begin":
            open_step.append(snap(args[0], False))
        elif method == "run":
            open_step.append(snap(*args))
        out = orig(method, *args)
        if method in ("run", "decode_end"):
            step = open_step.pop(0)          # FIFO: with the lookahead, step N+1 begins before step N ends
            step["tokens"] = list(out)
            rec.append(step)
        return out

    runner.call = spy
    sps = [SamplingParams(temperature=0.0, max_tokens=m, ignore_eos=True) for m in max_tokens]
    outs = llm.generate(prompts, sps, use_tqdm=False)
    nblk = llm.config.num_kvcache_blocks
    runner.call = orig
    llm.exit()
    return outs, rec, nblk


def _judge(path, prompts, max_tokens, rec, nblk, **sched_kw):
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3, load_weights
    cfg, w = load_weights(path)
    eng = OracleEngine(OracleQwen3(cfg, w, compiled=True), nblk, 256, **sched_kw)
    eng.keep_logits = True
    for p, m in zip(prompts, max_tokens):
        eng.add(p, 0.0, m, True)
    exact = total = 0
    worst = 0.0
    base = None
    for i, r in enumerate(rec):
        eng.step(forced_tokens=r["tokens"])
        o = eng.trace[-1]
        if base is None:
            base = r["seq_ids"][0] - o["seq_ids"][0]
        assert o["is_prefill"] == r["prefill"], f"step {i}: phase differs"
        assert [s + base for s in o["seq_ids"]] == r["seq_ids"], f"step {i}: batch composition differs"
        assert o["tables"] == r["tables"], f"step {i}: block tables differ"
        logits = o["logits"]
        for row, tok in enumerate(r["tokens"]):
            if r["prefill"] and o["sched"][row] + o["cached"][row] < 0:
                continue
            gap = float(logits[row].max() - logits[row, tok])
            worst = max(worst, gap)
            exact += gap == 0.0
            total += 1
    assert not eng.waiting and not eng.running
    return exact, total, worst


@pytest.fixture(scope="module")
def tiny_ckpt():
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    path = tempfile.mkdtemp(prefix="qwen3tiny_")
    write_synthetic_checkpoint(path, "qwen3-tiny", seed=0, vocab_size=512, max_position_embeddings=2048)
    return path


def _prompts(n, lo, hi, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, vocab, (int(torch.randint(lo, hi, (1,), generator=g)),), generator=g).tolist()
            for _ in range(n)]


@pytest.mark.parametrize("eager", [True, False])
def test_tiny_model_greedy_parity(tiny_ckpt, eager):
    prompts = _prompts(6, 5, 600, 512, seed=3)
    max_tokens = [24, 40, 8, 33, 1, 17]
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, enforce_eager=eager, max_model_len=2048,
                                num_kvcache_blocks=32, max_num_seqs=16)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    exact, total, worst = _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, max_num_seqs=16)
    print(f"tiny eager={eager}: {exact}/{total} exact argmax, worst logit gap {worst:.4f}")
    assert worst <= TOL and exact >= 0.9 * total


@pytest.mark.parametrize("name", ["qwen3-tiny-untied", "qwen3-tiny-g8"])
def test_other_head_geometries_greedy_parity(name):
    """GQA group sizes 4 (Qwen3-8B-like, untied lm_head) and 8 with a single kv head (Qwen3-32B at TP=8):
    the G=4 / G=8 instantiations of the fused decode kernel and of the MFMA prefill kernel, end to end."""
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    path = tempfile.mkdtemp(prefix=name.replace("-", "_") + "_")
    write_synthetic_checkpoint(path, name, seed=1, vocab_size=512, max_position_embeddings=2048)
    prompts = _prompts(5, 5, 700, 512, seed=9)
    max_tokens = [12, 30, 5, 21, 9]
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, enforce_eager=False, max_model_len=2048,
                                num_kvcache_blocks=24, max_num_seqs=8)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    exact, total, worst = _judge(path, prompts, max_tokens, rec, nblk, max_num_seqs=8)
    print(f"{name}: {exact}/{total} exact argmax, worst logit gap {worst:.4f}")
    assert worst <= TOL and exact >= 0.9 * total


def test_tiny_model_chunked_prefill_and_preemption(tiny_ckpt):
    """Small token budget + small block pool: chunked prefill (paged-prefix attention path),
    prefix-cache reuse of a shared 512-token prefix, and preemption by recompute."""
    g = torch.Generator().manual_seed(11)
    shared = torch.randint(0, 512, (512,), generator=g).tolist()
    prompts = [shared + torch.randint(0, 512, (int(n),), generator=g).tolist() for n in (30, 200, 77, 5)]
    prompts.append(torch.randint(0, 512, (900,), generator=g).tolist())
    max_tokens = [20, 20, 20, 20, 20]
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, enforce_eager=True, max_model_len=2048,
                                num_kvcache_blocks=9, max_num_seqs=8, max_num_batched_tokens=640)
    assert any(r["prefill"] and len(r["seq_ids"]) == 1 for r in rec)
    exact, total, worst = _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, max_num_seqs=8,
                                 max_num_batched_tokens=640)
    print(f"tiny chunked/preempt: {exact}/{total} exact argmax, worst logit gap {worst:.4f}")
    assert worst <= TOL and exact >= 0.9 * total


def test_sampling_temperature_runs_and_is_seeded(tiny_ckpt):
    from nano_vllm_amd import LLM, SamplingParams
    prompts = _prompts(4, 5, 100, 512, seed=5)
    res = []
    for _ in range(2):
        llm = LLM(tiny_ckpt, enforce_eager=True, max_model_len=1024, num_kvcache_blocks=16, seed=123)
        outs = llm.generate(prompts, SamplingParams(temperature=0.8, max_tokens=12, ignore_eos=True), use_tqdm=False)
        res.append([o["token_ids"] for o in outs])
        llm.exit()
    assert res[0] == res[1]                       # same seed => same draw
    assert all(len(t) == 12 for t in res[0])


def test_string_prompts_and_eos(tiny_ckpt):
    from nano_vllm_amd import LLM, SamplingParams
    llm = LLM(tiny_ckpt, enforce_eager=True, max_model_len=1024, num_kvcache_blocks=16)
    outs = llm.generate(["introduce yourself", "list all prime numbers within 100"],
                        SamplingParams(temperature=0.6, max_tokens=16), use_tqdm=False)
    assert len(outs) == 2 and all(isinstance(o["text"], str) and 1 <= len(o["token_ids"]) <= 16 for o in outs)
    eos = llm.tokenizer.eos_token_id
    for o in outs:
        assert eos not in o["token_ids"][:-1]       # generation stops at the first EOS (included)
    llm.exit()


# ---------------------------------------------------------------------------------------------
# BASELINE.json config 2 shapes (Qwen3-0.6B: hidden 1024, 28 layers, 16/8 heads, vocab 151,936) with
# seeded synthetic weights. The CPU oracle runs this size at ~4 tok/s, so the oracle comparison uses a
# small sample and the full bench-shaped workload is covered by size-independent properties.
@pytest.fixture(scope="module")
def ckpt_06b():
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    path = tempfile.mkdtemp(prefix="qwen3_06b_")
    write_synthetic_checkpoint(path, "qwen3-0.6b", with_weights=False)
    return path


def _oracle_weights_06b(device):
    from nano_vllm_amd.weights import parameter_shapes, qwen3_config_dict, synth_tensor
    cfg = qwen3_config_dict("qwen3-0.6b")
    return cfg, {n: synth_tensor(n, s, 0, device=device).cpu() for n, s in parameter_shapes(cfg).items()}


@pytest.mark.parametrize("fused_lm_head", [False, True])
def test_qwen3_06b_shape_greedy_parity_vs_oracle(ckpt_06b, fused_lm_head, monkeypatch):
    """(fused_lm_head: the opt-in nvl_lmhead_sample path — sampling inside the lm_head GEMM's epilogue.)
    Full-size layers (fused decode attention G=2, skinny decode GEMMs at K=1024/2048/3072, sampler
    over 151,936 logits): every token we pick must be the CPU oracle's argmax (or within TOL) for the
    same history, with identical scheduling."""
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3
    import nano_vllm_amd.layers as layers_mod
    monkeypatch.setattr(layers_mod, "_FUSED_LMHEAD", fused_lm_head)
    prompts = _prompts(3, 20, 300, 10000, seed=21)
    max_tokens = [24, 22, 26] if not fused_lm_head else [8, 6, 7]
    outs, rec, nblk = _run_ours(ckpt_06b, prompts, max_tokens, enforce_eager=False, max_model_len=1024,
                                num_kvcache_blocks=16, max_num_seqs=8, dummy_weights=True)
    cfg, w = _oracle_weights_06b("cuda")
    eng = OracleEngine(OracleQwen3(cfg, w, compiled=True), nblk, 256, max_num_seqs=8)
    eng.keep_logits = True
    for p, m in zip(prompts, max_tokens):
        eng.add(p, 0.0, m, True)
    exact = total = 0
    worst = 0.0
    for i, r in enumerate(rec):
        eng.step(forced_tokens=r["tokens"])
        o = eng.trace[-1]
        assert o["is_prefill"] == r["prefill"] and o["tables"] == r["tables"], f"step {i}: schedule differs"
        for row, tok in enumerate(r["tokens"]):
            gap = float(o["logits"][row].max() - o["logits"][row, tok])
            worst = max(worst, gap)
            exact += gap == 0.0
            total += 1
    print(f"0.6B shapes: {exact}/{total} exact argmax, worst logit gap {worst:.4f}")
    assert worst <= TOL and exact >= 0.8 * total


def test_config1_example_prompts_eager_exact_tokens(ckpt_06b):
    """BASELINE.json config 1: 0.6B shapes, enforce_eager, the reference example's two prompt strings
    (example.py:12-15) + two token-id lists, max_tokens 16, T = 0 => every token is the oracle's argmax for the same
    history. (The reference runs this case on a CPU torch device; this framework has no CPU path by design — the
    product fails loudly without the HIP library — so the case runs on the GPU with the CPU oracle as the judge.)"""
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(ckpt_06b, use_fast=True)
    strings = ["introduce yourself", "list all prime numbers within 100"]
    id_lists = _prompts(2, 5, 40, 10000, seed=31)
    ids = [tok.encode(t) for t in strings] + id_lists
    max_tokens = [16] * 4
    outs, rec, nblk = _run_ours(ckpt_06b, strings + id_lists, max_tokens, enforce_eager=True, max_model_len=1024,
                                num_kvcache_blocks=16, max_num_seqs=8, dummy_weights=True)
    assert [len(o["token_ids"]) for o in outs] == max_tokens and all(isinstance(o["text"], str) for o in outs)
    cfg, w = _oracle_weights_06b("cuda")
    eng = OracleEngine(OracleQwen3(cfg, w, compiled=True), nblk, 256, max_num_seqs=8)
    eng.keep_logits = True
    for p, m in zip(ids, max_tokens):
        eng.add(p, 0.0, m, True)
    exact = total = 0
    worst = 0.0
    for i, r in enumerate(rec):
        eng.step(forced_tokens=r["tokens"])
        o = eng.trace[-1]
        assert o["is_prefill"] == r["prefill"] and o["tables"] == r["tables"], f"step {i}: schedule differs"
        for row, t in enumerate(r["tokens"]):
            gap = float(o["logits"][row].max() - o["logits"][row, t])
            worst = max(worst, gap)
            exact += gap == 0.0
            total += 1
    print(f"config 1 (0.6B shapes, eager, 2 strings + 2 id lists): {exact}/{total} exact argmax, worst gap {worst:.4f}")
    assert worst <= TOL and exact >= 0.8 * total


def test_qwen3_06b_shape_bench_workload_properties(ckpt_06b):
    """The bench workload's shape (ragged prompts 100-1024, ragged outputs, T=0) at a reduced sequence
    count: (1) the captured-hipGraph engine and the eager engine pick identical tokens — same kernels,
    padded graph rows contribute no work; (2) a second identical run reproduces them bit for bit
    (no atomics / race-dependent summation order anywhere on the path); (3) a sequence decoded alone
    yields the same tokens as inside the batch up to bf16 near-ties (split points of the stream-K
    attention move with the batch): compared on its first tokens only."""
    from random import Random
    from nano_vllm_amd import LLM, SamplingParams
    rnd = Random(0)
    prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(100, 1024))] for _ in range(40)]
    outs_len = [rnd.randint(20, 60) for _ in range(40)]
    sps = [SamplingParams(temperature=0.0, ignore_eos=True, max_tokens=m) for m in outs_len]

    def run(**kw):
        llm = LLM(ckpt_06b, max_model_len=4096, dummy_weights=True, num_kvcache_blocks=400, **kw)
        toks = [o["token_ids"] for o in llm.generate(prompts, sps, use_tqdm=False)]
        llm.exit()
        return toks

    graph = run(enforce_eager=False)
    again = run(enforce_eager=False)
    eager = run(enforce_eager=True)
    assert [len(t) for t in graph] == outs_len
    assert graph == again
    assert graph == eager
    llm = LLM(ckpt_06b, max_model_len=4096, dummy_weights=True, num_kvcache_blocks=400, enforce_eager=True)
    solo = llm.generate([prompts[7]], SamplingParams(temperature=0.0, ignore_eos=True, max_tokens=4), use_tqdm=False)
    llm.exit()
    assert solo[0]["token_ids"][:1] == graph[7][:1]


def test_logits_close_to_oracle(tiny_ckpt):
    """Quantitative logits parity (BASELINE north star: "bf16 logits within 1e-3" is below the reference's own
    eager-vs-compiled floor; the bar used here is 2e-2 * absmax, the measured floor being
    ~2e-2 * absmax on Qwen3-0.6B shapes): logits of every prefill and decode step of a short greedy run vs the
    CPU oracle teacher-forced with the same tokens."""
    from nano_vllm_amd import LLM, SamplingParams
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3, load_weights
    prompts = _prompts(4, 5, 400, 512, seed=17)
    llm = LLM(tiny_ckpt, enforce_eager=True, max_model_len=2048, num_kvcache_blocks=16, max_num_seqs=8)
    ours, toks = [], []
    llm.model_runner.sampler.capture = ours            # every step's logits, whichever sampling path runs
    call = llm.model_runner.call

    def spy(method, *args):
        out = call(method, *args)
        if method in ("run", "decode_end"):
            toks.append(list(out))
        return out

    llm.model_runner.call = spy
    llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True), use_tqdm=False)
    nblk = llm.config.num_kvcache_blocks
    llm.exit()
    cfg, w = load_weights(tiny_ckpt)
    eng = OracleEngine(OracleQwen3(cfg, w, compiled=True), nblk, 256, max_num_seqs=8)
    eng.keep_logits = True
    for p in prompts:
        eng.add(p, 0.0, 6, True)
    worst = 0.0
    assert len(ours) == len(toks)
    for mine, t in zip(ours, toks):
        eng.step(forced_tokens=t)
        ref_logits = eng.trace[-1]["logits"].float()
        assert mine.shape == ref_logits.shape
        rel = float((mine - ref_logits).abs().max() / ref_logits.abs().max())
        worst = max(worst, rel)
    print(f"logits max|diff|/absmax over {len(ours)} steps: {worst:.5f}")
    assert worst <= 2e-2


# ---------------------------------------------------------------------------------------------
# Real layer WIDTHS of the larger models (BASELINE configs 3-5), two layers deep: hidden 4096 / 5120, GQA groups 4
# and 8, intermediate 12,288 / 25,600 — the projections whose K is too deep for the single-pass skinny GEMM take
# the multi-pass / library path with the separate SiLU and add-RMSNorm kernels, which no tiny model reaches.
def _oracle_weights(name, vocab, device="cuda"):
    from nano_vllm_amd.weights import parameter_shapes, qwen3_config_dict, synth_tensor
    cfg = qwen3_config_dict(name, vocab_size=vocab, max_position_embeddings=4096)
    return cfg, {n: synth_tensor(n, s, 0, device=device).cpu() for n, s in parameter_shapes(cfg).items()}


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("name", ["qwen3-8b-2l", "qwen3-32b-2l"])
def test_full_width_layers_greedy_parity_vs_oracle(name, wide, monkeypatch):
    """Real Qwen3-8B / 32B layer widths (2 layers) against the oracle, with the decode projections on the library GEMM
    (wide = 0: hipBLASLt + separate SiLU / add-RMSNorm launches) and on nvl_linear_wide (wide = 1: every decode
    projection, fused SiLU epilogue, split-K slabs into the add-RMSNorm; the engine's default picks per shape by
    timing)."""
    from nano_vllm_amd import layers
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    monkeypatch.setenv("NVL_GEMM_WIDE", wide)
    layers._wide_choice.clear()
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3
    vocab = 2048
    path = tempfile.mkdtemp(prefix=name.replace("-", "_") + "_")
    write_synthetic_checkpoint(path, name, with_weights=False, vocab_size=vocab, max_position_embeddings=4096)
    prompts = _prompts(5, 8, 400, vocab, seed=23)
    max_tokens = [10, 6, 12, 3, 9]
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, enforce_eager=False, max_model_len=1024,
                                num_kvcache_blocks=24, max_num_seqs=8, dummy_weights=True)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    cfg, w = _oracle_weights(name, vocab)
    eng = OracleEngine(OracleQwen3(cfg, w, compiled=True), nblk, 256, max_num_seqs=8)
    eng.keep_logits = True
    for p, m in zip(prompts, max_tokens):
        eng.add(p, 0.0, m, True)
    exact = total = 0
    worst = 0.0
    for i, r in enumerate(rec):
        eng.step(forced_tokens=r["tokens"])
        o = eng.trace[-1]
        assert o["is_prefill"] == r["prefill"] and o["tables"] == r["tables"], f"step {i}: schedule differs"
        for row, tok in enumerate(r["tokens"]):
            gap = float(o["logits"][row].max() - o["logits"][row, tok])
            worst = max(worst, gap)
            exact += gap == 0.0
            total += 1
    used = sum(layers.wide_choices().values())
    layers._wide_choice.clear()
    print(f"{name} wide={wide}: {exact}/{total} exact argmax, worst logit gap {worst:.4f}, wide shapes used {used}")
    assert worst <= TOL and exact >= 0.8 * total
    assert (used > 0) == (wide == "1")


def test_logits_error_vs_exact_arithmetic_is_at_the_reference_floor(tiny_ckpt):
    """SURVEY.md §8c(2) / north star "logits within 1e-3": that bar is only meaningful against EXACT arithmetic,
    not between two bf16 pipelines (the reference's eager and compiled modes already differ by ~2e-2 * absmax).
    Yardstick: the oracle with the same (bf16-valued) weights evaluated in fp32 end to end — no intermediate
    rounding. Measured against it: (a) the reference-faithful bf16 oracle = the floor any bf16 implementation of
    this model pays; (b) ours. Ours must not be worse than 1.5x the floor (+1e-3 * absmax), both are printed."""
    from nano_vllm_amd import LLM, SamplingParams
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3, load_weights
    prompts = _prompts(4, 5, 400, 512, seed=19)
    llm = LLM(tiny_ckpt, enforce_eager=True, max_model_len=2048, num_kvcache_blocks=16, max_num_seqs=8)
    ours, toks = [], []
    llm.model_runner.sampler.capture = ours
    call = llm.model_runner.call

    def spy(method, *args):
        out = call(method, *args)
        if method in ("run", "decode_end"):
            toks.append(list(out))
        return out

    llm.model_runner.call = spy
    llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True), use_tqdm=False)
    nblk = llm.config.num_kvcache_blocks
    llm.exit()
    cfg, w = load_weights(tiny_ckpt)
    engines = {"bf16": OracleEngine(OracleQwen3(cfg, w, compiled=True), nblk, 256, max_num_seqs=8),
               "exact": OracleEngine(OracleQwen3(cfg, {k: v.float() for k, v in w.items()}, compiled=True), nblk, 256,
                                     max_num_seqs=8)}
    for eng in engines.values():
        eng.keep_logits = True
        for p in prompts:
            eng.add(p, 0.0, 6, True)
    floor = err = 0.0
    for mine, t in zip(ours, toks):
        for eng in engines.values():
            eng.step(forced_tokens=t)
        exact = engines["exact"].trace[-1]["logits"].float()
        ref16 = engines["bf16"].trace[-1]["logits"].float()
        scale = float(exact.abs().max())
        floor = max(floor, float((ref16 - exact).abs().max()) / scale)
        err = max(err, float((mine - exact).abs().max()) / scale)
    print(f"max|logits - exact|/absmax: reference-faithful bf16 oracle {floor:.5f} (the floor), ours {err:.5f}")
    assert err <= 1.5 * floor + 1e-3


def test_prompt_longer_than_the_token_budget_end_to_end(tiny_ckpt):
    """BASELINE config 5's path through the ENGINE: one prompt longer than max_num_batched_tokens is prefilled in
    two steps (16,384 tokens, then the rest against the paged cache), next to two short prompts, then decoded;
    scheduling and every sampled token are judged against the oracle."""
    g = torch.Generator().manual_seed(29)
    prompts = [torch.randint(0, 512, (17000,), generator=g).tolist(), torch.randint(0, 512, (40,), generator=g).tolist(),
               torch.randint(0, 512, (300,), generator=g).tolist()]
    max_tokens = [5, 5, 5]
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    path = tempfile.mkdtemp(prefix="qwen3tiny_long_")
    write_synthetic_checkpoint(path, "qwen3-tiny", seed=0, vocab_size=512, max_position_embeddings=20480)
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, enforce_eager=False, max_model_len=20480,
                                num_kvcache_blocks=80, max_num_seqs=8, max_num_batched_tokens=16384)
    assert rec[0]["prefill"] and rec[0]["sched"] == [16384] and rec[1]["prefill"] and rec[1]["sched"][0] == 616
    exact, total, worst = _judge(path, prompts, max_tokens, rec, nblk, max_num_seqs=8, max_num_batched_tokens=16384)
    print(f"17k-token prompt: {exact}/{total} exact argmax, worst logit gap {worst:.4f}")
    assert worst <= TOL and exact >= 0.9 * total


def test_lookahead_and_microbatch_modes_reproduce_the_serial_engine(tiny_ckpt, monkeypatch):
    """Host-loop variants must not change results: the decode lookahead (default) vs the strictly serial loop
    (NVL_LOOKAHEAD=0), and the optional two-chain micro-batched decode graph (NVL_MICROBATCHES=2) — same sampled
    tokens at T=0.8 with a fixed seed (same Philox offsets per step), 40 sequences so that the micro-batched
    graphs (batch >= 32) are exercised. (The micro-batched sampler seeds its second chain differently, so it is
    compared at T=0.)"""
    from nano_vllm_amd import LLM, SamplingParams
    prompts = _prompts(40, 5, 300, 512, seed=31)
    # ragged output lengths: sequences finish at different steps, so batch rows shift between steps and the
    # device-side id feed (nvl_feed_tokens) has to follow them
    lens = [3 + (7 * i) % 23 for i in range(40)]

    def run(temp, **env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        llm = LLM(tiny_ckpt, enforce_eager=False, max_model_len=2048, num_kvcache_blocks=96, max_num_seqs=64, seed=7)
        sps = [SamplingParams(temperature=temp, max_tokens=m, ignore_eos=True) for m in lens]
        outs = llm.generate(prompts, sps, use_tqdm=False)
        llm.exit()
        for k in env:
            monkeypatch.delenv(k)
        return [o["token_ids"] for o in outs]

    assert run(0.8) == run(0.8, NVL_LOOKAHEAD="0")
    assert run(0.0) == run(0.0, NVL_MICROBATCHES="2")


@pytest.mark.parametrize("name", ["qwen3-tiny", "qwen3-tiny-untied", "qwen3-tiny-g8"])
def test_fp8_kv_cache_engine_runs_and_stays_close_to_the_bf16_engine(name):
    """kv_cache_dtype="fp8" end to end (prefill store, fused decode, chunked prefill reading the fp8 cache, hipGraph),
    at group sizes 2, 4 (streaming kernel) and 8 (matrix-core kernel): an extension outside the reference's numerics,
    so the bar is agreement with OUR bf16 engine on the first tokens (one or two forward passes deep, before
    quantisation noise can flip a near-tie and the histories diverge)."""
    from nano_vllm_amd import LLM, SamplingParams
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    tiny_ckpt = tempfile.mkdtemp(prefix=name.replace("-", "_") + "_fp8_")
    write_synthetic_checkpoint(tiny_ckpt, name, seed=2, vocab_size=512, max_position_embeddings=2048)
    prompts = _prompts(8, 5, 700, 512, seed=43)
    sp = SamplingParams(temperature=0.0, max_tokens=16, ignore_eos=True)

    def run(dt):
        llm = LLM(tiny_ckpt, enforce_eager=False, max_model_len=2048, num_kvcache_blocks=32, max_num_seqs=8,
                  max_num_batched_tokens=640, kv_cache_dtype=dt)
        outs = [o["token_ids"] for o in llm.generate(prompts, sp, use_tqdm=False)]
        llm.exit()
        return outs

    a, b = run("bf16"), run("fp8")
    assert all(len(t) == 16 for t in b)
    first = sum(x[0] == y[0] for x, y in zip(a, b))
    agree = sum(sum(p == q for p, q in zip(x, y)) for x, y in zip(a, b)) / (16 * len(a))
    print(f"fp8 KV vs bf16 KV: first tokens equal {first}/{len(a)}, all positions equal {agree:.2f}")
    assert first >= len(a) - 1
