"""End-to-end parity on the MI355X: our engine (HIP kernels, hipGraph decode) vs the oracle engine on the same
synthetic checkpoint and prompts.

Parity definition (implemented in oracle/judge.py): greedy (T = 0). bf16 noise makes a free-running
comparison of random-weight models meaningless after the first near-tie, so the oracle is TEACHER-FORCED with our
tokens and every one of our decisions is judged against the oracle's logits for the same history:
  * the noise FLOOR is measured on the run itself: the same history through the oracle with the reference's eager
    rounding vs the oracle with its compiled rounding (max|dlogit| / absmax — the reference against itself);
  * every row whose oracle top-1 / top-2 margin exceeds 2 x floor must be the oracle's argmax EXACTLY;
  * a sub-margin row (a near-tie below the noise) may differ, but only by a token within 2 x floor of the maximum.
Scheduling (batch composition, block tables) must be identical step for step.
"""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_ours(path, prompts, max_tokens, capture_logits=False, temperatures=None, info=None, **kw):
    """`temperatures` (per prompt; None = greedy). The engine's draw seed is `kw["seed"]` (Config.seed, default 0).
    `info` (dict): receives the runner's counters before the engine is torn down."""
    from nano_vllm_amd import LLM, SamplingParams
    llm = LLM(path, **kw)
    rec = []
    runner = llm.model_runner
    orig = runner.call
    logits_log = []
    if capture_logits:
        assert kw.get("enforce_eager"), "logits are captured by Python code: a replayed hipGraph runs none"
        runner.sampler.capture = logits_log                # every step's logits, whichever sampling path runs

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
    temperatures = temperatures if temperatures is not None else [0.0] * len(prompts)
    sps = [SamplingParams(temperature=t, max_tokens=m, ignore_eos=True) for t, m in zip(temperatures, max_tokens)]
    outs = llm.generate(prompts, sps, use_tqdm=False)
    nblk = llm.config.num_kvcache_blocks
    runner.call = orig
    if info is not None:
        info.update(prefix_steps=runner.prefix_steps, prefix_graphs=sorted(runner.graphs_px),
                    prefix_multi_steps=runner.prefix_multi_steps)
    llm.exit()
    if capture_logits:
        assert len(logits_log) == len(rec)
        for r, lg in zip(rec, logits_log):
            r["logits"] = lg
    return outs, rec, nblk


def _judge_run(cfg, w, prompts, max_tokens, rec, nblk, device=None, temperatures=None, seed=0, **sched_kw):
    from oracle.judge import judge_run
    return judge_run(cfg, w, prompts, max_tokens, rec, nblk, device=device, temperatures=temperatures, seed=seed,
                     **sched_kw)


def _check(name, v, min_rows=1):
    print(v.line(name) + (f"; logits max|ours - oracle| / absmax {v.worst_logit_err_rel:.5f}" if v.worst_logit_err_rel else ""))
    assert v.rows >= min_rows
    assert v.ok(), v.violations[:5]
    # whenever the product's own logits were captured (eager runs), they lie within ONE measured floor
    # (the reference against itself: eager vs compiled rounding, on this very run) of the oracle's
    if v.worst_logit_err_rel:
        assert v.worst_logit_err_rel <= v.floor_rel, (v.worst_logit_err_rel, v.floor_rel)


def _judge(path, prompts, max_tokens, rec, nblk, temperatures=None, seed=0, **sched_kw):
    from oracle.model import load_weights
    cfg, w = load_weights(path)
    return _judge_run(cfg, w, prompts, max_tokens, rec, nblk, temperatures=temperatures, seed=seed, **sched_kw)


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


@pytest.mark.parametrize("eager", [pytest.param(True, marks=pytest.mark.slow), False])
def test_tiny_model_greedy_parity(tiny_ckpt, eager):
    prompts = _prompts(6, 5, 600, 512, seed=3)
    max_tokens = [24, 40, 8, 33, 1, 17]
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, enforce_eager=eager, max_model_len=2048,
                                num_kvcache_blocks=32, max_num_seqs=16)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    _check(f"tiny eager={eager}", _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, max_num_seqs=16), sum(max_tokens))


@pytest.mark.parametrize("name", ["qwen3-tiny-untied", "qwen3-tiny-g8", "qwen3-tiny-g5"])
def test_other_head_geometries_greedy_parity(name):
    """GQA group sizes 4 (Qwen3-8B-like, untied lm_head), 8 with a single kv head (Qwen3-32B at TP=8) and 5 (Qwen3-14B's
    40 / 8, models/qwen3.py:29-38: a group size that does not divide the 16 matrix columns): the G=4 / G=8 / runtime-G
    instantiations of the fused decode kernel and of the MFMA prefill kernel, end to end."""
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    path = tempfile.mkdtemp(prefix=name.replace("-", "_") + "_")
    write_synthetic_checkpoint(path, name, seed=1, vocab_size=512, max_position_embeddings=2048)
    prompts = _prompts(5, 5, 700, 512, seed=9)
    max_tokens = [12, 30, 5, 21, 9]
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, enforce_eager=False, max_model_len=2048,
                                num_kvcache_blocks=24, max_num_seqs=8)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    _check(name, _judge(path, prompts, max_tokens, rec, nblk, max_num_seqs=8), sum(max_tokens))


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
    _check("tiny chunked/preempt", _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, max_num_seqs=8,
                                          max_num_batched_tokens=640), sum(max_tokens))


# ---------------------------------------------------------------------------------------------
# T > 0 — the path the headline bench runs (reference layers/sampler.py:8-12 behind model_runner.py:214-220, bench.py:18
# T = 0.6), judged EXACTLY: the product's draw is a counter-based function of (seed, request ordinal, position, column)
# which oracle/philox.py restates, so every sampled token must be the argmax of `l/T - log E` on the oracle's logits
# whenever the key margin exceeds 2 x floor / T (oracle/judge.py). What this pins end to end: the per-row key staging
# (`rkey` = ordinal | position << 32 in prefill AND decode images, through the lookahead's pre-staged steps), the
# graph-captured sampler reading keys/temperatures from the static device block, mid-prefill chunk rows, and mixed
# greedy / sampled batches.
@pytest.mark.parametrize("eager", [pytest.param(True, marks=pytest.mark.slow), False])   # (eager: --slow; the captured graph takes the same kernels)
def test_tiny_model_sampled_parity_draws_replayed(tiny_ckpt, eager):
    prompts = _prompts(7, 5, 600, 512, seed=3)
    max_tokens = [24, 40, 8, 33, 1, 17, 29]
    temps = [0.6, 1.0, 0.0, 0.6, 0.8, 1.5, 0.3]
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, temperatures=temps, enforce_eager=eager,
                                max_model_len=2048, num_kvcache_blocks=32, max_num_seqs=16, seed=1234)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    v = _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, temperatures=temps, seed=1234, max_num_seqs=16)
    _check(f"tiny T>0 eager={eager}", v, sum(max_tokens))
    assert v.sampled_rows >= sum(max_tokens) - 8
    # the judgement has teeth: the same run under another seed is refused
    assert not _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, temperatures=temps, seed=1235, max_num_seqs=16).ok()


def test_tiny_model_sampled_parity_through_chunked_prefill_and_preemption(tiny_ckpt):
    """T = 0.6 with chunked prefill (a mid-prefill chunk's discarded token is still drawn with the key of ITS end
    position), prefix-cache hits and a preemption by recompute: a re-prefilled sequence must draw, at each position, the
    numbers it would have drawn without the preemption (the key is (request, position), not a step counter)."""
    g = torch.Generator().manual_seed(11)
    shared = torch.randint(0, 512, (512,), generator=g).tolist()
    prompts = [shared + torch.randint(0, 512, (int(n),), generator=g).tolist() for n in (30, 200, 77, 5)]
    prompts.append(torch.randint(0, 512, (900,), generator=g).tolist())
    max_tokens, temps = [20] * 5, [0.6] * 5
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, temperatures=temps, enforce_eager=False,
                                max_model_len=2048, num_kvcache_blocks=9, max_num_seqs=8, max_num_batched_tokens=640,
                                seed=7)
    assert any(r["prefill"] and len(r["seq_ids"]) == 1 for r in rec)
    _check("tiny T=0.6 chunked/preempt", _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, temperatures=temps, seed=7,
                                                max_num_seqs=8, max_num_batched_tokens=640), sum(max_tokens))


@pytest.mark.slow      # (27 s: 320 concurrent sequences in the 512-row bucket; the 256-row bench width stays in the default suite)
def test_tiny_model_batch_above_256_rows_parity(tiny_ckpt):
    """The reference's DEFAULT `max_num_seqs` is 512 (config.py:11): 320 concurrent sequences put every decode step in the
    512-row hipGraph bucket — staging image, per-step attention plan, the GEMMs' row groups (20 row tiles), the sampler and
    the lookahead's token feed all above the 256 rows the bench workload reaches. Mixed greedy / T = 0.6 rows, judged by the
    CPU oracle with the draws replayed."""
    prompts = _prompts(320, 5, 60, 512, seed=71)
    max_tokens = [6] * 320
    temps = [0.0 if i % 3 else 0.6 for i in range(320)]
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, temperatures=temps, enforce_eager=False, max_model_len=256,
                                num_kvcache_blocks=400, max_num_seqs=512, seed=5)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    assert sum(1 for r in rec if not r["prefill"] and len(r["tokens"]) == 320) == 5
    v = _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, temperatures=temps, seed=5, max_num_seqs=512)
    _check("tiny, 320 concurrent sequences (512-row bucket)", v, 320 * 6)
    assert v.sampled_rows >= 100 * 6


@pytest.mark.parametrize("eager", [pytest.param(True, marks=pytest.mark.slow), False])   # (eager: --slow; the captured graph takes the same kernels)
def test_sequences_that_end_exactly_on_max_model_len_and_on_block_edges(tiny_ckpt, eager):
    """Edges of the paged layout end to end: a sequence whose last token fills `max_model_len` exactly (the widest block
    table the staging image carries: prompt 500 + 12 = 512 = two full blocks), one that ends exactly on a block edge (255 +
    1), one whose only generated tokens open a fresh block (256 + 2), a one-token prompt, and `max_tokens = 1` (the token
    comes from the prefill step alone)."""
    g = torch.Generator().manual_seed(83)
    lens, max_tokens = [500, 255, 256, 1, 511, 257], [12, 1, 2, 9, 1, 255]
    prompts = [torch.randint(0, 512, (n,), generator=g).tolist() for n in lens]
    outs, rec, nblk = _run_ours(tiny_ckpt, prompts, max_tokens, enforce_eager=eager, max_model_len=512,
                                num_kvcache_blocks=16, max_num_seqs=8)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    _check(f"block / max_model_len edges eager={eager}", _judge(tiny_ckpt, prompts, max_tokens, rec, nblk, max_num_seqs=8),
           sum(max_tokens))


@pytest.mark.parametrize("name,eager", [pytest.param("qwen3-tiny", True, marks=pytest.mark.slow), ("qwen3-tiny", False),
                                        pytest.param("qwen3-tiny-untied", False, marks=pytest.mark.slow),
                                        pytest.param("qwen3-tiny-g8", False, marks=pytest.mark.slow), ("qwen3-tiny-g5", False)])
def test_shared_system_prompt_runs_the_shared_prefix_attention_pass(name, eager, monkeypatch):
    """BASELINE config 3 in small: nine requests start with the same 530 tokens and one has nothing in common with them.
    The token budget lets the first prefill step take three of them — they compute the prefix themselves and keep private
    copies (block_manager.py:110-120 registers a block after the step that filled it) — the later ones get the two full
    blocks out of the prefix cache with the SAME block ids (:58-82). Decode steps therefore run with a GROUP of rows that
    shares two blocks next to rows that do not, and take the shared-prefix pass (forced on for these tiny shapes:
    NVL_SHARED_PREFIX_MIN_MB=0) — in graph mode through graphs captured when a bucket first wants it, for every bucket
    the shrinking batch passes through — until fewer than two members are left. Tokens, batches and block tables are
    judged against the oracle engine as in every other test here; greedy and sampled rows in one batch."""
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    monkeypatch.setenv("NVL_SHARED_PREFIX_MIN_MB", "0")
    path = tempfile.mkdtemp(prefix=name.replace("-", "_") + "_px_")
    write_synthetic_checkpoint(path, name, seed=2, vocab_size=512, max_position_embeddings=2048)
    g = torch.Generator().manual_seed(97)
    shared = torch.randint(0, 512, (530,), generator=g).tolist()
    prompts = [shared + torch.randint(0, 512, (int(n),), generator=g).tolist() for n in (1, 30, 200, 77, 5, 250, 13, 300, 64)]
    prompts.insert(3, torch.randint(0, 512, (700,), generator=g).tolist())
    max_tokens = [40, 22, 31, 6, 40, 17, 40, 9, 28, 35]
    temps = [0.0, 0.7, 0.0, 0.0, 0.6, 0.0, 0.0, 1.0, 0.0, 0.0]
    info = {}
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, temperatures=temps, info=info, enforce_eager=eager,
                                max_model_len=2048, num_kvcache_blocks=40, max_num_seqs=16, max_num_batched_tokens=2048,
                                seed=5)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    decode = [r for r in rec if not r["prefill"]]
    # the scenario is what the docstring says: some decode step holds rows that share their first two blocks AND rows that don't
    mixed = sum(1 for r in decode if len(r["tables"]) > 2 and 2 <= sum(t[:2] == r["tables"][-1][:2] for t in r["tables"]) < len(r["tables"]))
    assert mixed > 0
    assert 0 < info["prefix_steps"] <= len(decode), (info, len(decode))
    assert eager or len(info["prefix_graphs"]) >= 1
    _check(f"shared system prompt {name} eager={eager}",
           _judge(path, prompts, max_tokens, rec, nblk, temperatures=temps, seed=5, max_num_seqs=16,
                  max_num_batched_tokens=2048), sum(max_tokens))


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


def test_lookahead_equals_serial_when_sampled_eos_ends_sequences(tiny_ckpt, monkeypatch):
    """T = 0.8 WITHOUT ignore_eos: sequences end on a sampled EOS (discovered one step late by the lookahead, whose
    retroactively finished sequence still occupies a row of the next step). The sampler keys its draw by (request
    ordinal, token position), not by (step, batch row), so every request sees the serial loop's RANDOM NUMBERS; what
    can still differ is the last bf16 bit of its logits, because the batch — and with it the split points of the
    stream-K attention — differs by that one dead row (as between any two batch compositions): a rare near-tie flips.
    With row-keyed draws (the round-2 behaviour) every sequence behind the dead row diverged from its first token on.
    The exact form of the claim is tested on CPU (tests/test_engine_host.py: stand-in device, tokens = f(..., rkey))."""
    from nano_vllm_amd import LLM, SamplingParams
    prompts = _prompts(48, 5, 200, 512, seed=51)

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        llm = LLM(tiny_ckpt, enforce_eager=False, max_model_len=1024, num_kvcache_blocks=64, max_num_seqs=32, seed=11)
        llm.config.eos = llm.scheduler.eos = 7                       # a token the tiny model samples often enough
        outs = llm.generate(prompts, SamplingParams(temperature=0.8, max_tokens=64), use_tqdm=False)
        llm.exit()
        for k in env:
            monkeypatch.delenv(k)
        return [o["token_ids"] for o in outs]

    look, serial = run(), run(NVL_LOOKAHEAD="0")
    ended = sum(t[-1] == 7 and len(t) < 64 for t in serial)
    same_seq = sum(a == b for a, b in zip(look, serial))
    n_tok = sum(len(t) for t in serial)
    diff_tok = sum(sum(x != y for x, y in zip(a, b)) + abs(len(a) - len(b)) for a, b in zip(look, serial))
    print(f"{ended}/48 sequences ended on a sampled EOS; {same_seq}/48 sequences identical, {diff_tok}/{n_tok} tokens differ")
    assert ended >= 1 and same_seq >= 44 and diff_tok <= 0.01 * n_tok


def test_packed_weight_copies_obey_their_budget_and_do_not_change_results(tiny_ckpt, monkeypatch):
    """The tile-packed second copies of the decode GEMMs' weights are allocated before the KV pool is sized; under a
    budget that does not fit them (NVL_PACKED_BUDGET_FRAC ~ 0) the projections keep the row-major weight stream — same
    arithmetic in the same order, so the tokens are identical — and the runner reports what it left unpacked."""
    from nano_vllm_amd import LLM, SamplingParams
    prompts = _prompts(5, 5, 300, 512, seed=61)
    sp = SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True)

    def run(frac):
        monkeypatch.setenv("NVL_PACKED_BUDGET_FRAC", frac)
        llm = LLM(tiny_ckpt, enforce_eager=False, max_model_len=1024, num_kvcache_blocks=16, max_num_seqs=8)
        r = llm.model_runner
        stats = (r.packed_weight_bytes, r.packed_weight_skipped)
        toks = [o["token_ids"] for o in llm.generate(prompts, sp, use_tqdm=False)]
        llm.exit()
        return stats, toks

    (full_bytes, full_skipped), a = run("0.25")
    (none_bytes, none_skipped), b = run("1e-12")
    assert full_bytes > 0 and full_skipped == 0 and none_bytes == 0 and none_skipped > 0
    assert a == b


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


def _oracle_weights_06b(device, seed=0):
    """(`dummy_weights=True` engines seed their synthetic weights with Config.seed — the same seed the sampler draws
    with — so an engine started with seed=s is judged on the weights of seed s.)"""
    from nano_vllm_amd.weights import parameter_shapes, qwen3_config_dict, synth_tensor
    cfg = qwen3_config_dict("qwen3-0.6b")
    return cfg, {n: synth_tensor(n, s, seed, device=device).cpu() for n, s in parameter_shapes(cfg).items()}


def test_qwen3_06b_shape_greedy_parity_vs_cpu_oracle_and_device_oracle_agrees(ckpt_06b):
    """Full-size layers (fused decode attention G = 2, skinny decode GEMMs at K = 1024 / 2048 / 3072, sampler over
    151,936 logits) judged by the CPU oracle (both roundings, floor measured on the run) under the margin rule.
    The same history is also run through the SAME oracle code with its tensors on the GPU (torch's own kernels): its
    logits must agree with the CPU run to within ONE floor (two executions of the same restatement on different GEMM
    libraries are exactly the kind of pair the floor describes; measured 0.7 floors) — that is what licenses the
    device-resident oracle for the config-2-sized judgement below, which the CPU cannot finish in minutes."""
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3
    prompts = _prompts(3, 20, 300, 10000, seed=21)
    max_tokens = [16, 14, 18]
    outs, rec, nblk = _run_ours(ckpt_06b, prompts, max_tokens, enforce_eager=False, max_model_len=1024,
                                num_kvcache_blocks=16, max_num_seqs=8, dummy_weights=True)
    cfg, w = _oracle_weights_06b("cuda")
    v = _judge_run(cfg, w, prompts, max_tokens, rec, nblk, max_num_seqs=8)
    _check("0.6B shapes (CPU oracle)", v, sum(max_tokens))
    engs = [OracleEngine(OracleQwen3(cfg, w, compiled=True, device=d), nblk, 256, max_num_seqs=8) for d in (None, "cuda")]
    worst = 0.0
    for eng in engs:
        eng.keep_logits = True
        for p_, m in zip(prompts, max_tokens):
            eng.add(p_, 0.0, m, True)
    for r in rec[:6]:                                     # the prefill step and five decode steps
        for eng in engs:
            eng.step(forced_tokens=r["tokens"])
        a_, b_ = engs[0].trace[-1]["logits"], engs[1].trace[-1]["logits"]
        worst = max(worst, float((a_ - b_).abs().max()) / float(a_.abs().max()))
    print(f"device-resident oracle vs CPU oracle: max|dlogit| / absmax {worst:.5f} (floor of this run {v.floor_rel:.5f})")
    assert worst <= v.floor_rel


def test_qwen3_06b_width_eager_logits_within_one_floor_of_the_cpu_oracle(ckpt_06b):
    """SURVEY.md §8c(2) at a REAL width: the product's own logits (eager engine, every prefill and decode step captured
    in front of the sampler: 151,936 columns, 28 layers, the skinny decode GEMMs, fused decode attention, lm_head) against
    the CPU oracle's for the same history — max|ours - oracle| / absmax <= 1 x the floor measured on this run (the
    reference against itself, eager vs compiled rounding; `_check` asserts it whenever logits were captured). Tokens are
    judged as everywhere else."""
    prompts = _prompts(3, 20, 300, 10000, seed=27)
    max_tokens = [7, 5, 9]
    outs, rec, nblk = _run_ours(ckpt_06b, prompts, max_tokens, capture_logits=True, enforce_eager=True, max_model_len=1024,
                                num_kvcache_blocks=16, max_num_seqs=8, dummy_weights=True)
    assert all("logits" in r for r in rec)
    cfg, w = _oracle_weights_06b("cuda")
    v = _judge_run(cfg, w, prompts, max_tokens, rec, nblk, max_num_seqs=8)
    assert v.worst_logit_err_rel > 0.0
    _check("0.6B width, eager, logits captured (CPU oracle)", v, sum(max_tokens))


@pytest.mark.slow      # (35 s; the T = 0.6 twin below — the path the bench runs — and the greedy 0.6B cases above stay in the default suite)
def test_config2_shaped_batch_greedy_parity_vs_device_oracle(ckpt_06b):
    """BASELINE.json config 2's regime at Qwen3-0.6B width: 64 sequences with the bench's ragged prompt lengths
    (100-1024 tokens, ids < 10,000, seeded like the reference bench.py), 33 output tokens each => three 16,384-token
    prefill batches and 32 hipGraph decode steps at B = 64, ~2,100 judged decisions. Judge: the device-resident oracle
    (validated against the CPU oracle above), teacher-forced, both roundings, margin rule."""
    from random import Random
    rnd = Random(0)
    prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(100, 1024))] for _ in range(64)]
    max_tokens = [33] * 64
    outs, rec, nblk = _run_ours(ckpt_06b, prompts, max_tokens, enforce_eager=False, max_model_len=4096,
                                num_kvcache_blocks=400, max_num_seqs=64, dummy_weights=True)
    assert sum(1 for r in rec if not r["prefill"] and len(r["tokens"]) == 64) == 32
    cfg, w = _oracle_weights_06b("cuda")
    v = _judge_run(cfg, w, prompts, max_tokens, rec, nblk, device="cuda", max_num_seqs=64)
    _check("config-2-shaped batch (64 seqs x 33 tokens, 0.6B width)", v, 64 * 33)


def test_config2_shaped_batch_sampled_T06_parity_vs_device_oracle(ckpt_06b):
    """The headline configuration's OWN path judged end to end: the bench's ragged prompts (seeded like the reference
    bench.py), temperature 0.6, `ignore_eos`, 64 sequences x 33 tokens => three prefill batches + 32 hipGraph decode
    steps at B = 64 with the lookahead on, full 151,936-column vocabulary. Every one of the 2,112 sampled tokens must be
    the argmax of `l/0.6 - log E` on the device-resident oracle's logits (draws replayed by oracle/philox.py) wherever
    the key margin exceeds 2 x floor / T."""
    from random import Random
    rnd = Random(0)
    prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(100, 1024))] for _ in range(64)]
    max_tokens, temps = [33] * 64, [0.6] * 64
    outs, rec, nblk = _run_ours(ckpt_06b, prompts, max_tokens, temperatures=temps, enforce_eager=False,
                                max_model_len=4096, num_kvcache_blocks=400, max_num_seqs=64, dummy_weights=True, seed=0)
    assert sum(1 for r in rec if not r["prefill"] and len(r["tokens"]) == 64) == 32
    greedy_like = sum(len(set(o["token_ids"])) for o in outs)
    assert greedy_like > 64 * 8                            # (sampled text, not a stuck argmax loop)
    cfg, w = _oracle_weights_06b("cuda")
    v = _judge_run(cfg, w, prompts, max_tokens, rec, nblk, device="cuda", temperatures=temps, seed=0, max_num_seqs=64)
    _check("config-2-shaped batch at T = 0.6 (64 seqs x 33 tokens, 0.6B width)", v, 64 * 33)
    assert v.sampled_rows == 64 * 33


def test_full_decode_batch_of_the_bench_at_06b_width_sampled_parity(ckpt_06b):
    """The bench's decode batch at its FULL width: 256 sequences (short prompts, so that the oracle finishes in seconds),
    T = 0.6, 6 tokens each => 5 hipGraph decode steps at B = 256 over the full 151,936-column vocabulary — 16 row tiles in
    the skinny GEMMs, the 256-row sampler, 2,048 (sequence, kv head) segments in the stream-K attention plan. Judged like
    the 64-sequence case above (device-resident oracle, draws replayed, margin rule)."""
    from random import Random
    rnd = Random(3)
    prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(20, 120))] for _ in range(256)]
    max_tokens, temps = [6] * 256, [0.6] * 256
    outs, rec, nblk = _run_ours(ckpt_06b, prompts, max_tokens, temperatures=temps, enforce_eager=False,
                                max_model_len=1024, num_kvcache_blocks=400, max_num_seqs=256, dummy_weights=True, seed=0)
    assert sum(1 for r in rec if not r["prefill"] and len(r["tokens"]) == 256) == 5
    cfg, w = _oracle_weights_06b("cuda")
    v = _judge_run(cfg, w, prompts, max_tokens, rec, nblk, device="cuda", temperatures=temps, seed=0, max_num_seqs=256)
    _check("full bench decode batch at T = 0.6 (256 seqs x 6 tokens, 0.6B width)", v, 256 * 6)
    assert v.sampled_rows == 256 * 6


def test_config1_example_prompts_eager_exact_tokens(ckpt_06b):
    """BASELINE.json config 1: 0.6B shapes, enforce_eager, the reference example's two prompt strings
    (example.py:12-15) + two token-id lists, max_tokens 16, T = 0, judged under the margin rule by the CPU oracle.
    (The reference runs this case on a CPU torch device; this framework has no CPU path by design — the
    product fails loudly without the HIP library — so the case runs on the GPU with the CPU oracle as the judge.)"""
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
    _check("config 1 (0.6B shapes, eager, 2 strings + 2 id lists)",
           _judge_run(cfg, w, ids, max_tokens, rec, nblk, max_num_seqs=8), 64)


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
@pytest.mark.parametrize("name", ["qwen3-8b-2l", "qwen3-14b-2l", "qwen3-32b-2l"])
def test_full_width_layers_greedy_parity_vs_oracle(name, wide, monkeypatch):
    """Real Qwen3-8B / 14B (40 / 8 heads: group size 5) / 32B layer widths (2 layers) against the oracle, with the decode projections on the library GEMM
    (wide = 0: hipBLASLt + separate SiLU / add-RMSNorm launches) and on nvl_linear_wide (wide = 1: every decode
    projection, fused SiLU epilogue, split-K slabs into the add-RMSNorm; the engine's default picks per shape by
    timing)."""
    from nano_vllm_amd import layers
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    monkeypatch.setenv("NVL_GEMM_WIDE", wide)
    layers._wide_choice.clear()
    vocab = 2048
    path = tempfile.mkdtemp(prefix=name.replace("-", "_") + "_")
    write_synthetic_checkpoint(path, name, with_weights=False, vocab_size=vocab, max_position_embeddings=4096)
    prompts = _prompts(5, 8, 400, vocab, seed=23)
    max_tokens = [10, 6, 12, 3, 9]
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, enforce_eager=False, max_model_len=1024,
                                num_kvcache_blocks=24, max_num_seqs=8, dummy_weights=True)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    cfg, w = _oracle_weights(name, vocab)
    v = _judge_run(cfg, w, prompts, max_tokens, rec, nblk, device="cuda", max_num_seqs=8)
    used = sum(layers.wide_choices().values())
    layers._wide_choice.clear()
    _check(f"{name} wide={wide} (wide shapes used {used})", v, sum(max_tokens))
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
    _check("17k-token prompt", _judge(path, prompts, max_tokens, rec, nblk, max_num_seqs=8,
                                      max_num_batched_tokens=16384), sum(max_tokens))


def test_lookahead_reproduces_the_serial_engine(tiny_ckpt, monkeypatch):
    """Host-loop variants must not change results: the decode lookahead (default) vs the strictly serial loop
    (NVL_LOOKAHEAD=0) — same sampled tokens at T=0.8 with a fixed seed, 40 sequences of ragged output lengths."""
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


@pytest.mark.parametrize("eager", [pytest.param(True, marks=pytest.mark.slow), False])   # (eager: --slow; the captured graph takes the same kernels)
def test_kvcache_block_size_512_with_a_shared_system_prompt(eager, monkeypatch):
    """`kvcache_block_size` may be any multiple of 256 (config.py:22). 512-token blocks end to end: KV store and paged
    prefill across 512-token blocks, prefix-cache hits on two full 512-token blocks (block_manager.py:58-82 hashes whole
    blocks of the CONFIGURED size), decode attention walking 16 tiles per block, and the shared-prefix pass over the two
    common blocks (forced on for these tiny shapes) — judged against the oracle engine running the same block size
    (schedule, block tables and tokens)."""
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    monkeypatch.setenv("NVL_SHARED_PREFIX_MIN_MB", "0")
    path = tempfile.mkdtemp(prefix="qwen3_tiny_bs512_")
    write_synthetic_checkpoint(path, "qwen3-tiny-untied", seed=4, vocab_size=512, max_position_embeddings=4096)
    g = torch.Generator().manual_seed(211)
    shared = torch.randint(0, 512, (1040,), generator=g).tolist()             # two full 512-token blocks + 16 tokens
    prompts = [shared + torch.randint(0, 512, (int(n),), generator=g).tolist() for n in (2, 300, 45, 700, 130, 9)]
    prompts.insert(2, torch.randint(0, 512, (900,), generator=g).tolist())    # a row that shares nothing
    max_tokens = [24, 9, 17, 24, 5, 24, 13]
    temps = [0.0, 0.0, 0.8, 0.0, 0.0, 0.6, 0.0]
    info = {}
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, temperatures=temps, info=info, enforce_eager=eager,
                                max_model_len=4096, num_kvcache_blocks=40, max_num_seqs=16, max_num_batched_tokens=2560,
                                kvcache_block_size=512, seed=3)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    decode = [r for r in rec if not r["prefill"]]
    shared_rows = max(sum(t[:2] == r["tables"][-1][:2] for t in r["tables"]) for r in decode)
    assert shared_rows >= 2, "no decode step held two rows with the same two leading blocks"
    assert info["prefix_steps"] > 0, info
    _check(f"kvcache_block_size=512, shared system prompt, eager={eager}",
           _judge(path, prompts, max_tokens, rec, nblk, temperatures=temps, seed=3, block_size=512, max_num_seqs=16,
                  max_num_batched_tokens=2560), sum(max_tokens))


def test_two_shared_system_prompts_in_one_batch(monkeypatch):
    """Two groups of requests, each with its own 530-token system prompt, decode in ONE batch. The step's shared-prefix
    pass serves BOTH groups (engine/runner.py `shared_prefix_group`: group ids; round 5 shared the largest group only):
    rows of one group share their two leading blocks among themselves, not with the other group, and a pack of rows may
    hold members of both. Judged against the oracle engine (schedule, block tables, tokens; greedy and sampled rows)."""
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    monkeypatch.setenv("NVL_SHARED_PREFIX_MIN_MB", "0")
    path = tempfile.mkdtemp(prefix="qwen3_tiny_two_prefixes_")
    write_synthetic_checkpoint(path, "qwen3-tiny", seed=6, vocab_size=512, max_position_embeddings=2048)
    g = torch.Generator().manual_seed(307)
    sys_a = torch.randint(0, 512, (530,), generator=g).tolist()
    sys_b = torch.randint(0, 512, (530,), generator=g).tolist()
    tails = [int(n) for n in (3, 90, 41, 200, 17, 66, 120, 8, 150, 33)]
    which = [0, 1, 0, 0, 1, 0, 1, 0, 1, 0]                                     # six requests on prompt A, four on prompt B
    prompts = [(sys_a, sys_b)[w] + torch.randint(0, 512, (n,), generator=g).tolist() for w, n in zip(which, tails)]
    max_tokens = [30, 30, 12, 30, 30, 7, 21, 30, 30, 16]
    temps = [0.0, 0.0, 0.0, 0.7, 0.0, 0.0, 0.0, 0.0, 0.9, 0.0]
    info = {}
    outs, rec, nblk = _run_ours(path, prompts, max_tokens, temperatures=temps, info=info, enforce_eager=False,
                                max_model_len=2048, num_kvcache_blocks=48, max_num_seqs=16, max_num_batched_tokens=1280,
                                seed=8)
    assert [len(o["token_ids"]) for o in outs] == max_tokens
    # the scenario: some decode step holds >= 2 rows on prompt A's blocks AND >= 2 rows on prompt B's blocks
    def groups(r):
        heads = [tuple(t[:2]) for t in r["tables"] if len(t) > 2]
        return sorted((heads.count(h) for h in set(heads)), reverse=True)
    assert any(len(c) >= 2 and c[1] >= 2 for c in map(groups, (r for r in rec if not r["prefill"]))), "never two shared groups"
    assert info["prefix_steps"] > 0 and info["prefix_multi_steps"] > 0, info
    assert any(k[1] > 1 for k in info["prefix_graphs"]), info           # a graph with several group slots was captured
    _check("two shared system prompts in one batch",
           _judge(path, prompts, max_tokens, rec, nblk, temperatures=temps, seed=8, max_num_seqs=16,
                  max_num_batched_tokens=1280), sum(max_tokens))


@pytest.mark.parametrize("name", ["qwen3-tiny", "qwen3-tiny-untied", "qwen3-tiny-g8", "qwen3-tiny-g5"])
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
