"""Host side of the engine on CPU: the REAL LLMEngine loop (`_step_lookahead`), scheduler, block manager and
ModelRunner staging code (`prepare_prefill`, `prepare_decode`, `decode_begin`, `stage_next_decode`) driven by a
stand-in for the device: the "model" is a deterministic function of (input id, position) evaluated on the
staged images exactly as the device would consume them (including nvl_feed_tokens' row indirection).

Checks, on workloads with more prompts than max_num_seqs, ragged output lengths, chunked prefill and a block
pool small enough to force preemption:
  * the lookahead loop and the strictly serial loop produce identical outputs;
  * every staged decode image agrees with its sequences (ids after the feed, positions, context lengths, slots,
    and the block-table rows — which are cached per row and must be refreshed when a preempted sequence comes
    back with other block ids);
  * requests that cannot fit max_model_len are refused up front.
Plus the rank-0 -> workers control channel (ring of slots in shared memory).
"""
import os
import sys
import threading
import time
from random import Random

import numpy as np
import pytest
import torch

from nano_vllm_amd.api import Config, SamplingParams
from nano_vllm_amd.engine.core import LLMEngine
from nano_vllm_amd.engine.runner import ModelRunner, _Channel
from nano_vllm_amd.engine.sched import Scheduler
from nano_vllm_amd.engine.seq import Sequence

VOCAB = 1000


def _next_token(ids, pos, rkey=None):
    """The stand-in model + sampler: a function of (input id, position) and — like the real sampler's counter-based
    draw — of the row's staged key (request ordinal | position << 32), never of the batch row or the step number."""
    t = np.asarray(ids, dtype=np.int64) * 1103515245 + np.asarray(pos, dtype=np.int64) * 12345 + 1
    if rkey is not None:
        t = t + (np.asarray(rkey, dtype=np.int64) % 1000003) * 7919
    return t % VOCAB


class _HF:
    num_attention_heads, num_key_value_heads, hidden_size, num_hidden_layers = 4, 2, 512, 2
    intermediate_size, vocab_size, rms_norm_eps, max_position_embeddings = 512, VOCAB, 1e-6, 4096
    head_dim, tie_word_embeddings, torch_dtype = 128, True, "bfloat16"


class FakeRunner(ModelRunner):
    """ModelRunner with the device replaced by `_next_token`; all staging code is the product's."""

    def __init__(self, config):
        self.config = config
        self.block_size = config.kvcache_block_size
        self.world_size, self.rank, self.chan = 1, 0, None
        self.device = torch.device("cpu")
        self.geo = dict(heads=4, kv_heads=2, hidden=512, layers=2)
        self.graphs = {}
        self._alloc_stages()
        self.tokens = np.zeros(config.max_num_seqs + 8, dtype=np.int64)       # "tokens_dev"
        self.checked_rows = 0
        self._cur = None

    def decode_begin(self, seqs, staged=False):
        self._cur = list(seqs)
        return super().decode_begin(seqs, staged)

    def _launch_decode(self, n):
        st = self.dstage.np
        seqs = self._cur
        assert len(seqs) == n
        ids = st["ids"][:n].copy()
        src = st["src"][:n]
        fed = src >= 0
        ids[fed] = self.tokens[src[fed]]                   # nvl_feed_tokens
        bs = self.block_size
        for i, s in enumerate(seqs):
            # with the lookahead the newest token value may still be a placeholder on the host: the id must then
            # come from the device feed; everything else is known
            if s.last_token != Scheduler.PLACEHOLDER:
                assert ids[i] == s.last_token, (i, ids[i], s.last_token)
            assert st["pos"][i] == s.num_tokens - 1 and st["ctx"][i] == s.num_tokens
            assert st["rkey"][i] == s.rng_key | (s.num_tokens << 32)          # sampler key: (request, position drawn)
            assert st["slots"][i] == s.block_table[-1] * bs + (s.num_tokens - 1) % bs
            row = st["bt"][i]
            assert list(row[:len(s.block_table)]) == s.block_table, (i, s.seq_id, list(row[:8]), s.block_table)
            assert (row[len(s.block_table):] == -1).all()
            self.checked_rows += 1
        assert (st["ctx"][n:self.max_bs] == 0).all() and (st["slots"][n:self.max_bs] == -1).all()
        # shared-prefix group of the step (count 0 unless the pass is switched on): against a direct count on the sequences
        want, members = 0, []
        if self.share_prefix and n >= 2:
            firsts = [s.block_table[0] for s in seqs]
            top = max(set(firsts), key=lambda v: (firsts.count(v), -v))
            members = [i for i, v in enumerate(firsts) if v == top]
            if len(members) >= 2:
                cap = (min(seqs[i].num_tokens for i in members) - 1) // bs
                ref0 = seqs[members[0]].block_table
                while want < cap and all(seqs[i].block_table[want] == ref0[want] for i in members):
                    want += 1
        assert st["shp"][0] == want, (int(st["shp"][0]), want)
        if want:
            assert list(np.nonzero(st["shp"][1:1 + n])[0]) == members
            self.shared_partial = getattr(self, "shared_partial", 0) + (len(members) < n)
        self.shared_seen = max(getattr(self, "shared_seen", 0), want)
        toks = _next_token(ids, st["pos"][:n], st["rkey"][:n])
        self.tokens[:n] = toks
        self._inflight.append((n, toks.copy()))

    def decode_end(self):
        n, toks = self._inflight.pop(0)
        return toks.tolist()

    def _run_prefill(self, seqs):
        info = self.prepare_prefill(seqs)
        st = self.pstage.np
        cu = st["cu_q"][:info["ns"] + 1]
        last = cu[1:] - 1
        toks = _next_token(st["ids"][last], st["pos"][last], st["rkey"][:info["ns"]])
        n = 0
        for i, s in enumerate(seqs):                        # staged chunk == the sequence's scheduled tokens
            lq = s.num_scheduled_tokens
            assert list(st["ids"][n:n + lq]) == s.token_ids[s.num_cached_tokens:s.num_cached_tokens + lq]
            n += lq
        self.tokens[:info["ns"]] = toks
        return toks.tolist()

    def exit(self):
        pass


def _engine(lookahead: bool, **cfg_kw):
    Sequence.counter = __import__("itertools").count()
    cfg = Config(os.path.dirname(__file__), hf_config=_HF(), **cfg_kw)
    eng = LLMEngine.__new__(LLMEngine)
    eng.config = cfg
    Sequence.block_size = cfg.kvcache_block_size
    eng.ps = []
    eng.model_runner = FakeRunner(cfg)
    eng.tokenizer = None
    eng.scheduler = Scheduler(cfg)
    eng._lookahead = lookahead
    eng._unfilled = None
    eng._requests = 0
    eng._exited = True
    return eng


def _generate(eng, prompts, sps):
    for p, sp in zip(prompts, sps):
        eng.add_request(p, sp)
    done = {}
    pending = None
    steps = 0
    while not eng.is_finished() or pending is not None or eng._unfilled is not None:
        finished, _, pending = eng._step_lookahead(pending)
        for sid, toks in finished:
            done[sid] = list(toks)
        steps += 1
        assert steps < 100000
    return [done[k] for k in sorted(done)]


def _workload(seed, n, lo, hi, out_lo, out_hi):
    r = Random(seed)
    prompts = [[r.randint(0, VOCAB - 1) for _ in range(r.randint(lo, hi))] for _ in range(n)]
    sps = [SamplingParams(temperature=0.0, ignore_eos=True, max_tokens=r.randint(out_lo, out_hi)) for _ in range(n)]
    return prompts, sps


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lookahead_equals_serial_with_more_prompts_than_rows(seed):
    """running > max_num_seqs => a sequence that was NOT in the in-flight step enters the next
    one; its id must be staged, not looked up in the in-flight step's rows."""
    prompts, sps = _workload(seed, 14, 3, 40, 1, 9)
    kw = dict(max_num_seqs=4, max_model_len=1024, num_kvcache_blocks=64, max_num_batched_tokens=64)
    a = _engine(True, **kw)
    out_a = _generate(a, prompts, sps)
    b = _engine(False, **kw)
    out_b = _generate(b, prompts, sps)
    assert out_a == out_b
    assert [len(t) for t in out_a] == [sp.max_tokens for sp in sps]
    assert a.model_runner.checked_rows > 0


@pytest.mark.parametrize("seed", [0, 1])
def test_staged_shared_prefix_group(seed):
    """Requests that start with the same tokens get the same leading block ids from the prefix cache once the first of
    them has been registered (block_manager.py:58-82, :110-120: the requests of the prefill step that first computes the
    prefix keep private copies). Every staged decode image carries the group of rows that share their leading blocks and
    how many blocks that is (checked in FakeRunner._launch_decode against a direct count) — never the block a member is
    still writing; rows outside the group (other prompts, private copies) are not members. The same workload generates
    the same tokens with the staging switched off."""
    r = Random(seed)
    bs = 256
    common = [r.randint(0, VOCAB - 1) for _ in range(2 * bs + 17)]
    prompts = [common + [r.randint(0, VOCAB - 1) for _ in range(r.randint(0, 300))] for _ in range(7)]
    prompts.insert(2, [r.randint(0, VOCAB - 1) for _ in range(40)])
    prompts.append(common[:2 * bs])              # ends exactly on the shared blocks' edge: its newest token opens block 2
    sps = [SamplingParams(temperature=0.0, ignore_eos=True, max_tokens=m) for m in (9, 30, 4, 12, 25, 300, 7, 18, 11)]
    # a 1200-token budget: the first prefill step takes the first request and part of the second (both compute the prefix:
    # private copies), the later ones hit the cache
    kw = dict(max_num_seqs=8, max_model_len=2048, num_kvcache_blocks=40, max_num_batched_tokens=1200)
    outs = []
    for on in (True, False):
        eng = _engine(True, **kw)
        eng.model_runner.share_prefix = on
        eng.model_runner.share_prefix_min_bytes = 0.0
        outs.append(_generate(eng, prompts, sps))
        assert eng.model_runner.shared_seen == (2 if on else 0)
        assert not on or eng.model_runner.shared_partial > 0         # steps in which some rows were not members
    assert outs[0] == outs[1]


def test_shared_prefix_group_helper_and_the_launch_threshold():
    from nano_vllm_amd.engine.runner import shared_prefix_group
    bt = np.full((6, 8), -1, dtype=np.int32)
    bt[:5, :3] = [7, 9, 4]
    bt[:5, 3] = [10, 11, 12, 13, 14]
    bt[5, :4] = [20, 21, 22, 23]                 # a private copy of the same content: not a member
    lens = np.array([900, 800, 1000, 770, 1024, 1000])
    k, mem = shared_prefix_group(bt, lens, 256)
    assert k == 3 and mem.tolist() == [1] * 5 + [0]
    assert shared_prefix_group(bt, np.array([900, 800, 1000, 769, 1024, 1000]), 256)[0] == 3      # (769 - 1) // 256 = 3
    assert shared_prefix_group(bt, np.array([900, 800, 1000, 768, 1024, 1000]), 256)[0] == 2      # block 2 holds that row's newest token
    assert shared_prefix_group(bt, np.array([900, 800, 1000, 770, 1024, 10]), 256)[0] == 3        # a short NON-member does not clamp
    assert shared_prefix_group(bt[:1], lens[:1], 256) == (0, None)                                   # one row shares with nobody
    bt2 = bt.copy()
    bt2[3, 1] = 99
    assert shared_prefix_group(bt2, lens, 256)[0] == 1                  # members agree on the first block only
    bt2[:, 0] = np.arange(6)
    assert shared_prefix_group(bt2, lens, 256) == (0, None)             # all first blocks differ
    assert shared_prefix_group(np.full((4, 8), -1, dtype=np.int32), np.array([300] * 4), 256) == (0, None)   # empty tables
    # SEVERAL groups (two system prompts in one batch): the largest group is group 1 and sets k; a further group joins with
    # the next id when its rows agree on THEIR first k blocks and are long enough; a third one that is too short, one that
    # disagrees inside the k blocks, and single rows stay plain
    bt3 = np.full((12, 8), -1, dtype=np.int32)
    bt3[0:5, :3] = [7, 9, 4]                      # group A: 5 rows, 3 common blocks
    bt3[5:8, :3] = [30, 31, 32]                   # group B: 3 rows, 3 common blocks
    bt3[8:10, :3] = [40, 41, 42]                  # group C: 2 rows, too short for 3 blocks
    bt3[10, :3] = [50, 51, 52]                    # a single row
    bt3[11, :3] = [30, 99, 32]                    # starts like group B but disagrees on block 1: breaks B's agreement
    lens3 = np.array([900, 800, 1000, 770, 1024, 800, 801, 900, 600, 601, 900, 900])
    k3, mem3 = shared_prefix_group(bt3[:11], lens3[:11], 256)
    assert k3 == 3 and mem3.tolist() == [1] * 5 + [2] * 3 + [0, 0] + [0]
    k3, mem3 = shared_prefix_group(bt3, lens3, 256)
    assert k3 == 3 and mem3.tolist() == [1] * 5 + [0] * 7          # B's rows no longer agree on their first 3 blocks
    k3, mem3 = shared_prefix_group(bt3[:11], lens3[:11], 256, max_groups=1)
    assert mem3.tolist() == [1] * 5 + [0] * 6
    # the threshold: K/V bytes saved per layer = blocks x 256 tokens x (members - packs) x Hkv x 2 x 128 x 2 B
    eng = _engine(True, max_num_seqs=8, max_model_len=2048, num_kvcache_blocks=40)
    run = eng.model_runner                       # geometry: 4 query heads, 2 kv heads => packs of 8 rows
    saved = 3 * 256 * (5 - 1) * 2 * 2 * 128 * 2
    run.share_prefix_min_bytes = saved
    assert run._prefix_group_worth_a_pass(bt, lens, 6)[0] == 3
    run.share_prefix_min_bytes = saved + 1
    assert run._prefix_group_worth_a_pass(bt, lens, 6) == (0, None)


@pytest.mark.parametrize("seed", range(6))
def test_staged_block_tables_under_preemption(seed):
    """tiny block pool, sequences crossing block boundaries => preemption + re-prefill with
    prefix-cache revival; every staged block-table row must equal its sequence's table (checked by FakeRunner on
    every decode row), in both host loops, and the outputs must agree."""
    r = Random(100 + seed)
    shared = [r.randint(0, VOCAB - 1) for _ in range(256)]
    prompts, sps = [], []
    for i in range(10):
        tail = [r.randint(0, VOCAB - 1) for _ in range(r.randint(200, 300))]
        prompts.append((shared if i % 2 == 0 else []) + tail)
        sps.append(SamplingParams(temperature=0.0, ignore_eos=True, max_tokens=r.randint(40, 120)))
    kw = dict(max_num_seqs=6, max_model_len=2048, num_kvcache_blocks=12, max_num_batched_tokens=700)
    a = _engine(True, **kw)
    preempts = []
    orig = a.scheduler.preempt
    a.scheduler.preempt = lambda s: (preempts.append(s.seq_id), orig(s))[1]
    out_a = _generate(a, prompts, sps)
    out_b = _generate(_engine(False, **kw), prompts, sps)
    assert out_a == out_b
    assert preempts, "workload must preempt"


@pytest.mark.parametrize("seed", range(5))
def test_lookahead_with_eos_terminated_sequences_equals_serial(seed, monkeypatch):
    """Sequences WITHOUT ignore_eos: the lookahead enqueues step N+1 before it knows whether step N sampled EOS and
    finishes such a sequence retroactively. Outputs (tokens and lengths, EOS included as the last token) must equal
    the strictly serial loop's; more prompts than rows, a tight block pool (preemption) and a 1-in-23 EOS rate."""
    monkeypatch.setattr(sys.modules[__name__], "VOCAB", 23)
    r = Random(300 + seed)
    prompts = [[r.randint(0, 22) for _ in range(r.randint(3, 300))] for _ in range(16)]
    sps = [SamplingParams(temperature=0.0, ignore_eos=(i % 5 == 4), max_tokens=r.randint(5, 90)) for i in range(16)]
    kw = dict(max_num_seqs=5, max_model_len=1024, num_kvcache_blocks=9, max_num_batched_tokens=400)
    outs = {}
    for mode in (True, False):
        eng = _engine(mode, **kw)
        eng.scheduler.eos = 7
        outs[mode] = _generate(eng, prompts, sps)
        assert eng.scheduler.block_manager.num_free == 9, "every block returned"
    assert outs[True] == outs[False]
    stopped = [t for t, sp in zip(outs[True], sps) if not sp.ignore_eos and len(t) < sp.max_tokens]
    assert stopped and all(t[-1] == 7 and 7 not in t[:-1] for t in stopped), "some sequences must stop at EOS"
    for t, sp in zip(outs[True], sps):
        if sp.ignore_eos:
            assert len(t) == sp.max_tokens


def test_row_cache_is_refreshed_when_a_sequence_returns_with_other_blocks():
    """The exact ADVICE scenario, constructed: same row, same sequence id, same NUMBER of blocks, other ids."""
    eng = _engine(False, max_num_seqs=4, max_model_len=1024, num_kvcache_blocks=16)
    runner = eng.model_runner
    s = Sequence(list(range(300)), SamplingParams(temperature=0.0, ignore_eos=True, max_tokens=8))
    bm = eng.scheduler.block_manager
    bm.allocate(s, 0)
    first = list(s.block_table)
    for img in range(2):
        runner.dstage.flip()
        runner.prepare_decode([s])
        assert list(runner.dstage.np["bt"][0, :2]) == first
    bm.deallocate(s)
    other = Sequence(list(range(1000, 1300)))
    bm.allocate(other, 0)                                   # takes the head of the free list
    bm.allocate(s, 0)
    assert len(s.block_table) == len(first) and s.block_table != first
    for img in range(2):
        runner.dstage.flip()
        runner.prepare_decode([s])
        assert list(runner.dstage.np["bt"][0, :2]) == s.block_table


def test_over_length_request_is_refused_up_front():
    eng = _engine(True, max_num_seqs=4, max_model_len=512, num_kvcache_blocks=8)
    with pytest.raises(AssertionError, match="max_model_len"):
        eng.add_request(list(range(500)), SamplingParams(max_tokens=13))
    eng.add_request(list(range(500)), SamplingParams(max_tokens=12))
    # generate() validates every request before it enqueues any: a refused batch leaves nothing behind
    eng = _engine(True, max_num_seqs=4, max_model_len=512, num_kvcache_blocks=8)
    with pytest.raises(AssertionError, match="max_model_len"):
        eng.generate([list(range(10)), list(range(500))], [SamplingParams(max_tokens=5), SamplingParams(max_tokens=13)],
                     use_tqdm=False)
    assert eng.scheduler.is_finished() and not eng.scheduler.waiting


def test_control_channel_ring_order_backpressure_and_payloads():
    name = f"nvl_test_chan_{os.getpid()}"
    tx = _Channel(name, 4096, world=3, create=True)
    rxs = [_Channel(name, 4096, world=3, create=False) for _ in range(2)]
    got = {1: [], 2: []}
    N = 40

    def worker(rank, rx, delay):
        while True:
            op, args, body = rx.recv()
            if op == _Channel.OP_EXIT:
                rx.ack(rank)
                return
            got[rank].append((op, int(args[0]), int(args[1]), bytes(body)))
            time.sleep(delay)                               # slow consumer: rank 0 must wait, never overwrite
            rx.ack(rank)

    threads = [threading.Thread(target=worker, args=(1, rxs[0], 0.0)),
               threading.Thread(target=worker, args=(2, rxs[1], 0.002))]
    for t in threads:
        t.start()
    sent = []
    for k in range(N):
        payload = np.full(1 + (k * 37) % 4000, k % 251, dtype=np.uint8)
        tx.send(_Channel.OP_DECODE if k % 2 else _Channel.OP_PREFILL, (k, k * k), payload)
        sent.append((_Channel.OP_DECODE if k % 2 else _Channel.OP_PREFILL, k, k * k, payload.tobytes()))
        assert tx.sent - int(tx.cur[1:3].min()) <= _Channel.SLOTS
    tx.send(_Channel.OP_EXIT)
    for t in threads:
        t.join(30)
        assert not t.is_alive()
    assert got[1] == sent and got[2] == sent
    for rx in rxs:
        rx.close()
    tx.close()


# ---------------------------------------------------------------------------------------------------------------
# layers.decode_linear: which GEMM a decode-sized linear runs on (host logic; the kernels are stubbed)
def test_decode_gemm_chooser_policies(monkeypatch):
    """Skinny kernel by shape rule; else the wide-tile kernel according to NVL_GEMM_WIDE (0: never even planned,
    1: whenever the plan covers the shape, auto: timed once per shape outside a capture — inside a capture an
    untimed shape keeps the library GEMM and the decision is NOT cached); None = caller keeps the library GEMM."""
    from nano_vllm_amd import layers, ops
    calls = []
    monkeypatch.setattr(ops, "linear_decode_splits", lambda m, n, k, mode: 1 if k <= 1024 else 0)
    monkeypatch.setattr(ops, "linear_decode", lambda x, w, mode, out=None, packed=False: calls.append("skinny") or "skinny")
    monkeypatch.setattr(ops, "linear_wide_plan", lambda m, n, k, mode: calls.append("plan") or ((2, 64) if n != 48 else None))
    monkeypatch.setattr(ops, "linear_wide", lambda x, w, mode, out=None, workspace=None, packed=False:
                        calls.append("wide-packed" if packed else "wide") or "wide")
    monkeypatch.setattr(layers, "_scratch", lambda nbytes, device: None)
    monkeypatch.setattr(layers, "_wide_choice", {})
    x, shallow, deep, uncovered = torch.zeros(16, 1024), torch.zeros(64, 1024), torch.zeros(64, 4096), torch.zeros(48, 4096)
    xd = torch.zeros(16, 4096)

    assert layers.decode_linear(x, shallow, ops.LINEAR_BF16) == "skinny" and calls == ["skinny"]
    calls.clear()
    monkeypatch.setenv("NVL_GEMM_WIDE", "0")
    assert layers.decode_linear(xd, deep, ops.LINEAR_BF16) is None and calls == []          # never planned
    layers._wide_choice.clear()
    monkeypatch.setenv("NVL_GEMM_WIDE", "1")
    assert layers.decode_linear(xd, deep, ops.LINEAR_SILU) == "wide"
    assert layers.decode_linear(xd, uncovered, ops.LINEAR_BF16) is None                     # plan says no
    assert layers.wide_choices()[(16, 64, 4096, ops.LINEAR_SILU, None, False)] is True
    calls.clear()
    assert layers.decode_linear(xd, deep, ops.LINEAR_SILU) == "wide" and calls == ["plan", "wide"]   # decision cached
    calls.clear()                 # a module with a tile-packed copy streams THAT (its own cached decision)
    assert layers.decode_linear(xd, deep, ops.LINEAR_SILU, packed=deep.clone()) == "wide" and calls[-1] == "wide-packed"
    # auto: a deterministic rule (rows / matrix size), the same in every run
    layers._wide_choice.clear()
    monkeypatch.setenv("NVL_GEMM_WIDE", "auto")
    assert layers.decode_linear(xd, deep, ops.LINEAR_BF16) == "wide"                        # 16 rows: one row group
    x200 = torch.zeros(200, 4096)
    assert layers.decode_linear(x200, deep, ops.LINEAR_BF16) == "wide"                      # two row groups, small matrix
    huge = torch.zeros(20480, 4096)                                                         # 168 MB gate|up
    x176 = torch.zeros(176, 4096)
    assert layers.decode_linear(x176, huge, ops.LINEAR_SILU) is None                        # 145-192 rows: library GEMM
    assert layers.decode_linear(x200, huge, ops.LINEAR_SILU) == "wide"                      # 193-256 rows: 64-column k step
    assert layers.decode_linear(x200, torch.zeros(40960, 4096), ops.LINEAR_SILU) is None    # 336 MB gate|up: library GEMM
    assert layers.decode_linear(x200, huge, ops.LINEAR_BF16) == "wide"
    # tune: timed outside a capture, deferred (not cached) inside one
    layers._wide_choice.clear()
    monkeypatch.setenv("NVL_GEMM_WIDE", "tune")
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    assert layers.decode_linear(xd, deep, ops.LINEAR_PARTIAL) is None and layers.wide_choices() == {}
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    times = iter([1.0, 2.0])                                                               # wide 1.0 ms, library 2.0 ms
    monkeypatch.setattr(layers, "_time_cold", lambda fn, device, reps=3: next(times))
    assert layers.decode_linear(xd, deep, ops.LINEAR_BF16) == "wide"
    times = iter([2.0, 1.0])
    assert layers.decode_linear(xd, deep, ops.LINEAR_SILU) is None                          # library GEMM is faster
    assert layers.wide_choices() == {(16, 64, 4096, ops.LINEAR_BF16, None, False): True,
                                     (16, 64, 4096, ops.LINEAR_SILU, None, False): False}


# ---------------------------------------------------------------------------------------------------------------
# bench.py: the Qwen3-32B TP extra of a multi-GPU run is a child job; its outcome can never sink the primary line
def test_bench_tp_extra_child_job_outcomes(monkeypatch):
    import json as _json
    import subprocess
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    args = types.SimpleNamespace(num_seqs=48, gpu_memory_utilization=0.3, num_kvcache_blocks=300, kv_cache_dtype="bf16",
                                 eager=False)
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(empty_cache=lambda: None))
    seen = {}

    def fake_run(cmd, env, capture_output, text, timeout):
        seen.update(cmd=cmd, env=env, timeout=timeout)
        return subprocess.CompletedProcess(cmd, 0, stdout='noise\n{"metric": "x", "value": 1.5}\n', stderr="")

    monkeypatch.setenv("MASTER_PORT", "29533")
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    monkeypatch.setenv("NVL_BENCH_TP_EXTRA_TIMEOUT", "77")
    monkeypatch.setattr(subprocess, "run", fake_run)
    r = bench.tp_extra(args, fake_torch, None, 0, 4, {})
    assert r == {"metric": "x", "value": 1.5}
    assert seen["env"]["MASTER_PORT"] == "29634" and "TORCHELASTIC_USE_AGENT_STORE" not in seen["env"] and seen["timeout"] == 77.0
    cmd = seen["cmd"]
    assert cmd[1].endswith("bench.py") and cmd[cmd.index("--tp") + 1] == "4" and cmd[cmd.index("--model") + 1] == "qwen3-32b"
    assert "--no-tp-extra" in cmd and "--no-cpu-baseline" in cmd and "--eager" not in cmd
    assert bench.tp_extra(args, fake_torch, None, 1, 4, {}) is None                           # only rank 0 reports
    # a crashed child, a child without a JSON line, a timeout: reported as an error dict on rank 0
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: subprocess.CompletedProcess(a, -11, stdout="", stderr="boom"))
    r = bench.tp_extra(args, fake_torch, None, 0, 4, {})
    assert r["error"] == "child exit code -11" and r["stderr_tail"] == "boom"

    def timing_out(cmd, **k):
        raise subprocess.TimeoutExpired(cmd, k["timeout"])
    monkeypatch.setattr(subprocess, "run", timing_out)
    assert "timed out" in bench.tp_extra(args, fake_torch, None, 0, 4, {})["error"]
    assert bench.tp_extra(args, fake_torch, None, 2, 4, {}) is None
    _json.dumps(r)


def test_bench_tp_run_falls_back_to_the_process_group_and_never_reports_null():
    """bench.py's tensor-parallel runs (round-4 review item 1d): a pass whose xGMI P2P collectives latched a spin timeout
    is re-run with the process group's collectives on EVERY rank, and the line carries both attempts."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    calls = []

    def make_attempt(rank, latch_on):
        def attempt(p2p):
            calls.append((rank, p2p))
            if rank > 0:
                return None, (p2p and rank in latch_on)
            line = {"value": 100.0 if p2p else 80.0, "ms_per_step": 1.0,
                    "config": {"parallelism": "tp2", "p2p_status": "ok", "p2p_handoff": "lean"}}
            if p2p and 0 in latch_on:
                line["config"]["p2p_status"] = "NvlError('... spin limit ...')"
            return line, (p2p and 0 in latch_on)
        return attempt

    # nothing latched: one attempt, the P2P line is the result
    r = bench.with_p2p_fallback(make_attempt(0, set()), agree=lambda f: f)
    assert r["value"] == 100.0 and "tp_p2p_attempt" not in r and calls == [(0, True)]
    # a latch on ANOTHER rank (rank 0 saw nothing): `agree` is the OR over ranks, rank 0 re-runs too
    calls.clear()
    r = bench.with_p2p_fallback(make_attempt(0, {1}), agree=lambda f: True)
    assert calls == [(0, True), (0, False)]
    assert r["value"] == 80.0 and r["tp_p2p_attempt"]["value_invalid"] == 100.0 and "fallback" in r["config"]["parallelism"]
    # rank 0 latched: status travels into the record; a worker rank returns None from both attempts
    calls.clear()
    r = bench.with_p2p_fallback(make_attempt(0, {0}), agree=lambda f: f)
    assert r["value"] == 80.0 and "spin limit" in r["tp_p2p_attempt"]["p2p_status"]
    assert bench.with_p2p_fallback(make_attempt(1, {1}), agree=lambda f: f) is None and calls[-2:] == [(1, True), (1, False)]
    assert r["value"] is not None

    class NvlError(RuntimeError):
        pass
    assert bench._is_latched(NvlError("nvl error -2: nvl_allreduce: a peer did not arrive within the spin limit"))
    assert not bench._is_latched(NvlError("something else")) and not bench._is_latched(RuntimeError("spin limit"))
    # the latch as step() / generate() raise it since round 6 (engine/runner.py: _raise_if_collective_timed_out) — the wording
    # is taken from the runner's source, so that a rephrasing there cannot silently disable the re-run
    import inspect
    from nano_vllm_amd.engine import runner as runner_mod
    src = inspect.getsource(runner_mod.ModelRunner._raise_if_collective_timed_out)
    assert "gave up waiting for a peer" in src
    assert bench._is_latched(NvlError("a tensor-parallel P2P collective gave up waiting for a peer during a decode step: ..."))
    # ... and an attempt aborted INSIDE generate() (no number at all) still ends in the re-run's value
    calls.clear()

    def aborted(p2p):
        calls.append(p2p)
        if p2p:
            return {"value": None, "ms_per_step": None, "config": {"p2p_status": "NvlError('... gave up waiting for a peer ...')"}}, True
        return {"value": 80.0, "ms_per_step": 1.0, "config": {"parallelism": "tp2", "p2p_status": "n/a (process group)"}}, False
    r = bench.with_p2p_fallback(aborted, agree=lambda f: f)
    assert calls == [True, False] and r["value"] == 80.0 and r["tp_p2p_attempt"]["value_invalid"] is None
    assert "gave up waiting" in r["tp_p2p_attempt"]["p2p_status"]


def test_bench_line_stays_compact(monkeypatch, tmp_path):
    """The driver's record keeps only the tail of stdout: the one JSON line must stay well under 10 KB with all six extras
    attached (round-4 review item 3) — the children's full lines and the prose notes live in side files."""
    import json as _json
    import subprocess
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    child = {"metric": "m" * 60, "value": 17791.527986626046, "ms_per_step": 7529.76, "config": {"kv_blocks": 1234, "x": "y" * 900},
             "roofline": {"frac": 0.3211111, "decode_step_frac_of_8TBps": 0.2351111, "kernel": "k" * 200},
             "roofline_prefill": {"achieved": 447.3, "note": "n" * 300}}
    monkeypatch.setattr(subprocess, "run", lambda cmd, **k: subprocess.CompletedProcess(cmd, 0, stdout=_json.dumps(child) + "\n", stderr=""))
    monkeypatch.setattr(bench, "SIDE_DIR", str(tmp_path))
    monkeypatch.setattr(bench, "BENCH_T0", __import__("time").perf_counter())
    args = types.SimpleNamespace(gpu_memory_utilization=0.9)
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(empty_cache=lambda: None))
    ex = bench.extra_configs(args, fake_torch)
    assert set(ex) == {"config3", "config4_anchor", "config5", "tp8_rank_shape_bench", "tp4_rank_shape_bench",
                       "tp8_rank_shape_16k_prompts"}
    assert ex["config4_anchor"] == {"value": 17791.5, "ms": 7529.8, "attn": 0.321, "step": 0.235, "pf_TF": 447, "kv": 1234,
                                    "wall": ex["config4_anchor"]["wall"]}
    assert len(_json.dumps(ex)) < 1500
    full = _json.load(open(tmp_path / "bench_extras_full.json"))
    assert full["config3"]["config"]["x"] == "y" * 900                        # nothing is lost, it just is not on stdout
    # an exhausted wall budget skips the remaining extras and says so
    monkeypatch.setattr(bench, "BENCH_T0", __import__("time").perf_counter() - 10_000)
    assert all("skipped" in v["error"] for v in bench.extra_configs(args, fake_torch).values())
    # an extra is never STARTED with less of the 262 s budget left than it is known to need (the driver's 20 + 5 passes
    # leave ~100 s: BASELINE's own configs come first, then the rank shapes). Fake children take no time: with 42 s
    # left, the 48 s and 44 s ones are skipped and say so, the others run
    monkeypatch.setattr(bench, "BENCH_T0", __import__("time").perf_counter() - 220)
    ex = bench.extra_configs(args, fake_torch)
    assert list(ex) == ["config3", "config4_anchor", "config5", "tp8_rank_shape_bench", "tp4_rank_shape_bench",
                        "tp8_rank_shape_16k_prompts"]
    assert [k for k, v in ex.items() if "error" in v] == ["config4_anchor", "tp8_rank_shape_16k_prompts"]
    assert "needs ~48 s" in ex["config4_anchor"]["error"] and "262 s wall budget" in ex["config4_anchor"]["error"]
    assert len(_json.dumps(bench.NOTES)) < 6000 and bench.write_notes({}).startswith("gpurun_out/")


# ---------------------------------------------------------------------------------------------------------------
# ops.P2PComm construction is collective: a rank that fails locally must still join the exchange and the barrier
class _FakeCommLib:
    def __init__(self, fail_create=False, fail_connect=False):
        self.fail_create, self.fail_connect = fail_create, fail_connect
        self.calls = []

    def nvl_last_error(self):
        return b"fake failure"

    def nvl_allreduce_create(self, rank, world, max_bytes, href):
        self.calls.append("create")
        return -4 if self.fail_create else 0

    def nvl_allreduce_uid(self, h, uid):
        self.calls.append("uid")
        uid.raw = bytes([7]) * 64
        return 0

    def nvl_allreduce_connect(self, h, blob):
        self.calls.append("connect")
        return -4 if self.fail_connect else 0

    def nvl_allreduce_max_bytes(self, h):
        return 1 << 20

    def nvl_allreduce_set_fences(self, h, on):
        self.calls.append(f"fences={on}")
        return 0

    def nvl_allreduce_destroy(self, h):
        self.calls.append("destroy")
        return 0


@pytest.mark.parametrize("mode", ["ok", "create_fails", "connect_fails", "peer_failed"])
def test_p2p_comm_construction_always_joins_its_collectives(mode, monkeypatch):
    from nano_vllm_amd import ops
    fake = _FakeCommLib(fail_create=mode == "create_fails", fail_connect=mode == "connect_fails")
    monkeypatch.setattr(ops, "lib", lambda: fake)
    joined = []

    def exchange(blob):
        joined.append(("exchange", blob))
        peer = b"\0" * 64 if mode == "peer_failed" else bytes([9]) * 64
        return [blob, peer]

    def barrier():
        joined.append(("barrier",))

    if mode == "ok":
        c = ops.P2PComm(0, 2, 1 << 20, exchange, barrier)
        assert c.max_bytes == 1 << 20 and fake.calls == ["create", "uid", "connect", "fences=1"]   # fenced until validated
        assert c.fits(16, 1024) and not c.fits(16, 1028) and c.handoff == "fenced"
        c.set_handoff("lean")
        assert fake.calls[-1] == "fences=0" and c.handoff == "lean"
        c.close()
        assert fake.calls[-1] == "destroy"
    else:
        with pytest.raises(ops.NvlError):
            ops.P2PComm(0, 2, 1 << 20, exchange, barrier)
        assert "fences=0" not in fake.calls and "fences=1" not in fake.calls
        if mode == "create_fails":
            assert joined[0] == ("exchange", b"\0" * 64)          # an all-zero token tells the peers
            assert "connect" not in fake.calls
    assert [j[0] for j in joined] == ["exchange", "barrier"]      # both collectives, exactly once, in every outcome
