"""Drive a scheduler + block manager through a seeded synthetic workload with a FAKE runner
(tokens are a hash of (sequence, position)) and record what every step scheduled.

Used three ways (TEST INFRASTRUCTURE):
  * on the imported reference's Scheduler/Sequence (oracle/make_golden.py -> tests/golden/),
  * on the oracle's restatement (oracle/engine.py),
  * on the product's array-based scheduler (nano_vllm_amd.engine.sched),
and the traces must be identical step for step.
"""
from __future__ import annotations

import random
from types import SimpleNamespace

SCENARIOS = {
    # name: (seed, n_seqs, num_blocks, max_num_seqs, max_num_batched_tokens, shared_prefix_len)
    "plain": (1, 24, 64, 8, 2048, 0),
    "prefix_cache": (2, 32, 96, 16, 4096, 600),
    "preempt": (3, 20, 14, 8, 1024, 0),
    "chunked": (4, 6, 64, 4, 700, 300),
    "tight_prefix_preempt": (5, 28, 18, 6, 1536, 520),
}
EOS = 7


def make_workload(name: str):
    seed, n, num_blocks, max_seqs, max_tok, shared = SCENARIOS[name]
    rng = random.Random(seed)
    prefix = [rng.randrange(8, 5000) for _ in range(shared)]
    reqs = []
    for i in range(n):
        plen = rng.randrange(1, 900)
        body = [rng.randrange(8, 5000) for _ in range(plen)]
        use_prefix = shared and rng.random() < 0.7
        prompt = (prefix + body) if use_prefix else body
        reqs.append(dict(prompt=prompt, max_tokens=rng.randrange(1, 400), ignore_eos=rng.random() < 0.5,
                         arrive=rng.randrange(0, 40)))
    cfg = SimpleNamespace(max_num_seqs=max_seqs, max_num_batched_tokens=max_tok, eos=EOS, kvcache_block_size=256,
                          num_kvcache_blocks=num_blocks)
    return cfg, reqs


def fake_token(seq_index: int, position: int) -> int:
    """Deterministic stand-in for the model: depends only on (sequence, position)."""
    x = (seq_index * 1000003 + position * 7919 + 12345) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0x5BD1E995) & 0xFFFFFFFF
    x ^= x >> 15
    return EOS if x % 97 == 0 else 8 + x % 4000


def run_trace(name: str, make_scheduler, make_sequence, max_steps: int = 20000) -> list[dict]:
    """make_scheduler(cfg) -> object with add/schedule/postprocess/is_finished;
    make_sequence(prompt, max_tokens, ignore_eos) -> sequence object with the reference's attributes."""
    cfg, reqs = make_workload(name)
    sched = make_scheduler(cfg)
    order = sorted(range(len(reqs)), key=lambda i: (reqs[i]["arrive"], i))
    seqs, index_of = [], {}
    trace = []
    step = 0
    pending = list(order)
    while True:
        while pending and reqs[pending[0]]["arrive"] <= step:
            i = pending.pop(0)
            r = reqs[i]
            s = make_sequence(r["prompt"], r["max_tokens"], r["ignore_eos"])
            index_of[id(s)] = len(seqs)
            seqs.append(s)
            sched.add(s)
        if sched.is_finished():
            if not pending:
                break
            step = reqs[pending[0]]["arrive"]
            continue
        batch, is_prefill = sched.schedule()
        idx = [index_of[id(s)] for s in batch]
        tokens = [fake_token(i, len(s)) for i, s in zip(idx, batch)]
        trace.append(dict(prefill=bool(is_prefill), seqs=idx, sched=[s.num_scheduled_tokens for s in batch],
                          cached=[s.num_cached_tokens for s in batch], tables=[list(s.block_table) for s in batch]))
        sched.postprocess(batch, tokens, is_prefill)
        step += 1
        assert step < max_steps
    trace.append(dict(final=[list(s.token_ids[s.num_prompt_tokens:]) for s in seqs]))
    return trace
