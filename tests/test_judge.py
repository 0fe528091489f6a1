"""The token acceptance rule (oracle/judge.py) on CPU: synthetic cases of the rule itself, and the
reference-vs-reference consistency check — a free-running run of the oracle with the reference's EAGER rounding must
pass the gate when judged by the oracle with the COMPILED rounding (the floor is by definition the noise between those
two), and a run with a deliberately wrong token on a decisive row must fail it."""
import tempfile

import pytest
import torch

from oracle.judge import Judge, judge_run


def _logits(rows):
    return torch.tensor(rows, dtype=torch.float32)


def test_rule_on_synthetic_rows():
    lg = _logits([[5.0, 1.0, 0.0, -1.0],         # decisive (margin 4)
                  [2.00, 1.99, 0.0, -1.0]])      # near-tie (margin 0.01)
    eager = lg + 0.02                             # floor = 0.02 / 5 = 0.004 x absmax => floor_abs 0.02, 2 floors = 0.04
    j = Judge()
    j.add_step(lg, [0, 1], eager)                # exact on the decisive row, the runner-up on the near-tie: allowed
    v = j.verdict()
    assert v.ok() and v.rows == 2 and v.exact == 1 and v.decisive == 1 and v.decisive_exact == 1
    assert abs(v.floor_rel - 0.004) < 1e-6 and abs(v.worst_gap - 0.01) < 1e-6
    j = Judge()
    j.add_step(lg, [1, 0], eager)                # wrong token on the decisive row
    v = j.verdict()
    assert not v.ok() and v.violations[0]["row"] == 0 and v.violations[0]["gap"] == 4.0
    j = Judge()
    j.add_step(lg, [0, 2], eager)                # near-tie row, but a token 2.0 below the maximum: outside 2 floors
    assert not j.verdict().ok()


@pytest.fixture(scope="module")
def tiny():
    from nano_vllm_amd.weights import write_synthetic_checkpoint
    from oracle.model import load_weights
    path = tempfile.mkdtemp(prefix="qwen3tiny_")
    write_synthetic_checkpoint(path, "qwen3-tiny", seed=0, vocab_size=512, max_position_embeddings=2048)
    return load_weights(path)


def _free_run(cfg, w, prompts, max_tokens, compiled, nblk=16):
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3
    eng = OracleEngine(OracleQwen3(cfg, w, compiled=compiled), nblk, 256, max_num_seqs=8)
    for p, m in zip(prompts, max_tokens):
        eng.add(p, 0.0, m, True)
    rec = []
    while eng.waiting or eng.running:
        eng.step()
        t = eng.trace[-1]
        rec.append(dict(prefill=t["is_prefill"], seq_ids=list(t["seq_ids"]), tables=[list(x) for x in t["tables"]],
                        tokens=list(t["tokens"])))
    return rec


def test_reference_eager_run_passes_the_gate_of_the_compiled_reference(tiny):
    cfg, w = tiny
    g = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, 512, (n,), generator=g).tolist() for n in (40, 300, 7, 129)]
    max_tokens = [10, 8, 12, 6]
    rec = _free_run(cfg, w, prompts, max_tokens, compiled=False)
    v = judge_run(cfg, w, prompts, max_tokens, rec, 16, max_num_seqs=8)
    print(v.line("oracle(eager) judged by oracle(compiled)"))
    assert v.ok() and v.rows == sum(max_tokens) and v.floor_rel > 0
    # a compiled free run judged by itself: every token exact
    rec = _free_run(cfg, w, prompts, max_tokens, compiled=True)
    v = judge_run(cfg, w, prompts, max_tokens, rec, 16, max_num_seqs=8)
    assert v.ok() and v.exact == v.rows
    # flip one token on the most decisive decode row: the gate must catch it
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3
    eng = OracleEngine(OracleQwen3(cfg, w, compiled=True), 16, 256, max_num_seqs=8)
    eng.keep_logits = True
    for p, m in zip(prompts, max_tokens):
        eng.add(p, 0.0, m, True)
    best = None
    for i, r in enumerate(rec):
        eng.step(forced_tokens=r["tokens"])
        for row, mg in enumerate(eng.trace[-1]["margin"]):
            if i == len(rec) - 1 and (best is None or mg > best[0]):      # last step: the flip changes no later history
                best = (mg, i, row, int(eng.trace[-1]["logits"][row].argmin()))
    _, i, row, wrong = best
    bad = [dict(r, tokens=list(r["tokens"])) for r in rec]
    bad[i]["tokens"][row] = wrong
    v = judge_run(cfg, w, prompts, max_tokens, bad, 16, max_num_seqs=8)
    assert not v.ok() and v.violations[0]["step"] == i and v.violations[0]["row"] == row


# ---------------------------------------------------------------------------------------------------------------
# T > 0: the draw restated in oracle/philox.py, and the race-key form of the rule
def test_philox_known_answers_and_the_product_library_replay_agree():
    """oracle/philox.py against (a) the Random123 known-answer vectors of Philox4x32-10 and (b) the product library's
    host replay of its own draw (nvl_sample_exponentials_host: host code of libnvl_hip.so, no GPU needed) — two
    independent implementations of the keying written down in oracle/philox.py's header."""
    import numpy as np
    from nano_vllm_amd import ops
    from oracle.philox import exponentials, philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in philox4x32_10(*ctr, *key)) == want
    for seed, ordinal, position, col0, n in [(1234, 5, 77, 0, 4099), ((7 << 32) | 99, 0xfffffff0, (3 << 24) | 12345,
                                                                     151936 - 64, 64), (0, 0, 0, 8, 151936 // 8)]:
        theirs = ops.sample_exponentials_host(seed, position, ordinal, col0, n)
        mine = exponentials(seed, ordinal, position, n, col0)
        assert np.abs(theirs - mine).max() <= 2e-6 * max(1.0, float(theirs.max()))     # libm vs numpy log: last bit
        assert theirs.min() >= 1e-10 and mine.min() >= 1e-10


def test_race_key_rule_on_synthetic_rows():
    import numpy as np
    from oracle.philox import race_keys
    g = torch.Generator().manual_seed(3)
    lg = torch.randn(3, 4096, generator=g) * 3
    draws = [(0.6, 11, r, 40 + r) for r in range(3)]
    keys = [race_keys(lg[r].numpy(), *draws[r]) for r in range(3)]
    best = [int(np.argmax(k)) for k in keys]
    assert best != [int(x) for x in lg.argmax(-1)]                 # the draw matters: not the greedy tokens
    j = Judge()
    j.add_step(lg, best, lg + 1e-3, draws=draws)
    v = j.verdict()
    assert v.ok() and v.exact == 3 and v.sampled_rows == 3
    # the greedy token instead of the sampled one on a row whose key margin is decisive: caught
    j = Judge()
    j.add_step(lg, [int(lg[0].argmax())] + best[1:], lg + 1e-3, draws=draws)
    assert not j.verdict().ok()
    # a different position (= another draw) moves the winner: the key really is (ordinal, position)
    assert int(np.argmax(race_keys(lg[0].numpy(), 0.6, 11, 0, 41))) != best[0]
    # mixed batch: a T = 0 row next to sampled rows is judged on the logits
    j = Judge()
    j.add_step(lg, [int(lg[0].argmax())] + best[1:], lg + 1e-3, draws=[None] + draws[1:])
    v = j.verdict()
    assert v.ok() and v.sampled_rows == 2


def test_sampled_run_of_the_oracle_is_judged_exact_and_a_wrong_draw_is_caught(tiny):
    """A free run that samples with the replayed draws (temperature 0.6 and 1.3 next to a greedy request, chunk-free
    prefill + decode) passes `judge_run(..., temperatures=, seed=)` with every token exact; the same tokens judged under
    another seed fail."""
    import numpy as np
    from oracle.engine import OracleEngine
    from oracle.model import OracleQwen3
    from oracle.philox import race_keys
    cfg, w = tiny
    g = torch.Generator().manual_seed(8)
    prompts = [torch.randint(0, 512, (n,), generator=g).tolist() for n in (33, 260, 9)]
    max_tokens, temps, seed = [9, 7, 11], [0.6, 0.0, 1.3], 42
    eng = OracleEngine(OracleQwen3(cfg, w, compiled=True), 16, 256, max_num_seqs=8)
    for p, m in zip(prompts, max_tokens):
        eng.add(p, 0.0, m, True)

    def choose(t, logits):                    # sample with the replayed draw: request ordinal = id, position = cached + sched
        return [int(np.argmax(race_keys(logits[row].numpy(), temps[sid], seed, sid, c + n))) if temps[sid] > 0
                else int(logits[row].argmax()) for row, (sid, c, n) in enumerate(zip(t["seq_ids"], t["cached"], t["sched"]))]

    eng.choose = choose
    rec = []
    while eng.waiting or eng.running:
        eng.step()
        t = eng.trace[-1]
        rec.append(dict(prefill=t["is_prefill"], seq_ids=list(t["seq_ids"]), tables=[list(x) for x in t["tables"]],
                        tokens=list(t["tokens"])))
    v = judge_run(cfg, w, prompts, max_tokens, rec, 16, temperatures=temps, seed=seed, max_num_seqs=8)
    print(v.line("oracle sampled run judged by itself"))
    assert v.ok() and v.exact == v.rows == sum(max_tokens) and v.sampled_rows == 9 + 11
    assert not judge_run(cfg, w, prompts, max_tokens, rec, 16, temperatures=temps, seed=seed + 1, max_num_seqs=8).ok()


def test_restated_draw_is_column_addressable():
    """The draw is a function of the GLOBAL column: a vocabulary shard's exponentials (col0 > 0) are a slice of the full
    row's — what makes the vocab-parallel race merge to the TP = 1 token — and different (ordinal, position) pairs give
    different rows."""
    import numpy as np
    from oracle.philox import exponentials
    full = exponentials(99, 7, 1234, 4096)
    for col0, n in ((0, 512), (8, 1000), (2048, 2048), (4088, 8)):
        assert np.array_equal(exponentials(99, 7, 1234, n, col0), full[col0:col0 + n])
    assert not np.array_equal(exponentials(99, 8, 1234, 4096), full)
    assert not np.array_equal(exponentials(99, 7, 1235, 4096), full)
    assert not np.array_equal(exponentials(100, 7, 1234, 4096), full)
    # E ~ Exp(1): mean and variance of 4096 draws
    assert abs(float(full.mean()) - 1.0) < 0.08 and abs(float(full.var()) - 1.0) < 0.2
