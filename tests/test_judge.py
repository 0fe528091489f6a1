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
