"""The token-level acceptance rule of SURVEY.md §8c(3) (TEST INFRASTRUCTURE — see oracle/__init__.py).

The reference is not bit-identical to itself: its eager and its shipped (@torch.compile) forms round at different
points (layers/layernorm.py:16,28, activation.py:8), so two legitimate bf16 executions of the same model differ in
their logits by a measurable FLOOR (max 0.06 abs = 0.0195 x absmax on Qwen3-0.6B-shaped random
weights). Greedy decoding therefore cannot be compared free-running; instead the oracle is TEACHER-FORCED with the
product's tokens and every decision of the product is judged against the oracle's logits for the same history:

  * a row is DECISIVE when the oracle's top-1 / top-2 margin exceeds 2 x floor: the product's token must then be the
    oracle's argmax, exactly (both executions may be off by one floor in opposite directions and still agree);
  * on the other rows (near-ties below the noise) the product may pick another token, but only one whose oracle logit
    lies within 2 x floor of the oracle's maximum.

The floor is MEASURED for the run at hand: the same history is evaluated by a second oracle with the reference's
eager rounding (`compiled=False`), and floor = max over steps of max|logits_compiled - logits_eager| / absmax; the
per-step absolute floor scales it by that step's absmax. `SURVEY_FLOOR_REL` is printed next to it.

T > 0 (the path the headline bench runs: reference layers/sampler.py:8-12 behind model_runner.py:214-220, bench.py:18
T = 0.6). The reference's token is `argmax_i softmax(l/T)_i / E_i` = `argmax_i (l_i/T - log E_i)`: the same rule
applies to the race KEYS instead of the logits. The product's exponentials are a counter-based function of (seed,
request ordinal, token position, column) — oracle/philox.py restates it — so a row's keys can be rebuilt exactly on
the oracle's logits: a logit error of one floor moves a key by floor / T, hence a row is decisive when the top-1 /
top-2 gap OF THE KEYS exceeds 2 x floor / T (the product's token must then be the keys' argmax, exactly), and a
sub-margin row may pick any token whose key lies within 2 x floor / T of the maximum. `KEY_EPS` absorbs the
difference between the kernel's hardware log and numpy's (both sides evaluate `l/T - log E` in fp32).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch

SURVEY_FLOOR_REL = 0.0195          # SURVEY.md §8c(2), measured on Qwen3-0.6B-shaped random weights
KEY_EPS = 2e-4                     # fp32 evaluation noise of l/T - log E (v_log_f32 vs libm; keys are O(10))


@dataclass
class Verdict:
    floor_rel: float = 0.0                     # measured reference-eager vs reference-compiled floor of this run
    rows: int = 0
    exact: int = 0                             # product token == oracle argmax
    decisive: int = 0                          # rows with margin > 2 x floor
    decisive_exact: int = 0
    worst_gap: float = 0.0                     # max over rows of (oracle max logit - oracle logit of the product's token)
    worst_gap_in_floors: float = 0.0           # the same in units of that step's absolute floor
    worst_logit_err_rel: float = 0.0           # max|product logits - oracle logits| / absmax, when product logits given
    sampled_rows: int = 0                      # rows judged on race keys (T > 0)
    violations: list = field(default_factory=list)

    def ok(self) -> bool:
        return not self.violations

    def line(self, name: str) -> str:
        what = "the oracle's argmax" if not self.sampled_rows else (
            f"the argmax of the oracle's race keys l/T - log E ({self.sampled_rows} rows at T > 0, draws replayed)")
        return (f"{name}: {self.exact}/{self.rows} tokens are {what}; decisive rows (margin > 2 x floor) "
                f"{self.decisive_exact}/{self.decisive} exact; worst gap {self.worst_gap:.4f} = "
                f"{self.worst_gap_in_floors:.2f} floors; measured floor {self.floor_rel:.5f} x absmax "
                f"(SURVEY constant {SURVEY_FLOOR_REL}); {len(self.violations)} violations")


class Judge:
    """Accumulates one run: `add_step` per engine step, then `verdict()`. Keeps per-row statistics only (no logits)."""

    def __init__(self):
        self.floor_rel = 0.0
        self._rows: list[tuple] = []           # (step, row, token, gap, margin, absmax of the step, 1/T or 1, eps)
        self._err = 0.0
        self._step = 0

    def add_step(self, logits: torch.Tensor, tokens, logits_eager: torch.Tensor | None = None,
                 ours: torch.Tensor | None = None, skip_rows=(), draws=None) -> None:
        """logits: the oracle's (compiled rounding, teacher-forced with the product's `tokens`) [rows, V];
        logits_eager: the same history through the oracle with the reference's eager rounding (feeds the floor);
        ours: the product's own logits (optional, feeds worst_logit_err_rel);
        skip_rows: rows whose token the engine discards (mid-prefill chunks, scheduler.py:86-87);
        draws: None (greedy run), or per row None (a T = 0 row) / (temperature, seed, request ordinal, position of
        the token being drawn): that row is judged on its race keys `l/T - log E`, E replayed by oracle/philox.py."""
        logits = logits.float().cpu()
        absmax = max(float(logits.abs().max()), 1e-20)
        if logits_eager is not None:
            self.floor_rel = max(self.floor_rel, float((logits - logits_eager.float().cpu()).abs().max()) / absmax)
        if ours is not None:
            self._err = max(self._err, float((ours.float().cpu() - logits).abs().max()) / absmax)
        top2 = logits.topk(2, dim=-1).values
        for row, tok in enumerate(tokens):
            if row in skip_rows:
                continue
            d = draws[row] if draws is not None else None
            if d is None:
                self._rows.append((self._step, row, int(tok), float(top2[row, 0] - logits[row, tok]),
                                   float(top2[row, 0] - top2[row, 1]), absmax, 1.0, 0.0))
                continue
            from .philox import race_keys
            temp, seed, ordinal, position = d
            keys = race_keys(logits[row].numpy(), temp, seed, ordinal, position)
            first, second = np.partition(keys, -2)[-2:][::-1]
            self._rows.append((self._step, row, int(tok), float(first - keys[int(tok)]), float(first - second), absmax,
                               1.0 / temp, KEY_EPS))
        self._step += 1

    def verdict(self, floor_rel: float | None = None) -> Verdict:
        floor_rel = self.floor_rel if floor_rel is None else floor_rel
        v = Verdict(floor_rel=floor_rel, worst_logit_err_rel=self._err)
        for step, row, tok, gap, margin, absmax, scale, eps in self._rows:
            floor_abs = floor_rel * absmax * scale           # one floor of logit error moves a race key by floor / T
            decisive = margin > 2 * floor_abs + eps
            v.rows += 1
            v.sampled_rows += eps > 0.0
            v.exact += gap == 0.0
            v.decisive += decisive
            v.decisive_exact += decisive and gap == 0.0
            v.worst_gap = max(v.worst_gap, gap)
            v.worst_gap_in_floors = max(v.worst_gap_in_floors, gap / max(floor_abs, 1e-20))
            if (decisive and gap != 0.0) or gap > 2 * floor_abs + eps:
                v.violations.append(dict(step=step, row=row, token=tok, gap=gap, margin=margin, floor_abs=floor_abs))
        return v


def judge_run(cfg: dict, weights: dict, prompts, max_tokens, rec: list[dict], num_blocks: int, device=None,
              temperatures=None, seed: int = 0, floor_rel: float | None = None, on_step=None, ordinal_base: int = 0,
              block_size: int = 256, **sched_kw) -> Verdict:
    """Judge a recorded product run. `rec`: per engine step {"prefill": bool, "seq_ids": [...], "tables": [[...]],
    "tokens": [...], optional "logits": the product's logits}. Two oracle engines (compiled and eager rounding) are
    teacher-forced with the product's tokens; scheduling (phase, batch composition, block tables) must be identical
    step for step; returns the Verdict under the margin rule with the floor measured on this run.
    `temperatures` (per prompt; None = greedy run) and the engine's `seed`: rows at T > 0 are judged on their race
    keys, the draw of request i (its ordinal in submission order, engine/llm_engine.py:43-47 `add_request` order) for
    the token at position p rebuilt by oracle/philox.py; `ordinal_base`: requests the engine had already taken before
    this run (the ordinal counts per engine, not per generate() call). `floor_rel`: skip the eager-rounding oracle and use this
    floor (bench.py's bounded sample: one oracle pass). `on_step(i, seconds)`: timing hook of that leg."""
    from time import perf_counter

    from .engine import OracleEngine
    from .model import OracleQwen3
    # (block_size: Config.kvcache_block_size, config.py:22 — any multiple of 256)
    engines = [OracleEngine(OracleQwen3(cfg, weights, compiled=c, device=device), num_blocks, block_size, **sched_kw)
               for c in ((True, False) if floor_rel is None else (True,))]
    temps = list(temperatures) if temperatures is not None else [0.0] * len(prompts)
    for eng in engines:
        eng.keep_logits = True
        for p, m in zip(prompts, max_tokens):
            eng.add(p, 0.0, m, True)             # (teacher-forced: the oracle's own sampling is never used)
    j = Judge()
    base = None
    for i, r in enumerate(rec):
        for eng in engines:
            t0 = perf_counter()
            eng.step(forced_tokens=r["tokens"])
            if on_step is not None and eng is engines[0]:
                on_step(i, perf_counter() - t0)
        o, e = engines[0].trace[-1], engines[-1].trace[-1]
        if base is None:
            base = r["seq_ids"][0] - o["seq_ids"][0]
        assert o["is_prefill"] == r["prefill"], f"step {i}: phase differs"
        assert [s + base for s in o["seq_ids"]] == r["seq_ids"], f"step {i}: batch composition differs"
        assert o["tables"] == r["tables"], f"step {i}: block tables differ"
        # (the token of a mid-prefill chunk is discarded by the scheduler, scheduler.py:86-87, but it is still a decision
        # both sides computed from the same history: judged like any other row)
        # the draw key of a row: request ordinal = the oracle's sequence id (both count submissions from 0), position
        # = index the sampled token will occupy = cached + scheduled tokens (prefill: model_runner.py:136-141,
        # decode: len(seq), :176-179)
        draws = None
        if any(t > 0 for t in temps):
            draws = [None if temps[s] <= 0 else (temps[s], seed, ordinal_base + s, c + n)
                     for s, c, n in zip(o["seq_ids"], o["cached"], o["sched"])]
        j.add_step(o["logits"], r["tokens"], e["logits"] if len(engines) > 1 else None, ours=r.get("logits"),
                   draws=draws)
        o.pop("logits"), e.pop("logits", None)
    assert not engines[0].waiting and not engines[0].running
    return j.verdict(floor_rel)
