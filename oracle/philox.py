"""The draw of the product's sampler, restated independently (TEST INFRASTRUCTURE — see oracle/__init__.py).

The reference draws its exponentials from torch's generator (layers/sampler.py:11 `exponential_(1)`), consumed in
batch order: that stream is not even the same between the reference's eager and compiled forms (SURVEY.md §8c(4):
`[510, 36]` vs `[350, 192]` on one seed), so "the reference's random numbers" do not exist as a spec. The product
replaces the generator by a COUNTER-BASED draw that is a pure function of (seed, request ordinal, token position,
vocabulary column) — which makes an exact judgement at T > 0 possible: given the oracle's logits for the same history,
the sampled token must be `argmax_i(l_i / T - log E_i)` (sampler.py:8-12 in log space: the softmax normaliser is common
to the row) with THESE E_i. This file restates that draw in numpy from its written specification, not by calling the
product:

  Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; multipliers 0xD2511F53 /
  0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85), one call per four vocabulary columns:
      counter = (c >> 2  [low 32 bits],  (c >> 34) ^ (position << 8  [low 32 bits]),  ordinal,  position >> 24)
      key     = (seed [low 32], seed [high 32])
      bits    = output word  c & 3
      u       = ((bits >> 8) + 0.5) / 2^24                (24 random bits, centred, never 0 or 1)
      E       = max(-ln u, 1e-10)                          (sampler.py:11 `clamp_min_(1e-10)`)

`tests/test_judge.py` checks it against the published Philox4x32-10 known-answer vectors and against the product
library's own host replay (`nvl_sample_exponentials_host`) — two independent implementations of one specification.
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Ten rounds of Philox-4x32 on arrays of 32-bit counter words (any broadcastable shapes); returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = (p1 >> _S32) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> _S32) ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, p1 & _MASK, n2, p0 & _MASK
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def exponentials(seed: int, ordinal: int, position: int, vocab: int, col0: int = 0) -> np.ndarray:
    """E[j] for global vocabulary columns col0 .. col0 + vocab - 1 of the draw made for request `ordinal` when it
    samples the token that will sit at `position` (fp32 [vocab])."""
    cols = np.arange(col0, col0 + vocab, dtype=np.uint64)
    ctr = cols >> np.uint64(2)
    off = int(position)
    words = philox4x32_10(ctr & _MASK, (ctr >> _S32) ^ np.uint64((off << 8) & 0xFFFFFFFF), np.uint64(ordinal & 0xFFFFFFFF),
                          np.uint64((off >> 24) & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    bits = np.choose((cols & np.uint64(3)).astype(np.intp), words)
    u = ((bits >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    return np.maximum(-np.log(u, dtype=np.float32), np.float32(1e-10))


def race_keys(logits: np.ndarray, temperature: float, seed: int, ordinal: int, position: int, col0: int = 0) -> np.ndarray:
    """`l / T - log E` for one row (fp32): the quantity whose argmax is the sampled token (sampler.py:8-12 in log
    space). `logits` fp32 [V]."""
    e = exponentials(seed, ordinal, position, logits.shape[0], col0)
    return logits.astype(np.float32) / np.float32(temperature) - np.log(e, dtype=np.float32)
