"""Per-op CPU restatements (PyTorch fp32) of the reference's hot-path arithmetic.

TEST INFRASTRUCTURE (see oracle/__init__.py). File:line citations are into
GeeeekExplorer/nano-vllm v0.2.0.

Rounding convention. The reference decorates RMSNorm / rotary / SiluAndMul / Sampler with
@torch.compile unconditionally (layers/layernorm.py:16,28; rotary_embedding.py:37;
activation.py:8; sampler.py:7), and inductor keeps fp32 between fused ops, so each compiled
graph rounds to bf16 ONCE, at its output (`compiled=True`, the default here). Eager execution
of the same Python rounds at every `.to(bf16)` (`compiled=False`); both variants are checked
against the imported reference in tests/test_oracle_vs_reference.py.
"""
from __future__ import annotations

import math

import torch

BF16 = torch.bfloat16


# ----------------------------------------------------------------------------------------------
# layers/layernorm.py
def rms_forward(x: torch.Tensor, weight: torch.Tensor, eps: float, compiled: bool = True) -> torch.Tensor:
    """RMSNorm.rms_forward, layers/layernorm.py:16-26."""
    dt = x.dtype
    x32 = x.float()                                   # :22
    var = x32.pow(2).mean(dim=-1, keepdim=True)       # :23
    x32 = x32 * torch.rsqrt(var + eps)                # :24
    if compiled:
        return (x32 * weight.float()).to(dt)          # :25 fused: one rounding
    return x32.to(dt) * weight                        # :25 eager: round, multiply in bf16, round


def add_rms_forward(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                    compiled: bool = True) -> tuple[torch.Tensor, torch.Tensor]:
    """RMSNorm.add_rms_forward, layers/layernorm.py:28-40. Returns (y, new_residual)."""
    dt = x.dtype
    s = x.float() + residual.float()                  # :35
    new_residual = s.to(dt)                           # :36
    var = s.pow(2).mean(dim=-1, keepdim=True)         # :37  (un-rounded sum)
    s = s * torch.rsqrt(var + eps)                    # :38
    if compiled:
        return (s * weight.float()).to(dt), new_residual
    return s.to(dt) * weight, new_residual            # :39


# ----------------------------------------------------------------------------------------------
# layers/rotary_embedding.py
def rope_table(head_dim: int, max_position: int, base: float) -> torch.Tensor:
    """cos_sin_cache of RotaryEmbedding.__init__, layers/rotary_embedding.py:29-35, as
    fp32 [max_position, head_dim] (the reference keeps a singleton middle dim)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def apply_rotary(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb, layers/rotary_embedding.py:6-14 (neox half-split)."""
    x1, x2 = torch.chunk(x.float(), 2, dim=-1)
    y1 = x1 * cos - x2 * sin
    y2 = x2 * cos + x1 * sin
    return torch.cat((y1, y2), dim=-1).to(x.dtype)


def rotary_forward(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
                   table: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """RotaryEmbedding.forward, layers/rotary_embedding.py:37-48. q,k: [N, H, D]."""
    cos_sin = table[positions].unsqueeze(1)           # :44 (table[pos] -> [N,1,D])
    cos, sin = cos_sin.chunk(2, dim=-1)               # :45
    return apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)


# ----------------------------------------------------------------------------------------------
# layers/activation.py
def silu_and_mul(x: torch.Tensor, compiled: bool = True) -> torch.Tensor:
    """SiluAndMul.forward, layers/activation.py:8-11."""
    g, u = x.chunk(2, -1)
    if compiled:
        g32, u32 = g.float(), u.float()
        return (g32 * torch.sigmoid(g32) * u32).to(x.dtype)
    return torch.nn.functional.silu(g) * u


# ----------------------------------------------------------------------------------------------
# layers/attention.py (Triton KV store) — reference cache layout [num_blocks, block, Hkv, D]
def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                  slot_mapping: torch.Tensor) -> None:
    """store_kvcache_kernel, layers/attention.py:10-30: cache.view(-1, Hkv*D)[slot] = row,
    rows with slot == -1 skipped (:23)."""
    n, h, d = key.shape
    kc = k_cache.view(-1, h * d)
    vc = v_cache.view(-1, h * d)
    keep = slot_mapping >= 0
    slots = slot_mapping[keep].long()
    kc[slots] = key.reshape(n, h * d)[keep]
    vc[slots] = value.reshape(n, h * d)[keep]


def to_head_major(cache: torch.Tensor) -> torch.Tensor:
    """reference layout [nblk, block, Hkv, D] -> libnvl layout [nblk, Hkv, block, D]."""
    return cache.permute(0, 2, 1, 3).contiguous()


def from_head_major(cache: torch.Tensor) -> torch.Tensor:
    return cache.permute(0, 2, 1, 3).contiguous()


# ----------------------------------------------------------------------------------------------
# flash-attn entry points used at layers/attention.py:67-74 (package absent: restated from its
# documented semantics — fp32 scores/softmax, P cast to the input dtype before P.V, fp32
# accumulate, causal mask aligned to the bottom-right corner, GQA by head // (Hq/Hkv)).
def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, causal_offset: int | None,
            return_lse: bool = False):
    """q [Lq,Hq,D], k/v [Lk,Hkv,D] -> [Lq,Hq,D]. causal_offset = Lk - Lq (None: no mask).
    return_lse: also the fp32 log-sum-exp of the scaled scores [Lq, Hq] (flash-attn's softmax_lse)."""
    lq, hq, d = q.shape
    lk, hkv, _ = k.shape
    g = hq // hkv
    q32 = q.float().permute(1, 0, 2)                              # [Hq, Lq, D]
    k32 = k.float().permute(1, 0, 2).repeat_interleave(g, dim=0)  # [Hq, Lk, D]
    v32 = v.float().permute(1, 0, 2).repeat_interleave(g, dim=0)
    s = torch.matmul(q32, k32.transpose(1, 2)) * scale            # [Hq, Lq, Lk]
    if causal_offset is not None:
        i = torch.arange(lq, device=q.device).unsqueeze(1)
        j = torch.arange(lk, device=q.device).unsqueeze(0)
        s = s.masked_fill(j > i + causal_offset, float("-inf"))
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m)
    l = e.sum(dim=-1, keepdim=True)
    o = torch.matmul(e.to(q.dtype).float(), v32) / l              # P rounded to bf16 before P.V
    o = o.permute(1, 0, 2).to(q.dtype)
    if return_lse:
        return o, (m + torch.log(l)).squeeze(-1).transpose(0, 1).contiguous()
    return o


def _gather_paged(cache: torch.Tensor, table_row: torch.Tensor, length: int) -> torch.Tensor:
    """cache [nblk, block, Hkv, D] (reference layout), table_row int [max_blocks] -> [length, Hkv, D]."""
    block = cache.shape[1]
    nb = (length + block - 1) // block
    blocks = cache[table_row[:nb].long()]                         # [nb, block, Hkv, D]
    return blocks.reshape(nb * block, *cache.shape[2:])[:length]


def flash_attn_varlen_func(q, k, v, max_seqlen_q, cu_seqlens_q, max_seqlen_k, cu_seqlens_k,
                           softmax_scale, causal=True, block_table=None, return_softmax_lse=False):
    """Varlen attention as called at layers/attention.py:67-70. With block_table, k/v are the
    paged caches (reference layout) and sequence s reads keys cache[block_table[s, t//B], t%B].
    return_softmax_lse (flash-attn's option of the same name): also fp32 [sum Lq, Hq] log-sum-exp."""
    out = torch.empty_like(q)
    lse = torch.full(q.shape[:2], float("-inf"), dtype=torch.float32, device=q.device)
    ns = cu_seqlens_q.numel() - 1
    for s in range(ns):
        q0, q1 = int(cu_seqlens_q[s]), int(cu_seqlens_q[s + 1])
        k0, k1 = int(cu_seqlens_k[s]), int(cu_seqlens_k[s + 1])
        if q1 == q0:
            continue
        if block_table is None:
            ks, vs = k[k0:k1], v[k0:k1]
        else:
            ks = _gather_paged(k, block_table[s], k1 - k0)
            vs = _gather_paged(v, block_table[s], k1 - k0)
        off = (k1 - k0) - (q1 - q0) if causal else None
        out[q0:q1], lse[q0:q1] = _attend(q[q0:q1], ks, vs, softmax_scale, off, True)
    return (out, lse) if return_softmax_lse else out


def flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens, block_table, softmax_scale, causal=True,
                            return_softmax_lse=False):
    """Single-query paged attention as called at layers/attention.py:72-74. q [B,1,Hq,D] ->
    [B,1,Hq,D]. Rows with cache_seqlens == 0 (graph padding) return zeros (lse -inf)."""
    out = torch.zeros_like(q)
    lse = torch.full((q.shape[0], q.shape[2]), float("-inf"), dtype=torch.float32, device=q.device)
    for b in range(q.shape[0]):
        n = int(cache_seqlens[b])
        if n == 0:
            continue
        ks = _gather_paged(k_cache, block_table[b], n)
        vs = _gather_paged(v_cache, block_table[b], n)
        out[b], lse[b:b + 1] = _attend(q[b], ks, vs, softmax_scale, n - 1 if causal else None, True)
    return (out, lse) if return_softmax_lse else out


# ----------------------------------------------------------------------------------------------
# layers/sampler.py
def sampler_forward(logits: torch.Tensor, temperatures: torch.Tensor,
                    generator: torch.Generator | None = None) -> torch.Tensor:
    """Sampler.forward, layers/sampler.py:8-12 (exponential race)."""
    l = logits.float() / temperatures.unsqueeze(1)                # :9
    probs = torch.softmax(l, dim=-1)                              # :10
    e = torch.empty_like(probs).exponential_(1, generator=generator).clamp_min_(1e-10)
    return (probs / e).argmax(dim=-1)                             # :11


def sampler_keys(logits: torch.Tensor, temperatures: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """The race in log space, given the exponentials: argmax(p/E) == argmax(l/T - log E)."""
    return logits.float() / temperatures.unsqueeze(1) - torch.log(e.clamp_min(1e-10))


def greedy(logits: torch.Tensor) -> torch.Tensor:
    """temperature -> 0 limit of the sampler (first maximal index), the parity mode of SURVEY §8c."""
    return logits.float().argmax(dim=-1)


def bf16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|a - b| in units of bf16 ulps (ordered-integer distance of the bit patterns)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a.to(BF16)) - key(b.to(BF16))).abs()
