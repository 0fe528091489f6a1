"""Op-level drop-in layers: same class names, constructor arguments and forward signatures as
the reference's `nanovllm.layers.*`, implemented on the hand-written gfx950 kernels
(`nano_vllm_amd.ops` -> libnvl_hip.so). GEMMs and the embedding gather stay on PyTorch-ROCm
(hipBLASLt) — SURVEY.md §2c K9/K10.

Reference interfaces mirrored (file:line in GeeeekExplorer/nano-vllm):
  RMSNorm            layers/layernorm.py:5-50
  RotaryEmbedding    layers/rotary_embedding.py:17-59 (+ get_rope)
  SiluAndMul         layers/activation.py:6-11
  Attention          layers/attention.py:43-75
  Sampler            layers/sampler.py:5-12
  *Linear            layers/linear.py:12-156
  VocabParallelEmbedding / ParallelLMHead   layers/embed_head.py:9-66

There is no eager-PyTorch fallback for the hot ops: tensors must live on the GPU.
"""
from __future__ import annotations

import os
from functools import lru_cache

import torch
import torch.nn.functional as F
from torch import nn

from . import ops, tp
from .attn_meta import get_context


# ------------------------------------------------------------------------------------------------
class PartialSum:
    """A row-parallel GEMM output that has NOT been all-reduced yet: the consumer (RMSNorm.forward with a
    residual) performs the reduction fused with its own arithmetic."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


class RMSNorm(nn.Module):
    """y = bf16(x32 * rsqrt(mean x32^2 + eps) * w32); with `residual`: fused add, residual updated."""

    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        if residual is None:
            return ops.rmsnorm(x, self.weight, self.eps)
        # the reference returns (normed, bf16(x + residual)); we update `residual` in place
        if isinstance(x, PartialSum):
            # x = this rank's bf16 partial sums of a RowParallelLinear (`forward_decode`, TP > 1): the all-reduce
            # over xGMI and this norm are one launch (tp.all_reduce_add_rmsnorm)
            return tp.all_reduce_add_rmsnorm(x.t, residual, self.weight, self.eps), residual
        if x.dtype == torch.float32 and x.dim() == 3:
            # x = fp32 split-K partials [S, N, hidden] of a decode-step RowParallelLinear
            # (`forward_decode`): the slab sum is this kernel's prologue
            y = ops.add_rmsnorm_splitk(x, residual, self.weight, self.eps)
        else:
            y = ops.add_rmsnorm(x, residual, self.weight, self.eps)
        return y, residual


class SiluAndMul(nn.Module):

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        y = ops.silu_mul(x.reshape(-1, x.shape[-1]))
        return y.view(*lead, y.shape[-1])


class RotaryEmbedding(nn.Module):

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float) -> None:
        super().__init__()
        assert rotary_dim == head_size == 128, "libnvl kernels are built for head_dim 128"
        self.head_size = head_size
        # fp32 table cos || sin, computed on the host exactly as rotary_embedding.py:29-34
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float, device="cpu") / rotary_dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float, device="cpu")
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        table = torch.cat((freqs.cos(), freqs.sin()), dim=-1)
        self.register_buffer("cos_sin_cache", table.contiguous(), persistent=False)   # [max_pos, 128]

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        table = self.cos_sin_cache
        return ops.rope_neox(positions, table, query), ops.rope_neox(positions, table, key)


@lru_cache(4)
def _rope_singleton(head_size: int, rotary_dim: int, max_position: int, base: float, device: str):
    return RotaryEmbedding(head_size, rotary_dim, max_position, base).to(device)


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, device=None) -> RotaryEmbedding:
    """One shared table per geometry (the reference memoises with lru_cache(1), rotary_embedding.py:51-59)."""
    if device is None:
        device = f"cuda:{torch.cuda.current_device()}"
    return _rope_singleton(head_size, rotary_dim, max_position, float(base), str(device))


# NVL_FUSED_DECODE=0 keeps the separate q/k-norm+RoPE+KV-store launch on decode steps (A/B measurements)
_FUSED_DECODE = os.environ.get("NVL_FUSED_DECODE", "1") != "0"

class Attention(nn.Module):
    """Paged attention. `k_cache` / `v_cache` are injected by the runner; layout
    [num_blocks, num_kv_heads, block_size, 128] (head-major — see include/nvl.h)."""

    def __init__(self, num_heads: int, head_dim: int, scale: float, num_kv_heads: int):
        super().__init__()
        assert head_dim == 128, "libnvl kernels are built for head_dim 128"
        self.num_heads, self.head_dim, self.scale, self.num_kv_heads = num_heads, head_dim, scale, num_kv_heads
        self.k_cache = self.v_cache = torch.tensor([])

    # -- reference-shaped entry point: q/k/v already normed + rotated ---------------------------
    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        ctx = get_context()
        has_cache = self.k_cache.numel() > 0 and self.v_cache.numel() > 0
        if has_cache:
            ops.store_kvcache(k, v, self.k_cache, self.v_cache, ctx.slot_mapping)
        return self._attend(q, k, v, ctx)

    def fuses_decode_step(self, ctx) -> bool:
        """Does `forward_fused` take the one-launch decode path (norm + rope + KV store inside the attention kernel) for
        this step? Only then may `qkv` arrive as fp32 split-K slabs (QKVParallelLinear.forward_for_fused_decode)."""
        return not ctx.is_prefill and self.k_cache.numel() > 0 and _FUSED_DECODE

    # -- fused entry point: raw qkv GEMM output -> (q/k norm, rope, KV store) in one launch -------
    def forward_fused(self, qkv: torch.Tensor, positions: torch.Tensor, q_norm_w, k_norm_w, eps: float,
                      rope_table: torch.Tensor) -> torch.Tensor:
        ctx = get_context()
        n = qkv.shape[-2]             # ([S,] N, (Hq + 2 Hkv) * 128): fp32 split-K slabs only on the fused decode path
        hq, hkv = self.num_heads, self.num_kv_heads
        has_cache = self.k_cache.numel() > 0
        assert qkv.dim() == 2 or self.fuses_decode_step(ctx)
        if self.fuses_decode_step(ctx):
            # decode step: norm + rope + KV store happen inside the attention kernel (position and slot
            # of each sequence's new token follow from context_lens / block_tables)
            ws = ctx.decode_workspace
            max_context = ctx.max_context or ctx.block_tables.shape[1] * self.k_cache.shape[2]
            if ws is None:
                ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(n, hq, max_context), dtype=torch.uint8,
                                 device=qkv.device)
            return ops.paged_attn_decode_fused(qkv, q_norm_w, k_norm_w, eps, rope_table, self.k_cache, self.v_cache,
                                               ctx.block_tables, ctx.context_lens, hq, self.scale, max_context, ws,
                                               plan=ctx.decode_plan)
        q = torch.empty((n, hq, 128), dtype=qkv.dtype, device=qkv.device)
        need_k = ctx.is_prefill and ctx.block_tables is None      # non-paged prefill reads packed K
        k = torch.empty((n, hkv, 128), dtype=qkv.dtype, device=qkv.device) if need_k else None
        ops.qknorm_rope_kvstore(qkv, positions, q_norm_w, k_norm_w, eps, rope_table,
                                ctx.slot_mapping if has_cache else None, q, k,
                                self.k_cache if has_cache else None, self.v_cache if has_cache else None, hq, hkv)
        v = qkv[:, (hq + hkv) * 128:].view(n, hkv, 128) if need_k else None
        return self._attend(q, k, v, ctx)

    def _attend(self, q, k, v, ctx) -> torch.Tensor:
        if ctx.is_prefill:
            if ctx.block_tables is not None:        # prefix cache / chunk continuation: read the cache
                return ops.attn_prefill_varlen(q, self.k_cache, self.v_cache, ctx.cu_seqlens_q, ctx.cu_seqlens_k,
                                               ctx.max_seqlen_q, self.scale, block_tables=ctx.block_tables)
            return ops.attn_prefill_varlen(q, k, v, ctx.cu_seqlens_q, ctx.cu_seqlens_k, ctx.max_seqlen_q, self.scale)
        ws = ctx.decode_workspace
        max_context = ctx.max_context or ctx.block_tables.shape[1] * self.k_cache.shape[2]
        if ws is None:
            ws = torch.zeros(ops.paged_attn_decode_workspace_bytes(q.shape[0], self.num_heads, max_context),
                             dtype=torch.uint8, device=q.device)
        return ops.paged_attn_decode(q, self.k_cache, self.v_cache, ctx.block_tables, ctx.context_lens, self.scale,
                                     max_context, ws, plan=ctx.decode_plan)


class Sampler(nn.Module):
    """Exponential-race categorical sampling in one pass; temperature 0 => argmax (extension)."""

    def __init__(self, seed: int = 0, max_rows: int = 512):
        super().__init__()
        self.seed = seed
        self.max_rows = max_rows              # rows of the vocab-parallel winner buffers, allocated ONCE (graph-safe)
        self.calls = 0
        self._ws = None
        self.capture: list | None = None        # tests: every step's logits (fp32, host) are appended here

    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor, out: torch.Tensor | None = None,
                offset_dev: torch.Tensor | None = None, row_keys: torch.Tensor | None = None) -> torch.Tensor:
        """`row_keys` (int64 [B]: request ordinal | position << 32, extension): each row's draw is keyed by its
        sequence and position instead of by its row in this batch (the reference's generator is consumed in batch
        order, sampler.py:11)."""
        b = logits.shape[0]
        need = ops.sample_workspace_bytes(max(b, 512))
        if self._ws is None or self._ws.numel() < need or self._ws.device != logits.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=logits.device)
        if logits.dtype != torch.bfloat16:
            logits = logits.to(torch.bfloat16)
        offset = 0 if offset_dev is not None else self.calls
        self.calls += 1
        if self.capture is not None:
            self.capture.append(logits.float().cpu())
        return ops.sample(logits, temperatures, self.seed, offset, self._ws, out=out, offset_dev=offset_dev,
                          row_keys=row_keys)

    def _pair_buffers(self, b: int, size: int, device) -> None:
        """Select (allocating on first use, never regrowing) the {key, index} winner buffers for a b-row step:
        a 512-row pair for every captured decode step (4 KiB per rank travels: the P2P all-gather's limit, and a
        shape-independent message => graph-safe), and a `max_rows` pair for prefill steps with more sequences than
        that (never captured). A regrow would leave captured graphs pointing at freed memory."""
        if getattr(self, "_bufs", None) is None or self._bufs_device != device:
            self._bufs, self._bufs_device = {}, device
        rows = 512 if b <= 512 else max(self.max_rows, b)
        if rows not in self._bufs:
            assert rows == 512 or not torch.cuda.is_current_stream_capturing()
            self._bufs[rows] = (torch.zeros((rows, 2), dtype=torch.int32, device=device),
                                torch.zeros((size, rows, 2), dtype=torch.int32, device=device))
        self._mine, self._pairs = self._bufs[rows]

    def forward_shard(self, logits: torch.Tensor, temperatures: torch.Tensor, col_offset: int, out: torch.Tensor,
                      offset_dev: torch.Tensor | None = None, row_keys: torch.Tensor | None = None) -> torch.Tensor:
        """Vocab-parallel sampling (TP > 1): `logits` is this rank's [B, V/tp] shard starting at global column
        `col_offset`. Each rank reduces its shard to one {key, index} pair per row, the pairs (8 B per row)
        are all-gathered over xGMI and merged — same draw as `forward` on the gathered logits, on EVERY rank,
        without the reference's [B, V] gather to rank 0 (embed_head.py:62-65)."""
        b = logits.shape[0]
        _, size = tp.world()
        need = ops.sample_workspace_bytes(max(b, 512))
        if self._ws is None or self._ws.numel() < need or self._ws.device != logits.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=logits.device)
        self._pair_buffers(b, size, logits.device)
        offset = 0 if offset_dev is not None else self.calls
        self.calls += 1
        if self.capture is not None:
            self.capture.append(logits.float().cpu())
        ops.sample_shard(logits, temperatures, col_offset, self.seed, offset, self._ws, self._mine, offset_dev=offset_dev,
                         row_keys=row_keys)
        tp.all_gather_small(self._mine, self._pairs)
        return ops.sample_merge(self._pairs, size, b, out)


# ------------------------------------------------------------------------------------------------
# TP-sharded linears. Shard layout follows layers/linear.py:54-156.
def _divide(a: int, b: int) -> int:
    assert a % b == 0, f"{a} is not divisible by {b}"
    return a // b


class LinearBase(nn.Module):

    def __init__(self, input_size: int, output_size: int, bias: bool = False, tp_dim: int | None = None):
        super().__init__()
        self.tp_dim = tp_dim
        self.tp_rank, self.tp_size = tp.world()
        self.weight = nn.Parameter(torch.empty(output_size, input_size), requires_grad=False)
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.empty(output_size), requires_grad=False)
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)
        self.weight_packed: torch.Tensor | None = None     # tile-packed copy for nvl_linear_wide (pack_for_decode)

    def pack_for_decode(self, budget_bytes: float = float("inf")) -> int:
        """After the weights are loaded: keep a tile-packed copy (ops.pack_weight_tiles) of a projection whose decode
        GEMM is one of the hand-written kernels — the skinny kernel (K <= 1024-3072: Qwen3-0.6B) or the wide-tile one
        (deep reductions: Qwen3-8B / 32B shapes). The row-major parameter stays: prefill-sized GEMMs run on hipBLASLt
        from it. Returns the extra bytes (0: shape not covered; -1: covered, but the copy does not fit `budget_bytes`
        — ModelRunner._pack_weights). NVL_PACKED_WEIGHTS=0 keeps the row-major weight stream (A/B measurements)."""
        n, k = self.weight.shape
        if (self.bias is not None or not self.weight.is_cuda or n % 16 or k % 32
                or os.environ.get("NVL_PACKED_WEIGHTS", "1") == "0"):
            return 0
        skinny = any(ops.linear_decode_splits(144, n, k, mode) for mode in (ops.LINEAR_BF16, ops.LINEAR_PARTIAL))
        wide = (k % 128 == 0 and os.environ.get("NVL_GEMM_WIDE", "auto") != "0"
                and ops.linear_wide_plan(144, n, k, ops.LINEAR_BF16) is not None)
        if not (skinny or wide):
            return 0
        if n * k * 2 > budget_bytes:
            return -1
        self.weight_packed = ops.pack_weight_tiles(self.weight.data)
        return self.weight_packed.numel() * 2

    def _my_slice(self, loaded: torch.Tensor, dim: int) -> torch.Tensor:
        size = loaded.shape[dim] // self.tp_size
        return loaded.narrow(dim, self.tp_rank * size, size)

    def weight_loader(self, param: nn.Parameter, loaded: torch.Tensor, shard_id=None):
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.bias is None and _decode_sized(x):
            y = decode_linear(x, self.weight, ops.LINEAR_BF16, packed=self.weight_packed)
            if y is not None:
                return y
        return F.linear(x, self.weight, self.bias)


# Row counts up to this use the hand-written decode GEMMs (nvl_linear_decode / nvl_linear_wide); above it the
# library GEMM wins (measured crossover on MI355X, tools/gemm_bench.py).
DECODE_LINEAR_MAX_ROWS = 256


def _decode_sized(x: torch.Tensor) -> bool:
    return (x.dim() == 2 and x.shape[0] <= DECODE_LINEAR_MAX_ROWS and x.dtype == torch.bfloat16 and x.is_cuda
            and x.is_contiguous())


# ---- choice of the decode GEMM ---------------------------------------------------------------------------------
# K <= 1024 (Qwen3-0.6B): the skinny kernel (nvl_linear_decode), by shape rule. Deep reductions (Qwen3-8B / 32B, full
# width or per-rank): the wide-tile streaming kernel (nvl_linear_wide) on the module's tile-packed weight copy, by a
# DETERMINISTIC rule read off the measurements on MI355X (profiles/r03_gemm_wide_packed_vs_rowmajor.json: ours / hipBLASLt
# at 16 / 64 / 144 / 256 rows for every Qwen3-8B, 32B, 32B/TP4 and 32B/TP8 projection): with one row group (<= 144 rows)
# it is faster than the library GEMM on all but two of the 48 (shape, rows) pairs (each within 2.4 us); at 145-256 rows
# (one row group of 12 / 16 row tiles, or two paired ones: gemm_wide.hip) it wins on every bf16 and slab-output shape
# measured (32B down at 256 rows: 130 vs 264 us) and on SiLU outputs while the matrix is moderate (<= 140 MB: the
# per-rank gate_up shapes; at 145-192 rows the full-width 8B / 32B gate_up stay on the library) —
# profiles/r03_gemm_wide_m200_m256.json. At 193-256 rows the one-row-group form steps K by 64 columns since round 4
# (gemm_wide.hip::wide_bk64_pays) and takes the 8B gate_up as well (201 MB: 65.0 / 61.9 us at 208 / 256 rows against
# 74.1 / 68.1 for hipBLASLt + the SiLU launch; the 32B gate_up, 524 MB, stays on the library: 209 vs 161 us) —
# profiles/r04_gemm_wide_m256_bk64_vs_bk128.json. The same shapes therefore take the same kernel — hence the same bf16
# rounding — in every run.
# NVL_GEMM_WIDE=0 never uses it, =1 always (whenever the plan covers the shape), =tune decides by timing both once per
# (rows, n, k, mode) the first time the shape is seen outside a graph capture (a new device / shape family).
_WIDE_MAX_BYTES_ABOVE_144_ROWS = {0: float("inf"), 1: 140e6, 2: float("inf")}      # by ops.LINEAR_* mode
_WIDE_MAX_BYTES_ABOVE_192_ROWS = {0: float("inf"), 1: 256e6, 2: float("inf")}
_wide_choice: dict[tuple, bool] = {}
_wide_scratch: dict[tuple, torch.Tensor] = {}
_flush: dict[int, torch.Tensor] = {}


def _scratch(nbytes: int, device: torch.device) -> torch.Tensor | None:
    """Split-K slab scratch of the wide kernel's bf16 / SiLU modes: one tensor per distinct (size, stream), never freed
    or regrown (captured graphs hold its address). Per STREAM: launches on different streams must not share slabs."""
    if not nbytes:
        return None
    key = (nbytes, device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _wide_scratch.get(key)
    if t is None:
        t = _wide_scratch[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return t


def _time_cold(fn, device: torch.device, reps: int = 3) -> float:
    """Best-of GPU time (ms) of fn() with L2 / MALL flushed before every sample. The flush is enqueued first, so the
    events and fn are already queued when the GPU gets to them (no host gaps inside the bracket)."""
    buf = _flush.get(device.index)
    if buf is None:
        buf = _flush[device.index] = torch.empty(384 << 20, dtype=torch.uint8, device=device)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = float("inf")
    for _ in range(reps):
        buf.zero_()
        start.record()
        fn()
        stop.record()
        stop.synchronize()
        best = min(best, start.elapsed_time(stop))
    return best


def release_tuning_scratch() -> None:
    """Free the cache-flush buffer (called once the graphs are captured)."""
    _flush.clear()


def wide_choices() -> dict:
    """{(rows, n, k, mode, device): bool} decisions taken so far (diagnostics / bench.py)."""
    return dict(_wide_choice)


def _use_wide(x: torch.Tensor, weight: torch.Tensor, mode: int, packed: torch.Tensor | None = None) -> bool:
    m, k = x.shape
    n = weight.shape[0]
    key = (m, n, k, mode, x.device.index, packed is not None)
    c = _wide_choice.get(key)
    if c is not None:
        return c
    policy = os.environ.get("NVL_GEMM_WIDE", "auto")   # read when a shape is first seen
    plan = ops.linear_wide_plan(m, n, k, mode) if policy != "0" else None
    if plan is None:
        c = False
    elif policy == "1":
        c = True
    elif policy != "tune":
        limit = _WIDE_MAX_BYTES_ABOVE_192_ROWS if m > 192 else _WIDE_MAX_BYTES_ABOVE_144_ROWS
        c = m <= 144 or (m <= 256 and n * k * 2 <= limit[mode])
    elif torch.cuda.is_current_stream_capturing():
        return False                                   # an untimed shape inside a capture: library GEMM, not cached
    else:
        splits, ws_bytes = plan
        ws = _scratch(ws_bytes, x.device)
        wsrc, pk = (packed, True) if packed is not None else (weight, False)
        out = ops.linear_wide(x, wsrc, mode, workspace=ws, packed=pk)
        t_wide = _time_cold(lambda: ops.linear_wide(x, wsrc, mode, out=out, workspace=ws, packed=pk), x.device)
        if mode == ops.LINEAR_SILU:
            t_lib = _time_cold(lambda: ops.silu_mul(F.linear(x, weight)), x.device)
        else:
            t_lib = _time_cold(lambda: F.linear(x, weight), x.device)
        if mode == ops.LINEAR_PARTIAL:                 # the consumer reads S fp32 slabs instead of one bf16 matrix
            t_wide += (splits * 4 - 2) * m * n / 4e9
        c = t_wide < 0.97 * t_lib
    _wide_choice[key] = c
    return c


def decode_linear(x: torch.Tensor, weight: torch.Tensor, mode: int, out: torch.Tensor | None = None,
                  packed: torch.Tensor | None = None):
    """x [M, K] . weight[N, K]^T on a hand-written decode GEMM, or None when neither kernel takes the shape (the
    caller keeps the library GEMM). mode as in ops.linear_decode. `packed`: the module's tile-packed copy of
    `weight` (LinearBase.pack_for_decode), streamed by the wide kernel instead of the row-major parameter."""
    m, k = x.shape
    n = weight.shape[0]
    if ops.linear_decode_splits(m, n, k, mode):
        if packed is not None:
            return ops.linear_decode(x, packed, mode, out=out, packed=True)
        return ops.linear_decode(x, weight, mode, out=out)
    if _use_wide(x, weight, mode, packed):
        plan = ops.linear_wide_plan(m, n, k, mode)
        if packed is not None:
            return ops.linear_wide(x, packed, mode, out=out, workspace=_scratch(plan[1], x.device), packed=True)
        return ops.linear_wide(x, weight, mode, out=out, workspace=_scratch(plan[1], x.device))
    return None


class ReplicatedLinear(LinearBase):

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, output_size, bias)

    def weight_loader(self, param, loaded, shard_id=None):
        param.data.copy_(loaded)


class ColumnParallelLinear(LinearBase):
    """Output features split across ranks: rank r holds rows [r*out/tp, (r+1)*out/tp)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        _, size = tp.world()
        super().__init__(input_size, _divide(output_size, size), bias, tp_dim=0)

    def weight_loader(self, param, loaded, shard_id=None):
        param.data.copy_(self._my_slice(loaded, 0))


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel projections fused along the output dim (gate || up)."""

    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias)

    def weight_loader(self, param, loaded, shard_id=None):
        off = sum(self.output_sizes[:shard_id]) // self.tp_size
        size = self.output_sizes[shard_id] // self.tp_size
        param.data.narrow(0, off, size).copy_(self._my_slice(loaded, 0))

    def forward_silu(self, x: torch.Tensor) -> torch.Tensor:
        """SiluAndMul(self(x)) for a gate|up pair; on decode-sized inputs the activation is the GEMM's
        epilogue (one launch, no [N, 2*inter] round trip)."""
        if (self.bias is None and len(self.output_sizes) == 2 and self.output_sizes[0] == self.output_sizes[1]
                and _decode_sized(x)):
            y = decode_linear(x, self.weight, ops.LINEAR_SILU, packed=self.weight_packed)
            if y is not None:
                return y
        return ops.silu_mul(self.forward(x))


class QKVParallelLinear(ColumnParallelLinear):
    """Per-rank rows = [q: H/tp*D | k: Hkv/tp*D | v: Hkv/tp*D]."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        _, size = tp.world()
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = _divide(total_num_heads, size)
        self.num_kv_heads = _divide(total_num_kv_heads, size)
        super().__init__(hidden_size, (total_num_heads + 2 * total_num_kv_heads) * head_size, bias)

    def weight_loader(self, param, loaded, shard_id=None):
        q_rows = self.num_heads * self.head_size
        kv_rows = self.num_kv_heads * self.head_size
        off, size = {"q": (0, q_rows), "k": (q_rows, kv_rows), "v": (q_rows + kv_rows, kv_rows)}[shard_id]
        param.data.narrow(0, off, size).copy_(self._my_slice(loaded, 0))


    def forward_for_fused_decode(self, x: torch.Tensor) -> torch.Tensor:
        """The qkv projection of a decode step whose consumer is the fused attention kernel: where the deep-K GEMM splits
        K over workgroups anyway (Qwen3-8B / 32B, full width or per rank), its fp32 slabs [S, N, out] go straight to the
        attention prologue, which sums and rounds them (ops.paged_attn_decode_fused) — the slab-reduce launch between
        the two disappears (5 us + a launch boundary per layer on the per-rank shapes of Qwen3-32B at TP = 8). Same
        bits as forward(): the reduce kernel's sum order and rounding point are the prologue's. NVL_QKV_SLABS=0 keeps
        the reduce launch (A/B)."""
        if (_QKV_SLABS and self.bias is None and _decode_sized(x)
                and ops.decode_attention_takes_qkv_slabs(self.num_heads, self.num_kv_heads)):
            m, k = x.shape
            n = self.weight.shape[0]
            if not ops.linear_decode_splits(m, n, k, ops.LINEAR_BF16):       # (the skinny kernel never leaves slabs)
                plain = ops.linear_wide_plan(m, n, k, ops.LINEAR_BF16) if k % 128 == 0 else None
                slabs = ops.linear_wide_plan(m, n, k, ops.LINEAR_PARTIAL) if plain else None
                if (plain and plain[0] > 1 and slabs and slabs[0] <= 8
                        and _use_wide(x, self.weight, ops.LINEAR_PARTIAL, self.weight_packed)):
                    return decode_linear(x, self.weight, ops.LINEAR_PARTIAL, packed=self.weight_packed)
        return self.forward(x)


_QKV_SLABS = os.environ.get("NVL_QKV_SLABS", "1") != "0"


class RowParallelLinear(LinearBase):
    """Input features split across ranks; partial products are summed with an all-reduce
    (RCCL over xGMI; chunk-overlapped on a side stream for prefill-sized inputs — tp.py)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        _, size = tp.world()
        super().__init__(_divide(input_size, size), output_size, bias, tp_dim=1)

    def weight_loader(self, param, loaded, shard_id=None):
        if param.data.ndim == 1:          # bias: replicated, added on rank 0 only
            param.data.copy_(loaded)
        else:
            param.data.copy_(self._my_slice(loaded, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.tp_size == 1:
            return LinearBase.forward(self, x)
        bias = self.bias if self.tp_rank == 0 else None
        if bias is None and _decode_sized(x):
            return tp.all_reduce(LinearBase.forward(self, x))
        return tp.linear_allreduce(x, self.weight, bias)

    def forward_decode(self, x: torch.Tensor):
        """Like forward, but the result may come back UNREDUCED for the consumer to finish — the only consumer
        is RMSNorm.forward(x, residual):
          TP = 1, decode-sized: the GEMM's fp32 split-K partials [S, N, out] (summed + rounded in the norm's
                  prologue);
          TP > 1, message fits the P2P comm buffer: this rank's bf16 partial sums as `PartialSum` (the xGMI
                  all-reduce is fused with the residual-add + RMSNorm)."""
        n, k = self.weight.shape
        if self.tp_size == 1:
            if self.bias is None and _decode_sized(x):
                y = decode_linear(x, self.weight, ops.LINEAR_PARTIAL, packed=self.weight_packed)
                if y is not None:
                    return y
            return self.forward(x)
        c = tp.comm()
        if c is not None and self.bias is None and x.dim() == 2 and c.fits(x.shape[0], n):
            # the GEMM writes its partial sums straight into this rank's shared comm region: the all-reduce kernel
            # then starts at its first flag instead of a copy-in phase
            buf = c.input_buffer(x.shape[0], n, x.device)
            if not _decode_sized(x) or decode_linear(x, self.weight, ops.LINEAR_BF16, out=buf, packed=self.weight_packed) is None:
                torch.mm(x, self.weight.t(), out=buf)
            return PartialSum(buf)
        return self.forward(x)


class VocabParallelEmbedding(nn.Module):

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.tp_rank, self.tp_size = tp.world()
        self.num_embeddings = num_embeddings
        self.num_embeddings_per_partition = _divide(num_embeddings, self.tp_size)
        self.vocab_start_idx = self.num_embeddings_per_partition * self.tp_rank
        self.vocab_end_idx = self.vocab_start_idx + self.num_embeddings_per_partition
        self.weight = nn.Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim), requires_grad=False)
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param, loaded, shard_id=None):
        n = param.data.shape[0]
        param.data.copy_(loaded.narrow(0, self.tp_rank * n, n))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.tp_size == 1:
            return F.embedding(x, self.weight)
        mine = (x >= self.vocab_start_idx) & (x < self.vocab_end_idx)
        y = F.embedding((x - self.vocab_start_idx) * mine, self.weight)
        y = y * mine.unsqueeze(1)
        return tp.all_reduce(y)


class ParallelLMHead(VocabParallelEmbedding):

    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False):
        assert not bias
        super().__init__(num_embeddings, embedding_dim)
        self.weight_packed: torch.Tensor | None = None

    def pack_for_decode(self, budget_bytes: float = float("inf")) -> int:
        """Tile-packed copy of the vocabulary matrix for the decode step's lm_head GEMM: nvl_linear_wide on packed weights
        beats the library GEMM at <= 144 rows (Qwen3-8B / 32B, K >= 2048: profiles/r03_gemm_wide_lm_head.json; Qwen3-0.6B,
        K = 1024, round 6: 87-88 us vs 97 at 131 / 144 rows, profiles/r06_lm_head_06b_wide_vs_library.json — the skinny
        kernel's plan also covers that shape but re-reads x once per 32 columns, so `_logits` calls the wide kernel
        directly). Returns the extra bytes."""
        n, k = self.weight.shape
        if (not self.weight.is_cuda or n % 16 or k % 128 or k < 1024 or os.environ.get("NVL_PACKED_WEIGHTS", "1") == "0"
                or os.environ.get("NVL_GEMM_WIDE", "auto") == "0" or os.environ.get("NVL_PACKED_LM_HEAD", "1") == "0"
                or ops.linear_wide_plan(144, n, k, ops.LINEAR_BF16) is None):
            return 0
        if n * k * 2 > budget_bytes:
            return -1
        self.weight_packed = ops.pack_weight_tiles(self.weight.data)
        return self.weight_packed.numel() * 2

    def _logits(self, x: torch.Tensor) -> torch.Tensor:
        if self.weight_packed is not None and _decode_sized(x) and x.shape[0] <= 144:
            plan = ops.linear_wide_plan(x.shape[0], self.weight.shape[0], x.shape[1], ops.LINEAR_BF16)
            if plan is not None:
                return ops.linear_wide(x, self.weight_packed, ops.LINEAR_BF16, workspace=_scratch(plan[1], x.device), packed=True)
        return F.linear(x, self.weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor | None:
        ctx = get_context()
        if ctx.is_prefill:                                   # only each sequence's last token is sampled
            x = x[(ctx.cu_seqlens_q[1:] - 1).long()].contiguous()
        logits = self._logits(x)
        if self.tp_size == 1:
            return logits
        # reference-shaped result (full logits on rank 0, None elsewhere, embed_head.py:62-65). The engine
        # itself never takes this path at TP > 1: it samples from the shards (forward_shard + Sampler.forward_shard)
        import torch.distributed as dist
        if dist.get_backend() == "gloo" and logits.is_cuda:        # gloo has no device gather
            host = logits.cpu()
            parts = [torch.empty_like(host) for _ in range(self.tp_size)] if self.tp_rank == 0 else None
            dist.gather(host, parts, 0)
            return torch.cat(parts, -1).to(logits.device) if self.tp_rank == 0 else None
        parts = [torch.empty_like(logits) for _ in range(self.tp_size)] if self.tp_rank == 0 else None
        dist.gather(logits, parts, 0)
        return torch.cat(parts, -1) if self.tp_rank == 0 else None

    def forward_shard(self, x: torch.Tensor) -> torch.Tensor:
        """This rank's [S, V/tp] logits (columns vocab_start_idx ...), no collective."""
        ctx = get_context()
        if ctx.is_prefill:
            x = x[(ctx.cu_seqlens_q[1:] - 1).long()].contiguous()
        return self._logits(x)
