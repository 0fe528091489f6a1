"""ctypes binding of libnvl_hip.so (C ABI in include/nvl.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function below
hands raw device pointers + sizes to a hand-written gfx950 kernel. There is NO fallback:
if the library is missing or fails to load, importing callers get a loud RuntimeError.

`import torch` must precede loading the library so that it binds to the HIP runtime torch
already loaded (one runtime => shared streams, pointers and graph capture).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.environ.get("NVL_LIBDIR") or os.path.join(_HERE, "lib"), "libnvl_hip.so")   # (as build.LIBDIR)
ABI_VERSION = 6

# name -> (restype, argtypes); mirrors include/nvl.h one to one (checked by tests/test_abi.py)
SIGNATURES = {
    "nvl_abi_version": (c_int, []),
    "nvl_last_error": (c_char_p, []),
    "nvl_device_cu_count": (c_int, []),
    "nvl_rmsnorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_float, c_void_p]),
    "nvl_add_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "nvl_linear_decode_splits": (c_int, [c_int64, c_int, c_int, c_int]),
    "nvl_linear_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "nvl_linear_wide_plan": (c_int, [c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nvl_linear_wide": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                c_void_p]),
    "nvl_pack_weight_tiles": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "nvl_add_rmsnorm_splitk": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "nvl_silu_mul": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "nvl_rope_neox": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "nvl_store_kvcache": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64,
                                  c_int, c_void_p]),
    "nvl_qknorm_rope_kvstore": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                        c_int64, c_int, c_void_p]),
    "nvl_paged_attn_decode_workspace_bytes": (c_size_t, [c_int64, c_int, c_int64]),
    "nvl_paged_attn_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                      c_int, c_int, c_int, c_int64, c_int64, c_float, c_void_p, c_size_t, c_int,
                                      c_void_p, c_void_p, c_void_p]),
    "nvl_decode_plan_bytes": (c_size_t, []),
    "nvl_decode_plan": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "nvl_paged_attn_decode_fused": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p,
                                            c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int,
                                            c_int, c_int64, c_int64, c_float, c_void_p, c_size_t, c_int, c_void_p,
                                            c_void_p, c_int, c_int64, c_void_p]),
    "nvl_attn_prefill_varlen": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_float,
                                        c_int, c_void_p, c_void_p]),
    "nvl_sample_workspace_bytes": (c_size_t, [c_int64]),
    "nvl_sample": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_uint64, c_uint64, c_void_p,
                           c_void_p, c_void_p, c_size_t, c_void_p]),
    "nvl_sample_shard": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_uint64, c_uint64,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nvl_sample_merge": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "nvl_feed_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "nvl_allreduce_create": (c_int, [c_int, c_int, c_int64, ctypes.POINTER(c_void_p)]),
    "nvl_allreduce_uid": (c_int, [c_void_p, c_void_p]),
    "nvl_allreduce_connect": (c_int, [c_void_p, c_void_p]),
    "nvl_allreduce_max_bytes": (c_int64, [c_void_p]),
    "nvl_allreduce_buffer": (c_void_p, [c_void_p]),
    "nvl_allreduce_set_fences": (c_int, [c_void_p, c_int]),
    "nvl_allreduce_run": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "nvl_allreduce_add_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float,
                                          c_void_p]),
    "nvl_allreduce_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "nvl_allreduce_status": (c_int, [c_void_p]),
    "nvl_allreduce_status_async": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nvl_allreduce_destroy": (c_int, [c_void_p]),
    "nvl_sample_exponentials_host": (None, [c_uint64, c_uint64, c_int64, c_int64, c_int64, c_void_p]),
}

_lib = None


class NvlError(RuntimeError):
    pass


def load_library(path: str | None = None) -> ctypes.CDLL:
    """Load libnvl_hip.so and bind every symbol of the ABI. Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise NvlError(
            f"{path} not found: build it with `python -m nano_vllm_amd.build` (hipcc, gfx950). "
            "There is no CPU/PyTorch fallback for the hot path.")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => ABI mismatch, fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    ver = lib.nvl_abi_version()
    if ver != ABI_VERSION:
        raise NvlError(f"libnvl_hip.so ABI version {ver} != binding version {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def lib() -> ctypes.CDLL:
    return _lib if _lib is not None else load_library()


def _check(rc: int) -> None:
    if rc != 0:
        raise NvlError(f"nvl error {rc}: {lib().nvl_last_error().decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


KV_BF16, KV_FP8 = 0, 1


def kv_dtype_of(cache: torch.Tensor) -> int:
    """NVL_KV_* code of a cache tensor: bf16 (the reference's precision) or OCP fp8 e4m3 (opt-in extension)."""
    if cache.dtype == torch.bfloat16:
        return KV_BF16
    if cache.dtype == torch.float8_e4m3fn:
        return KV_FP8
    raise NvlError(f"unsupported KV-cache dtype {cache.dtype} (bfloat16 or float8_e4m3fn)")


def _dev(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise NvlError(f"{name} must be a device (HIP) tensor; the hot path has no CPU fallback")


# ---------------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: torch.Tensor | None = None) -> torch.Tensor:
    """y = bf16(float(x) * rsqrt(mean x^2 + eps) * float(w)); x is [..., hidden] or a strided
    [N, H, 128] head view (stride(-1) == 1, stride(-2) == hidden when 3-D)."""
    _dev(x, "x")
    hidden = x.shape[-1]
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.stride(-1) == 1
    if x.dim() == 3:
        n_outer, n_inner = x.shape[0], x.shape[1]
        assert x.stride(1) == hidden
        xs = x.stride(0)
    else:
        assert x.dim() == 2
        n_outer, n_inner, xs = x.shape[0], 1, x.stride(0)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    ys = out.stride(0)
    _check(lib().nvl_rmsnorm(x.data_ptr(), xs, weight.data_ptr(), out.data_ptr(), ys, n_outer, n_inner, hidden,
                             eps, _stream()))
    return out


def add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                out: torch.Tensor | None = None) -> torch.Tensor:
    """residual <- bf16(x + residual) in place; returns y = norm(un-rounded sum) * w."""
    _dev(x, "x")
    assert x.dim() == 2 and x.is_contiguous() and residual.is_contiguous() and x.shape == residual.shape
    assert x.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16
    if out is None:
        out = torch.empty_like(x)
    _check(lib().nvl_add_rmsnorm(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), out.data_ptr(), x.shape[0],
                                 x.shape[1], eps, _stream()))
    return out


def silu_mul(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    _dev(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.bfloat16
    inter = x.shape[1] // 2
    if out is None:
        out = torch.empty((x.shape[0], inter), dtype=x.dtype, device=x.device)
    _check(lib().nvl_silu_mul(x.data_ptr(), x.stride(0), out.data_ptr(), x.shape[0], inter, _stream()))
    return out


# ---- skinny (decode) linears ------------------------------------------------------------------
LINEAR_BF16, LINEAR_SILU, LINEAR_PARTIAL = 0, 1, 2
_splits_cache: dict[tuple, int] = {}


def linear_decode_splits(m: int, n: int, k: int, mode: int) -> int:
    """K-slices nvl_linear_decode emits for this shape (1 for modes 0/1); 0 = shape not covered."""
    key = (m, n, k, mode)
    r = _splits_cache.get(key)
    if r is None:
        r = _splits_cache[key] = lib().nvl_linear_decode_splits(m, n, k, mode)
    return r


def linear_decode(x: torch.Tensor, weight: torch.Tensor, mode: int = LINEAR_BF16,
                  out: torch.Tensor | None = None, packed: bool = False) -> torch.Tensor:
    """x [M, K] bf16, weight [N, K] bf16 (torch Linear layout; `packed`: its pack_weight_tiles() copy). mode 0: bf16
    [M, N]; mode 1: bf16 [M, N/2] = silu(x.Wgate) * (x.Wup); mode 2: fp32 split-K partials [S, M, N]."""
    _dev(x, "x")
    assert x.dim() == 2 and x.is_contiguous() and weight.is_contiguous()
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
    m, k = x.shape
    n = weight.shape[0]
    assert weight.shape[1] == k
    splits = linear_decode_splits(m, n, k, mode)
    if splits == 0:
        raise NvlError(f"nvl_linear_decode does not cover m={m} n={n} k={k} mode={mode}")
    if out is None:
        if mode == LINEAR_PARTIAL:
            out = torch.empty((splits, m, n), dtype=torch.float32, device=x.device)
        else:
            out = torch.empty((m, n // 2 if mode == LINEAR_SILU else n), dtype=torch.bfloat16, device=x.device)
    _check(lib().nvl_linear_decode(x.data_ptr(), weight.data_ptr(), out.data_ptr(), m, n, k, mode, 1 if packed else 0,
                                   _stream()))
    return out


_wide_cache: dict = {}


def linear_wide_plan(m: int, n: int, k: int, mode: int) -> tuple[int, int] | None:
    """(K splits, workspace bytes) of nvl_linear_wide's plan for this shape, or None when it is not covered."""
    key = (m, n, k, mode)
    if key not in _wide_cache:
        splits, ws = ctypes.c_int(0), ctypes.c_size_t(0)
        ok = lib().nvl_linear_wide_plan(m, n, k, mode, ctypes.byref(splits), ctypes.byref(ws))
        _wide_cache[key] = (splits.value, ws.value) if ok else None
    return _wide_cache[key]


def pack_weight_tiles(weight: torch.Tensor) -> torch.Tensor:
    """Tile-packed copy of a row-major [N, K] bf16 weight (nvl_pack_weight_tiles): same shape and bytes, stored as
    [N/16][K/32][64 lanes][8] so that every wave load of nvl_linear_wide's weight stream is one contiguous KiB."""
    _dev(weight, "weight")
    assert weight.dim() == 2 and weight.is_contiguous() and weight.dtype == torch.bfloat16
    n, k = weight.shape
    packed = torch.empty_like(weight)
    _check(lib().nvl_pack_weight_tiles(weight.data_ptr(), packed.data_ptr(), n, k, _stream()))
    return packed


def linear_wide(x: torch.Tensor, weight: torch.Tensor, mode: int = LINEAR_BF16, out: torch.Tensor | None = None,
                workspace: torch.Tensor | None = None, packed: bool = False) -> torch.Tensor:
    """Deep-K decode linear (same modes as linear_decode). `workspace`: uint8 scratch of the plan's size for
    modes 0 / 1 when the plan splits K (allocated here when omitted). `packed`: `weight` is the pack_weight_tiles()
    copy of the [N, K] parameter (same shape)."""
    _dev(x, "x")
    assert x.dim() == 2 and x.is_contiguous() and weight.is_contiguous()
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
    m, k = x.shape
    n = weight.shape[0]
    assert weight.shape[1] == k
    plan = linear_wide_plan(m, n, k, mode)
    if plan is None:
        raise NvlError(f"nvl_linear_wide does not cover m={m} n={n} k={k} mode={mode}")
    splits, ws_bytes = plan
    if out is None:
        if mode == LINEAR_PARTIAL:
            out = torch.empty((splits, m, n), dtype=torch.float32, device=x.device)
        else:
            out = torch.empty((m, n // 2 if mode == LINEAR_SILU else n), dtype=torch.bfloat16, device=x.device)
    if ws_bytes and workspace is None:
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    assert not ws_bytes or workspace.numel() * workspace.element_size() >= ws_bytes
    _check(lib().nvl_linear_wide(x.data_ptr(), weight.data_ptr(), out.data_ptr(), m, n, k, mode, 1 if packed else 0,
                                 workspace.data_ptr() if ws_bytes else None, ws_bytes, _stream()))
    return out


def add_rmsnorm_splitk(partials: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                       out: torch.Tensor | None = None) -> torch.Tensor:
    """partials fp32 [S, M, H]; residual <- bf16(bf16(sum_s partials) + residual) in place; returns the norm."""
    _dev(partials, "partials")
    assert partials.dim() == 3 and partials.is_contiguous() and partials.dtype == torch.float32
    s, m, h = partials.shape
    assert residual.shape == (m, h) and residual.is_contiguous() and residual.dtype == torch.bfloat16
    if out is None:
        out = torch.empty_like(residual)
    _check(lib().nvl_add_rmsnorm_splitk(partials.data_ptr(), s, residual.data_ptr(), weight.data_ptr(), out.data_ptr(),
                                        m, h, eps, _stream()))
    return out



def rope_neox(positions: torch.Tensor, cos_sin: torch.Tensor, x: torch.Tensor,
              out: torch.Tensor | None = None) -> torch.Tensor:
    """x: [N, H, 128] (stride(1) == 128); cos_sin: fp32 [max_pos, 128]."""
    _dev(x, "x")
    assert x.dim() == 3 and x.shape[2] == 128 and x.stride(2) == 1 and x.stride(1) == 128
    assert positions.dtype == torch.int64 and cos_sin.dtype == torch.float32 and cos_sin.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _check(lib().nvl_rope_neox(positions.data_ptr(), cos_sin.data_ptr(), cos_sin.shape[0], x.data_ptr(), x.stride(0),
                               out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], _stream()))
    return out


def store_kvcache(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                  slot_mapping: torch.Tensor) -> None:
    """k, v: [N, Hkv, 128] (stride(1) == 128); caches: [num_blocks, Hkv, block_size, 128]."""
    _dev(k, "k")
    n, hkv, d = k.shape
    assert d == 128 and k.stride(1) == 128 and v.stride(1) == 128 and k.stride(2) == 1 and v.stride(2) == 1
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and slot_mapping.dtype == torch.int32
    assert slot_mapping.numel() == n
    _check(lib().nvl_store_kvcache(k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), k_cache.data_ptr(),
                                   v_cache.data_ptr(), slot_mapping.data_ptr(), n, hkv, k_cache.shape[2],
                                   k_cache.shape[0], kv_dtype_of(k_cache), _stream()))


def qknorm_rope_kvstore(qkv: torch.Tensor, positions: torch.Tensor, q_norm_w, k_norm_w, eps: float,
                        cos_sin: torch.Tensor, slot_mapping, q_out: torch.Tensor, k_out, k_cache, v_cache,
                        num_q_heads: int, num_kv_heads: int) -> None:
    _dev(qkv, "qkv")
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and qkv.dtype == torch.bfloat16
    has_cache = k_cache is not None and k_cache.numel() > 0
    _check(lib().nvl_qknorm_rope_kvstore(
        qkv.data_ptr(), qkv.stride(0), positions.data_ptr(),
        q_norm_w.data_ptr() if q_norm_w is not None else None,
        k_norm_w.data_ptr() if k_norm_w is not None else None, eps,
        cos_sin.data_ptr(), cos_sin.shape[0],
        slot_mapping.data_ptr() if (has_cache and slot_mapping is not None) else None,
        q_out.data_ptr(), k_out.data_ptr() if k_out is not None else None,
        k_cache.data_ptr() if has_cache else None, v_cache.data_ptr() if has_cache else None,
        qkv.shape[0], num_q_heads, num_kv_heads,
        k_cache.shape[2] if has_cache else 0, k_cache.shape[0] if has_cache else 0,
        kv_dtype_of(k_cache) if has_cache else KV_BF16, _stream()))


def paged_attn_decode_workspace_bytes(max_batch: int, num_q_heads: int, max_context: int) -> int:
    return lib().nvl_paged_attn_decode_workspace_bytes(max_batch, num_q_heads, max_context)


def decode_plan_bytes() -> int:
    return lib().nvl_decode_plan_bytes()


def decode_plan(context_lens: torch.Tensor, num_q_heads: int, num_kv_heads: int, max_context: int,
                plan: torch.Tensor | None = None, shared_prefix: torch.Tensor | None = None,
                block_size: int = 256, prefix_groups: int = 1) -> torch.Tensor:
    """Per-step work plan of the decode attention launches (same for every layer of the step): `plan` is a uint8
    device buffer of decode_plan_bytes() bytes, allocated when None. `shared_prefix`: int32 device tensor [1 + batch] —
    [0] = number of leading KV blocks the member sequences have in common, [1 + b] != 0 marks sequence b as a member;
    the attention calls that consume this plan then run the shared-prefix pass, and read the member flags from this
    tensor's memory when they run (include/nvl.h). `prefix_groups` > 1: the flags are group ids (several shared prefixes in one
    step) and a pack of rows may hold up to that many different groups."""
    _dev(context_lens, "context_lens")
    assert context_lens.dtype == torch.int32 and context_lens.is_contiguous()
    if plan is None:
        plan = torch.empty(decode_plan_bytes(), dtype=torch.uint8, device=context_lens.device)
    shp = None
    if shared_prefix is not None:
        assert shared_prefix.dtype == torch.int32 and shared_prefix.device == context_lens.device
        assert shared_prefix.is_contiguous() and shared_prefix.numel() >= 1 + context_lens.numel()
        shp = shared_prefix.data_ptr()
    _check(lib().nvl_decode_plan(context_lens.data_ptr(), context_lens.numel(), num_q_heads, num_kv_heads, max_context,
                                 shp, block_size, prefix_groups, plan.data_ptr(), plan.numel(), _stream()))
    return plan


def decode_attention_shares_prefixes(num_q_heads: int, num_kv_heads: int, block_size: int) -> bool:
    """Can a decode plan carry a shared prefix for this geometry? (matrix-core kernel, 128-token-aligned blocks)"""
    return block_size % 128 == 0 and decode_attention_takes_qkv_slabs(num_q_heads, num_kv_heads)


def _lse_ptr(lse: torch.Tensor | None, shape: tuple) -> int | None:
    if lse is None:
        return None
    assert lse.dtype == torch.float32 and lse.is_contiguous() and tuple(lse.shape) == shape, (lse.shape, shape)
    return lse.data_ptr()


def paged_attn_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor,
                      context_lens: torch.Tensor, scale: float, max_context: int, workspace: torch.Tensor,
                      out: torch.Tensor | None = None, plan: torch.Tensor | None = None,
                      lse: torch.Tensor | None = None) -> torch.Tensor:
    """q: [B, Hq, 128]; caches [num_blocks, Hkv, block_size, 128]; block_tables int32 [B, W]. `plan`: decode_plan()
    of this step; `lse`: optional fp32 [B, Hq] output (log-sum-exp of the scaled scores)."""
    _dev(q, "q")
    b, hq, d = q.shape
    assert d == 128 and q.is_contiguous() and block_tables.dtype == torch.int32 and context_lens.dtype == torch.int32
    assert block_tables.stride(1) == 1
    if out is None:
        out = torch.empty_like(q)
    _check(lib().nvl_paged_attn_decode(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(),
                                       block_tables.stride(0), context_lens.data_ptr(), out.data_ptr(), b, hq,
                                       k_cache.shape[1], k_cache.shape[2], k_cache.shape[0], max_context, scale,
                                       workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                       kv_dtype_of(k_cache), plan.data_ptr() if plan is not None else None,
                                       _lse_ptr(lse, (b, hq)), _stream()))
    return out


def paged_attn_decode_fused(qkv: torch.Tensor, q_norm_w, k_norm_w, eps: float, cos_sin: torch.Tensor,
                            k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor,
                            context_lens: torch.Tensor, num_q_heads: int, scale: float, max_context: int,
                            workspace: torch.Tensor, out: torch.Tensor | None = None,
                            plan: torch.Tensor | None = None, lse: torch.Tensor | None = None) -> torch.Tensor:
    """Decode step in one launch: q/k-norm + RoPE (position = context_len - 1) + KV-cache store of the
    new token + paged attention. qkv: raw qkv GEMM output [B, (Hq + 2*Hkv)*128] bf16 — or, for a deep-K projection,
    the fp32 split-K slabs [S, B, (Hq + 2*Hkv)*128] of `linear_wide(..., LINEAR_PARTIAL)` (S <= 8; summed and rounded
    in the attention prologue: no separate slab-reduce launch; matrix-core kernel only, Hq / Hkv in 2 ... 16)."""
    _dev(qkv, "qkv")
    splits, split_stride = 0, 0
    if qkv.dtype == torch.float32:
        assert qkv.dim() == 3 and qkv.is_contiguous() and 1 <= qkv.shape[0] <= 8
        splits, split_stride = qkv.shape[0], qkv.stride(0)
        b, row_stride = qkv.shape[1], qkv.stride(1)
    else:
        assert qkv.dim() == 2 and qkv.stride(1) == 1 and qkv.dtype == torch.bfloat16
        b, row_stride = qkv.shape[0], qkv.stride(0)
    assert block_tables.dtype == torch.int32 and context_lens.dtype == torch.int32 and block_tables.stride(1) == 1
    assert cos_sin.dtype == torch.float32 and cos_sin.is_contiguous()
    if out is None:
        out = torch.empty((b, num_q_heads, 128), dtype=torch.bfloat16, device=qkv.device)
    _check(lib().nvl_paged_attn_decode_fused(
        qkv.data_ptr(), row_stride, q_norm_w.data_ptr() if q_norm_w is not None else None,
        k_norm_w.data_ptr() if k_norm_w is not None else None, eps, cos_sin.data_ptr(), cos_sin.shape[0],
        k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(), block_tables.stride(0),
        context_lens.data_ptr(), out.data_ptr(), b, num_q_heads, k_cache.shape[1], k_cache.shape[2], k_cache.shape[0],
        max_context, scale, workspace.data_ptr(), workspace.numel() * workspace.element_size(), kv_dtype_of(k_cache),
        plan.data_ptr() if plan is not None else None, _lse_ptr(lse, (b, num_q_heads)), splits, split_stride, _stream()))
    return out


def decode_attention_takes_qkv_slabs(num_q_heads: int, num_kv_heads: int) -> bool:
    """Can `paged_attn_decode_fused` sum fp32 qkv slabs itself for this head geometry? (the matrix-core kernel: group
    sizes 2 ... 16 — 2, 4, 8 unless their packed-dot forms are forced by NVL_DECODE_MFMA=0 / NVL_DECODE_G8_VALU=1; every other
    group size, e.g. Qwen3-14B's 40 / 8 = 5, only has the matrix-core kernel)"""
    g = num_q_heads // max(num_kv_heads, 1)
    if g == 8:
        return os.environ.get("NVL_DECODE_G8_VALU", "0") != "1"
    if g in (2, 4):
        return os.environ.get("NVL_DECODE_MFMA", "1") != "0"
    return 1 < g <= 16


def attn_prefill_varlen(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens_q: torch.Tensor,
                        cu_seqlens_k: torch.Tensor, max_seqlen_q: int, scale: float,
                        block_tables: torch.Tensor | None = None, num_kv_heads: int | None = None,
                        out: torch.Tensor | None = None, lse: torch.Tensor | None = None) -> torch.Tensor:
    """q: [Nq, Hq, 128]. block_tables None: k, v packed [Nk, Hkv, 128] (token-strided views ok);
    otherwise k, v are the paged caches [num_blocks, Hkv, block_size, 128]. `lse`: optional fp32 [Nq, Hq] output."""
    _dev(q, "q")
    nq, hq, d = q.shape
    assert d == 128 and q.is_contiguous()
    assert cu_seqlens_q.dtype == torch.int32 and cu_seqlens_k.dtype == torch.int32
    if out is None:
        out = torch.empty_like(q)
    num_seqs = cu_seqlens_q.numel() - 1
    if block_tables is None:
        hkv = k.shape[1]
        assert k.stride(1) == 128 and v.stride(1) == 128 and k.stride(2) == 1 and v.stride(2) == 1
        _check(lib().nvl_attn_prefill_varlen(q.data_ptr(), k.data_ptr(), v.data_ptr(), k.stride(0), v.stride(0),
                                             cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr(), None, 0,
                                             out.data_ptr(), nq, num_seqs, max_seqlen_q, hq, hkv, 0, 0, scale,
                                             KV_BF16, _lse_ptr(lse, (nq, hq)), _stream()))
    else:
        assert block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
        _check(lib().nvl_attn_prefill_varlen(q.data_ptr(), k.data_ptr(), v.data_ptr(), 0, 0,
                                             cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr(),
                                             block_tables.data_ptr(), block_tables.stride(0), out.data_ptr(), nq,
                                             num_seqs, max_seqlen_q, hq, k.shape[1], k.shape[2], k.shape[0], scale,
                                             kv_dtype_of(k), _lse_ptr(lse, (nq, hq)), _stream()))
    return out


def sample_workspace_bytes(max_batch: int) -> int:
    return lib().nvl_sample_workspace_bytes(max_batch)


def _keys_ptr(row_keys: torch.Tensor | None, b: int) -> int | None:
    if row_keys is None:
        return None
    assert row_keys.dtype == torch.int64 and row_keys.is_contiguous() and row_keys.numel() >= b and row_keys.is_cuda
    return row_keys.data_ptr()


def sample(logits: torch.Tensor, temperatures: torch.Tensor, seed: int, offset: int, workspace: torch.Tensor,
           out: torch.Tensor | None = None, offset_dev: torch.Tensor | None = None,
           row_keys: torch.Tensor | None = None) -> torch.Tensor:
    """logits bf16 [B, V]; temperatures fp32 [B] (0 => argmax); returns int64 [B]. `row_keys` (int64 [B] holding
    sequence_id | position << 32): draws keyed by (sequence, position) instead of by batch row."""
    _dev(logits, "logits")
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype == torch.bfloat16
    assert temperatures.dtype == torch.float32
    b, vocab = logits.shape
    if out is None:
        out = torch.empty(b, dtype=torch.int64, device=logits.device)
    _check(lib().nvl_sample(logits.data_ptr(), logits.stride(0), temperatures.data_ptr(), out.data_ptr(), b, vocab,
                            seed & 0xFFFFFFFFFFFFFFFF, offset & 0xFFFFFFFFFFFFFFFF,
                            offset_dev.data_ptr() if offset_dev is not None else None, _keys_ptr(row_keys, b),
                            workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()))
    return out


def sample_shard(logits: torch.Tensor, temperatures: torch.Tensor, col_offset: int, seed: int, offset: int,
                 workspace: torch.Tensor, out_packed: torch.Tensor, offset_dev: torch.Tensor | None = None,
                 row_keys: torch.Tensor | None = None) -> torch.Tensor:
    """This rank's vocabulary shard -> one {key bits, global index} pair per row (int32 [B, 2])."""
    _dev(logits, "logits")
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype == torch.bfloat16
    assert temperatures.dtype == torch.float32 and out_packed.dtype == torch.int32 and out_packed.is_contiguous()
    b, vocab = logits.shape
    assert out_packed.numel() >= 2 * b
    _check(lib().nvl_sample_shard(logits.data_ptr(), logits.stride(0), temperatures.data_ptr(), out_packed.data_ptr(), b,
                                  vocab, col_offset, seed & 0xFFFFFFFFFFFFFFFF, offset & 0xFFFFFFFFFFFFFFFF,
                                  offset_dev.data_ptr() if offset_dev is not None else None, _keys_ptr(row_keys, b),
                                  workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()))
    return out_packed


def sample_merge(packed: torch.Tensor, parts: int, batch: int, out: torch.Tensor) -> torch.Tensor:
    """packed int32 [parts, >= batch, 2] (contiguous rows) -> out int64 [batch]: index of the largest key."""
    _dev(packed, "packed")
    assert packed.dtype == torch.int32 and packed.dim() == 3 and packed.shape[0] == parts and packed.shape[2] == 2
    assert packed.stride(2) == 1 and packed.stride(1) == 2 and out.dtype == torch.int64
    _check(lib().nvl_sample_merge(packed.data_ptr(), parts, packed.stride(0) * 4, out.data_ptr(), batch, _stream()))
    return out


class P2PComm:
    """Handle of the hand-written xGMI collectives (csrc/comm.hip). Construction is collective:
    `exchange(bytes) -> list[bytes]` must all-gather a 64-byte token across the tensor-parallel ranks
    and `barrier()` must synchronise them (both host-side, out of band)."""

    def __init__(self, rank: int, world: int, max_bytes: int, exchange, barrier):
        self.rank, self.world = rank, world
        self._h = None
        self._views: dict[tuple, torch.Tensor] = {}
        # Every rank takes part in the exchange AND in the barrier whatever happened locally: a rank that raised
        # before them would leave its peers waiting inside a collective it never joins. A rank that could not
        # create its communicator publishes an all-zero token, which makes every peer fail the same way.
        err: Exception | None = None
        none = b"\0" * 64
        uid = ctypes.create_string_buffer(64)
        try:
            h = c_void_p()
            _check(lib().nvl_allreduce_create(rank, world, max_bytes, ctypes.byref(h)))
            self._h = h
            _check(lib().nvl_allreduce_uid(self._h, uid))
        except Exception as ex:  # noqa: BLE001 — re-raised after the collectives below
            err = ex
        uids = exchange(uid.raw if err is None else none)
        if err is None:
            try:
                assert len(uids) == world and all(len(u) == 64 for u in uids)
                if any(u == none for u in uids):
                    raise NvlError("nvl_allreduce: a peer could not create its communicator")
                blob = ctypes.create_string_buffer(b"".join(uids), 64 * world)
                _check(lib().nvl_allreduce_connect(self._h, blob))
            except Exception as ex:  # noqa: BLE001
                err = ex
        barrier()
        if err is not None:
            self.close()
            raise err
        self.max_bytes = int(lib().nvl_allreduce_max_bytes(self._h))
        # Hand-off flavour: FENCED (system-scope release / acquire around every flag) until somebody has validated
        # the lean form — per-wave store drains only, resting on the shared buffer being uncached on both sides of a
        # link — on the topology at hand: tp.init_p2p does that with a randomised stress run and then calls
        # set_handoff("lean") (14.7 -> 8.7 us per 131 x 5120 all-reduce, profiles/r02_p2p_bench_w2.json).
        self.handoff = "fenced"
        _check(lib().nvl_allreduce_set_fences(self._h, 1))

    def set_handoff(self, flavour: str) -> None:
        """"fenced" | "lean" — must be called with the same value on every rank, between collectives."""
        assert flavour in ("fenced", "lean")
        _check(lib().nvl_allreduce_set_fences(self._h, 1 if flavour == "fenced" else 0))
        self.handoff = flavour

    def input_buffer(self, rows: int, hidden: int, device) -> torch.Tensor:
        """A [rows, hidden] bf16 tensor that IS this rank's shared input region: a GEMM that writes its output
        here hands it to `all_reduce*` without the kernel's copy-in phase."""
        key = (rows, hidden)
        t = self._views.get(key)
        if t is None:
            assert rows * hidden * 2 <= self.max_bytes
            ptr = int(lib().nvl_allreduce_buffer(self._h))

            class _Region:                      # torch wraps foreign device memory through the CUDA array interface
                __cuda_array_interface__ = {"shape": (rows * hidden,), "typestr": "<i2", "data": (ptr, False), "version": 2}
            t = torch.as_tensor(_Region(), device=device).view(torch.bfloat16).view(rows, hidden)
            self._views[key] = t
        return t

    def fits(self, rows: int, hidden: int) -> bool:
        return rows * hidden * 2 <= self.max_bytes and hidden % (8 * self.world) == 0 and hidden <= 8192

    def all_reduce(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        _dev(x, "x")
        assert x.dim() == 2 and x.is_contiguous() and x.dtype == torch.bfloat16
        out = x if out is None else out
        _check(lib().nvl_allreduce_run(self._h, x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _stream()))
        return out

    def all_reduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                               out: torch.Tensor | None = None) -> torch.Tensor:
        _dev(x, "x")
        assert x.dim() == 2 and x.is_contiguous() and residual.is_contiguous() and x.shape == residual.shape
        assert x.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16
        if out is None:
            out = torch.empty_like(x)
        _check(lib().nvl_allreduce_add_rmsnorm(self._h, x.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                               out.data_ptr(), x.shape[0], x.shape[1], eps, _stream()))
        return out

    def all_gather(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """out[world, nbytes(x)] <- every rank's x (nbytes a multiple of 16, <= 4096)."""
        _dev(x, "x")
        n = x.numel() * x.element_size()
        assert x.is_contiguous() and out.is_contiguous() and out.numel() * out.element_size() >= n * self.world
        _check(lib().nvl_allreduce_gather(self._h, x.data_ptr(), out.data_ptr(), n, _stream()))
        return out

    def status(self) -> None:
        _check(lib().nvl_allreduce_status(self._h))

    def status_async(self, out: torch.Tensor) -> None:
        """out[0] (int32, device) <- 1 if any rank of the group has latched a spin timeout so far, else 0. Enqueue-only
        (capturable): the serving path's form of `status`."""
        _dev(out, "out")
        assert out.dtype == torch.int32 and out.numel() >= 1
        _check(lib().nvl_allreduce_status_async(self._h, out.data_ptr(), _stream()))

    def close(self) -> None:
        if self._h is not None:
            lib().nvl_allreduce_destroy(self._h)
            self._h = None


def feed_tokens(ids: torch.Tensor, src_row: torch.Tensor, prev_tokens: torch.Tensor) -> None:
    """ids[i] = prev_tokens[src_row[i]] where src_row[i] >= 0 (in place; int64 ids, int32 rows)."""
    _dev(ids, "ids")
    assert ids.dtype == torch.int64 and prev_tokens.dtype == torch.int64 and src_row.dtype == torch.int32
    assert ids.is_contiguous() and src_row.is_contiguous() and src_row.numel() == ids.numel()
    _check(lib().nvl_feed_tokens(ids.data_ptr(), src_row.data_ptr(), prev_tokens.data_ptr(), ids.numel(), _stream()))


def sample_exponentials_host(seed: int, offset: int, row: int, col0: int, n: int):
    import numpy as np
    e = np.empty(n, dtype=np.float32)
    lib().nvl_sample_exponentials_host(seed, offset, row, col0, n, e.ctypes.data_as(c_void_p))
    return e
