// Rotary embedding, paged KV-cache store, and the fused q/k-norm -> RoPE -> KV-store kernel.
// Layout: 16 lanes x 8 bf16 cover one 128-wide head; a wave handles 4 heads.
// The neox half-split pairs element i with i+64, i.e. lane `sub` with lane `sub ^ 8` of the
// same 16-lane DPP row, so the rotation partner arrives by one row_ror:8.
// Reference semantics: layers/rotary_embedding.py:6-14,37-48; layers/attention.py:10-40;
// models/qwen3.py:82-85 (q_norm/k_norm before the rotation).
#include "common.h"
#include "rope_common.h"

// Keep mul/add separately rounded (as the fp32 oracle does): no FMA contraction in this file.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int64_t cache_row_offset(int64_t slot, int head, int num_kv_heads, int block_size) {
  const int64_t blk = slot / block_size;
  const int64_t off = slot - blk * block_size;
  return ((blk * num_kv_heads + head) * block_size + off) * NVL_HEAD_DIM;
}

__global__ __launch_bounds__(256) void rope_kernel(const int64_t* __restrict__ positions,
                                                    const float* __restrict__ cos_sin, int64_t max_pos,
                                                    const bf16_t* __restrict__ x, int64_t x_tok_stride,
                                                    bf16_t* __restrict__ out, int64_t out_tok_stride,
                                                    int64_t n_tok, int n_heads) {
  const int64_t total = n_tok * n_heads;
  const int64_t unit0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const bool live = unit0 < total;
  const int64_t unit = live ? unit0 : 0;
  const int64_t tok = unit / n_heads;
  const int head = (int)(unit - tok * n_heads);
  int64_t pos = positions[tok];
  pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
  float v[8], o[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(x + tok * x_tok_stride + head * 128 + sub * 8), v);
  rope8(v, cos_sin + pos * 128, sub, o);
  if (live) *reinterpret_cast<u32x4_t*>(out + tok * out_tok_stride + head * 128 + sub * 8) = pack8(o);
}

// One cache row of this lane's 8 elements: bf16 (16 bytes) or, KV8, OCP fp8 e4m3 (8 bytes; the cache then holds
// 128 bytes per (token, head) row and every offset below is in BYTES = elements).
template <bool KV8>
__device__ __forceinline__ void put_row8(void* cache, int64_t elem_off, const u32x4_t& w) {
  if constexpr (KV8) {
    *reinterpret_cast<u32x2_t*>(static_cast<unsigned char*>(cache) + elem_off) = bf16x8_to_fp8x8(w);
  } else {
    *reinterpret_cast<u32x4_t*>(static_cast<bf16_t*>(cache) + elem_off) = w;
  }
}

template <bool KV8>
__global__ __launch_bounds__(256) void store_kv_kernel(const bf16_t* __restrict__ k, int64_t k_tok_stride,
                                                        const bf16_t* __restrict__ v, int64_t v_tok_stride,
                                                        void* __restrict__ k_cache, void* __restrict__ v_cache,
                                                        const int32_t* __restrict__ slot_mapping, int64_t n_tok,
                                                        int num_kv_heads, int block_size) {
  const int64_t total = n_tok * num_kv_heads;
  const int64_t unit = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  if (unit >= total) return;
  const int64_t tok = unit / num_kv_heads;
  const int head = (int)(unit - tok * num_kv_heads);
  const int64_t slot = slot_mapping[tok];
  if (slot < 0) return;  // layers/attention.py:23
  const int64_t dst = cache_row_offset(slot, head, num_kv_heads, block_size) + sub * 8;
  put_row8<KV8>(k_cache, dst, *reinterpret_cast<const u32x4_t*>(k + tok * k_tok_stride + head * 128 + sub * 8));
  put_row8<KV8>(v_cache, dst, *reinterpret_cast<const u32x4_t*>(v + tok * v_tok_stride + head * 128 + sub * 8));
}

// Fused: unit = (token, head) over Hq + 2*Hkv heads of the qkv GEMM output row.
//   q head : norm -> rope -> q_out
//   k head : norm -> rope -> k_out (optional) and k_cache[slot]
//   v head : copy -> v_cache[slot]
template <bool KV8>
__global__ __launch_bounds__(256) void qknorm_rope_kvstore_kernel(
    const bf16_t* __restrict__ qkv, int64_t qkv_tok_stride, const int64_t* __restrict__ positions,
    const bf16_t* __restrict__ q_norm_w, const bf16_t* __restrict__ k_norm_w, float eps,
    const float* __restrict__ cos_sin, int64_t max_pos, const int32_t* __restrict__ slot_mapping,
    bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_out, void* __restrict__ k_cache,
    void* __restrict__ v_cache, int64_t n_tok, int hq, int hkv, int block_size) {
  const int htot = hq + 2 * hkv;
  const int64_t total = n_tok * htot;
  const int64_t unit0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const bool live = unit0 < total;
  const int64_t unit = live ? unit0 : 0;
  const int64_t tok = unit / htot;
  const int head = (int)(unit - tok * htot);
  const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qkv + tok * qkv_tok_stride + head * 128 + sub * 8);
  const int64_t slot = (k_cache != nullptr) ? (int64_t)slot_mapping[tok] : -1;

  if (head >= hq + hkv) {  // value head: straight copy into the cache
    if (live && slot >= 0) {
      const int kvh = head - hq - hkv;
      put_row8<KV8>(v_cache, cache_row_offset(slot, kvh, hkv, block_size) + sub * 8, raw);
    }
    return;  // whole 16-lane rows take this branch together; DPP below stays row-local
  }
  const bool is_q = head < hq;
  float v[8], o[8];
  unpack8(raw, v);
  const bf16_t* w = is_q ? q_norm_w : k_norm_w;
  if (w != nullptr) headnorm8(v, w, sub, eps);
  int64_t pos = positions[tok];
  pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
  rope8(v, cos_sin + pos * 128, sub, o);
  const u32x4_t packed = pack8(o);
  if (!live) return;
  if (is_q) {
    *reinterpret_cast<u32x4_t*>(q_out + (tok * hq + head) * 128 + sub * 8) = packed;
  } else {
    const int kvh = head - hq;
    if (k_out != nullptr) *reinterpret_cast<u32x4_t*>(k_out + (tok * hkv + kvh) * 128 + sub * 8) = packed;
    if (slot >= 0) put_row8<KV8>(k_cache, cache_row_offset(slot, kvh, hkv, block_size) + sub * 8, packed);
  }
}

}  // namespace

extern "C" int nvl_rope_neox(const int64_t* positions, const float* cos_sin, int64_t max_pos, const void* x,
                             int64_t x_tok_stride, void* out, int64_t out_tok_stride, int64_t n_tok, int n_heads,
                             void* stream) {
  NVL_REQUIRE(positions && cos_sin && x && out, "nvl_rope_neox: null pointer");
  NVL_REQUIRE(n_tok >= 0 && n_heads > 0 && max_pos > 0, "nvl_rope_neox: bad sizes");
  NVL_REQUIRE(x_tok_stride % 8 == 0 && out_tok_stride % 8 == 0, "nvl_rope_neox: strides must be multiples of 8");
  NVL_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)cos_sin) % 16 == 0, "nvl_rope_neox: pointers must be 16-byte aligned");
  const int64_t total = n_tok * n_heads;
  if (total == 0) return NVL_OK;
  NVL_REQUIRE((total + 15) / 16 < (1ll << 31), "nvl_rope_neox: too many rows");
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, (hipStream_t)stream, positions,
                     cos_sin, max_pos, (const bf16_t*)x, x_tok_stride, (bf16_t*)out, out_tok_stride, n_tok, n_heads);
  return nvl_check_launch("nvl_rope_neox");
}

extern "C" int nvl_store_kvcache(const void* k, int64_t k_tok_stride, const void* v, int64_t v_tok_stride,
                                 void* k_cache, void* v_cache, const int32_t* slot_mapping, int64_t n_tok,
                                 int num_kv_heads, int block_size, int64_t num_blocks, int kv_dtype, void* stream) {
  NVL_REQUIRE(k && v && k_cache && v_cache && slot_mapping, "nvl_store_kvcache: null pointer");
  NVL_REQUIRE(kv_dtype == NVL_KV_BF16 || kv_dtype == NVL_KV_FP8, "nvl_store_kvcache: kv_dtype=%d (0 bf16, 1 fp8 e4m3)", kv_dtype);
  NVL_REQUIRE(n_tok >= 0 && num_kv_heads > 0 && block_size > 0 && num_blocks > 0, "nvl_store_kvcache: bad sizes");
  NVL_REQUIRE(k_tok_stride % 8 == 0 && v_tok_stride % 8 == 0, "nvl_store_kvcache: strides must be multiples of 8");
  NVL_REQUIRE(((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)v_cache) % 16 == 0,
              "nvl_store_kvcache: pointers must be 16-byte aligned");
  const int64_t total = n_tok * num_kv_heads;
  if (total == 0) return NVL_OK;
  NVL_REQUIRE((total + 15) / 16 < (1ll << 31), "nvl_store_kvcache: too many rows");
  if (kv_dtype == NVL_KV_FP8)
    hipLaunchKernelGGL(store_kv_kernel<true>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)k, k_tok_stride, (const bf16_t*)v, v_tok_stride, k_cache, v_cache, slot_mapping,
                       n_tok, num_kv_heads, block_size);
  else
    hipLaunchKernelGGL(store_kv_kernel<false>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)k, k_tok_stride, (const bf16_t*)v, v_tok_stride, k_cache, v_cache, slot_mapping,
                       n_tok, num_kv_heads, block_size);
  return nvl_check_launch("nvl_store_kvcache");
}

extern "C" int nvl_qknorm_rope_kvstore(const void* qkv, int64_t qkv_tok_stride, const int64_t* positions,
                                       const void* q_norm_w, const void* k_norm_w, float eps, const float* cos_sin,
                                       int64_t max_pos, const int32_t* slot_mapping, void* q_out, void* k_out,
                                       void* k_cache, void* v_cache, int64_t n_tok, int num_q_heads,
                                       int num_kv_heads, int block_size, int64_t num_blocks, int kv_dtype,
                                       void* stream) {
  NVL_REQUIRE(qkv && positions && cos_sin && q_out, "nvl_qknorm_rope_kvstore: null pointer");
  NVL_REQUIRE(kv_dtype == NVL_KV_BF16 || kv_dtype == NVL_KV_FP8, "nvl_qknorm_rope_kvstore: kv_dtype=%d (0 bf16, 1 fp8 e4m3)", kv_dtype);
  NVL_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), "nvl_qknorm_rope_kvstore: q/k norm weights must both be set or both NULL");
  NVL_REQUIRE((k_cache == nullptr) == (v_cache == nullptr), "nvl_qknorm_rope_kvstore: k_cache/v_cache must both be set or both NULL");
  NVL_REQUIRE(k_cache == nullptr || slot_mapping != nullptr, "nvl_qknorm_rope_kvstore: slot_mapping required with a cache");
  NVL_REQUIRE(k_cache == nullptr || (block_size > 0 && num_blocks > 0), "nvl_qknorm_rope_kvstore: bad cache geometry");
  NVL_REQUIRE(n_tok >= 0 && num_q_heads > 0 && num_kv_heads > 0 && max_pos > 0, "nvl_qknorm_rope_kvstore: bad sizes");
  NVL_REQUIRE(qkv_tok_stride % 8 == 0 && qkv_tok_stride >= (int64_t)(num_q_heads + 2 * num_kv_heads) * 128,
              "nvl_qknorm_rope_kvstore: bad qkv stride");
  NVL_REQUIRE(((uintptr_t)qkv | (uintptr_t)q_out | (uintptr_t)k_out | (uintptr_t)k_cache | (uintptr_t)v_cache |
               (uintptr_t)q_norm_w | (uintptr_t)k_norm_w | (uintptr_t)cos_sin) % 16 == 0,
              "nvl_qknorm_rope_kvstore: pointers must be 16-byte aligned");
  const int64_t total = n_tok * (num_q_heads + 2 * num_kv_heads);
  if (total == 0) return NVL_OK;
  NVL_REQUIRE((total + 15) / 16 < (1ll << 31), "nvl_qknorm_rope_kvstore: too many rows");
  if (kv_dtype == NVL_KV_FP8)
    hipLaunchKernelGGL(qknorm_rope_kvstore_kernel<true>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)qkv, qkv_tok_stride, positions, (const bf16_t*)q_norm_w,
                       (const bf16_t*)k_norm_w, eps, cos_sin, max_pos, slot_mapping, (bf16_t*)q_out, (bf16_t*)k_out,
                       k_cache, v_cache, n_tok, num_q_heads, num_kv_heads, block_size);
  else
    hipLaunchKernelGGL(qknorm_rope_kvstore_kernel<false>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)qkv, qkv_tok_stride, positions, (const bf16_t*)q_norm_w,
                       (const bf16_t*)k_norm_w, eps, cos_sin, max_pos, slot_mapping, (bf16_t*)q_out, (bf16_t*)k_out,
                       k_cache, v_cache, n_tok, num_q_heads, num_kv_heads, block_size);
  return nvl_check_launch("nvl_qknorm_rope_kvstore");
}
