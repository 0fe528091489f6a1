// Single-pass token sampler for gfx950.
// Replaces Sampler.forward (nano-vllm layers/sampler.py:7-12):
//     argmax_i softmax(l/T)_i / E_i ,  E_i ~ Exp(1), clamp_min 1e-10
// = argmax_i ( l_i/T - log E_i )  — the softmax normaliser is common to the row and cancels,
// so one streaming read of the bf16 logits (2 B/logit) replaces the reference's fp32 [B,V]
// temporaries. E_i comes from Philox4x32-10 keyed by (seed; offset, row, column/4): the draw
// does not depend on grid shape, so host code can replay it (nvl_sample_exponentials_host).
// With `row_keys` (what the engine passes) the "row" of that key is not the batch row but the SEQUENCE:
// row_keys[r] = sequence id | position << 32 -> the draw uses row = low word, offset + high word, i.e. it depends on
// (seed, sequence, position, column) only — not on where the sequence sits in the batch, which step of the engine this
// is, or what else is in flight (lookahead, preemption, other requests).
// T == 0 selects plain argmax (lowest index wins ties), an extension the reference forbids
// (sampling_params.py:11) but the parity harness needs.
#include "common.h"
#include <math.h>

namespace {

constexpr int kSplits = 8;  // workgroups per row

// u in (0,1): 24 random bits, centred.
__host__ __device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__host__ __device__ __forceinline__ float exp1_from_bits(uint32_t x) {
#ifdef __HIP_DEVICE_COMPILE__
  const float e = -log_normal_f32(u01(x));
#else
  const float e = -logf(u01(x));
#endif
  return e < 1e-10f ? 1e-10f : e;
}

struct Best {
  float v;
  int idx;
};
__device__ __forceinline__ Best better(Best a, Best b) {
  // larger value wins; on ties the lower index (torch.argmax: first maximal value)
  if (b.v > a.v || (b.v == a.v && b.idx < a.idx)) return b;
  return a;
}

__global__ __launch_bounds__(256) void sample_partial_kernel(const bf16_t* __restrict__ logits, int64_t row_stride,
                                                              const float* __restrict__ temps, int64_t vocab,
                                                              int64_t col_offset, uint64_t seed, uint64_t offset,
                                                              const uint64_t* __restrict__ offset_dev,
                                                              const uint64_t* __restrict__ row_keys,
                                                              float* __restrict__ ws_val, int* __restrict__ ws_idx) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int64_t row = blockIdx.y;
  const int split = blockIdx.x;
  const float T = temps[row];
  const bool greedy = !(T > 0.f);
  const float invT = greedy ? 1.f : 1.f / T;
  const uint64_t rk = row_keys ? row_keys[row] : (uint64_t)row;
  const uint32_t rowid = (uint32_t)rk;
  const uint64_t off = offset + (offset_dev ? *offset_dev : 0ull) + (rk >> 32);
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const bf16_t* lr = logits + row * row_stride;
  const int64_t nchunks = (vocab + 7) >> 3;
  const int64_t per = (nchunks + kSplits - 1) / kSplits;
  const int64_t c_begin = split * per;
  const int64_t c_end = min(nchunks, c_begin + per);
  Best best{-INFINITY, 0x7fffffff};
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(lr) & 15) == 0);
  // `col_offset` (a multiple of 8) = global vocabulary index of local column 0: a vocab-parallel shard
  // (embed_head.py:56-66) draws the SAME exponentials and reports the SAME indices as the full row would.
  const int64_t gchunk0 = col_offset >> 3;
  for (int64_t c = c_begin + threadIdx.x; c < c_end; c += 256) {
    const int64_t col0 = c * 8;
    const int64_t gc = c + gchunk0;
    float f[8];
    if (vec_ok && col0 + 8 <= vocab) {
      unpack8(*reinterpret_cast<const u32x4_t*>(lr + col0), f);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = (col0 + i < vocab) ? (float)lr[col0 + i] : -INFINITY;
    }
    float key[8];
    if (greedy) {
#pragma unroll
      for (int i = 0; i < 8; ++i) key[i] = f[i];
    } else {
      const Philox4 r0 = philox4x32_10((uint32_t)(2 * gc), (uint32_t)((2 * gc) >> 32) ^ (uint32_t)(off << 8), rowid,
                                       (uint32_t)(off >> 24), k0, k1);
      const Philox4 r1 = philox4x32_10((uint32_t)(2 * gc + 1), (uint32_t)((2 * gc + 1) >> 32) ^ (uint32_t)(off << 8),
                                       rowid, (uint32_t)(off >> 24), k0, k1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t bits = i < 4 ? r0.v[i] : r1.v[i - 4];
        key[i] = f[i] * invT - log_normal_f32(exp1_from_bits(bits));
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (col0 + i < vocab) best = better(best, Best{key[i], (int)(col_offset + col0 + i)});
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other{__shfl_xor(best.v, o, 64), __shfl_xor(best.idx, o, 64)};
    best = better(best, other);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sv[wave] = best.v;
    si[wave] = best.idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Best b{sv[0], si[0]};
#pragma unroll
    for (int w = 1; w < 4; ++w) b = better(b, Best{sv[w], si[w]});
    ws_val[row * kSplits + split] = b.v;
    ws_idx[row * kSplits + split] = b.idx;
  }
}

// Merge `parts` (value, index) partials per row: part p of row r lives at val[p * part_stride + r * inner + j],
// j < inner (inner = kSplits for the workgroup partials of one launch, 1 for per-rank results). Either writes the
// winning index (int64, what Sampler.forward returns) or the packed {value bits, index} pair of a shard.
__global__ __launch_bounds__(64) void sample_merge_kernel(const float* __restrict__ val, const int* __restrict__ idx,
                                                           int parts, int64_t part_stride, int inner,
                                                           int64_t* __restrict__ out, uint32_t* __restrict__ out_packed,
                                                           int64_t batch) {
  const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (row >= batch) return;
  Best b{-INFINITY, 0x7fffffff};
  for (int p = 0; p < parts; ++p)
    for (int j = 0; j < inner; ++j) {
      const int64_t at = p * part_stride + row * inner + j;
      b = better(b, Best{val[at], idx[at]});
    }
  if (out) out[row] = b.idx == 0x7fffffff ? 0 : (int64_t)b.idx;
  if (out_packed) {
    out_packed[row * 2] = __float_as_uint(b.v);
    out_packed[row * 2 + 1] = (uint32_t)b.idx;
  }
}

// packed {value bits, index} pairs [parts][batch][2] (one part per tensor-parallel rank) -> winning index
__global__ __launch_bounds__(64) void sample_merge_packed_kernel(const uint32_t* __restrict__ packed, int parts,
                                                                  int64_t part_stride_words, int64_t* __restrict__ out,
                                                                  int64_t batch) {
  const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (row >= batch) return;
  Best b{-INFINITY, 0x7fffffff};
  for (int p = 0; p < parts; ++p) {
    const uint32_t* q = packed + p * part_stride_words + row * 2;
    b = better(b, Best{__uint_as_float(q[0]), (int)q[1]});
  }
  out[row] = b.idx == 0x7fffffff ? 0 : (int64_t)b.idx;
}

}  // namespace

extern "C" size_t nvl_sample_workspace_bytes(int64_t max_batch) {
  if (max_batch <= 0) return 0;
  return (size_t)max_batch * kSplits * (sizeof(float) + sizeof(int));
}

namespace {
int sample_common(const void* logits, int64_t logits_row_stride, const float* temperatures, int64_t* out,
                  uint32_t* out_packed, int64_t batch, int64_t vocab, int64_t col_offset, uint64_t seed, uint64_t offset,
                  const uint64_t* offset_dev, const uint64_t* row_keys, void* workspace, size_t workspace_bytes,
                  void* stream, const char* who) {
  NVL_REQUIRE(logits && temperatures && (out || out_packed) && workspace, "%s: null pointer", who);
  NVL_REQUIRE(batch >= 0 && batch <= 65535, "%s: batch=%lld out of range [0, 65535]", who, (long long)batch);
  NVL_REQUIRE(vocab > 0 && col_offset >= 0 && col_offset + vocab < (1ll << 31) - 8, "%s: bad vocab=%lld (+%lld)", who,
              (long long)vocab, (long long)col_offset);
  NVL_REQUIRE(col_offset % 8 == 0, "%s: col_offset=%lld must be a multiple of 8", who, (long long)col_offset);
  NVL_REQUIRE(logits_row_stride >= vocab, "%s: row stride < vocab", who);
  NVL_REQUIRE(workspace_bytes >= nvl_sample_workspace_bytes(batch), "%s: workspace too small", who);
  NVL_REQUIRE(((uintptr_t)workspace) % 8 == 0, "%s: workspace must be 8-byte aligned", who);
  if (batch == 0) return NVL_OK;
  float* ws_val = (float*)workspace;
  int* ws_idx = (int*)(ws_val + (size_t)batch * kSplits);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sample_partial_kernel, dim3(kSplits, (unsigned)batch), dim3(256), 0, s, (const bf16_t*)logits,
                     logits_row_stride, temperatures, vocab, col_offset, seed, offset, offset_dev, row_keys, ws_val, ws_idx);
  hipLaunchKernelGGL(sample_merge_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, s, ws_val, ws_idx, 1,
                     (int64_t)0, kSplits, out, out_packed, batch);
  return nvl_check_launch(who);
}
}  // namespace

extern "C" int nvl_sample(const void* logits, int64_t logits_row_stride, const float* temperatures, int64_t* out,
                          int64_t batch, int64_t vocab, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                          const uint64_t* row_keys, void* workspace, size_t workspace_bytes, void* stream) {
  return sample_common(logits, logits_row_stride, temperatures, out, nullptr, batch, vocab, 0, seed, offset,
                       offset_dev, row_keys, workspace, workspace_bytes, stream, "nvl_sample");
}

extern "C" int nvl_sample_shard(const void* logits, int64_t logits_row_stride, const float* temperatures,
                                void* best_packed, int64_t batch, int64_t vocab_local, int64_t col_offset,
                                uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                const uint64_t* row_keys, void* workspace, size_t workspace_bytes, void* stream) {
  NVL_REQUIRE(((uintptr_t)best_packed) % 8 == 0, "nvl_sample_shard: best_packed must be 8-byte aligned");
  return sample_common(logits, logits_row_stride, temperatures, nullptr, (uint32_t*)best_packed, batch, vocab_local,
                       col_offset, seed, offset, offset_dev, row_keys, workspace, workspace_bytes, stream,
                       "nvl_sample_shard");
}

extern "C" int nvl_sample_merge(const void* best_packed, int parts, int64_t part_stride_bytes, int64_t* out,
                                int64_t batch, void* stream) {
  NVL_REQUIRE(best_packed && out, "nvl_sample_merge: null pointer");
  NVL_REQUIRE(parts >= 1 && parts <= 1024, "nvl_sample_merge: parts=%d out of range", parts);
  NVL_REQUIRE(batch >= 0 && batch <= 65535, "nvl_sample_merge: batch=%lld out of range", (long long)batch);
  NVL_REQUIRE(part_stride_bytes % 8 == 0 && part_stride_bytes >= batch * 8, "nvl_sample_merge: bad part stride");
  NVL_REQUIRE(((uintptr_t)best_packed) % 8 == 0, "nvl_sample_merge: best_packed must be 8-byte aligned");
  if (batch == 0) return NVL_OK;
  hipLaunchKernelGGL(sample_merge_packed_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     (const uint32_t*)best_packed, parts, part_stride_bytes / 4, out, batch);
  return nvl_check_launch("nvl_sample_merge");
}

namespace {
__global__ __launch_bounds__(256) void feed_tokens_kernel(int64_t* __restrict__ ids, const int32_t* __restrict__ src,
                                                          const int64_t* __restrict__ prev, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int32_t r = src[i];
    if (r >= 0) ids[i] = prev[r];
  }
}
}  // namespace

extern "C" int nvl_feed_tokens(int64_t* ids, const int32_t* src_row, const int64_t* prev_tokens, int64_t n,
                               void* stream) {
  NVL_REQUIRE(ids && src_row && prev_tokens, "nvl_feed_tokens: null pointer");
  NVL_REQUIRE(n >= 0 && n < (1ll << 31), "nvl_feed_tokens: bad n=%lld", (long long)n);
  if (n == 0) return NVL_OK;
  hipLaunchKernelGGL(feed_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids,
                     src_row, prev_tokens, n);
  return nvl_check_launch("nvl_feed_tokens");
}

extern "C" void nvl_sample_exponentials_host(uint64_t seed, uint64_t offset, int64_t row, int64_t col0, int64_t n,
                                             float* e_host) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int64_t j = 0; j < n; ++j) {
    const int64_t col = col0 + j;
    const int64_t ctr = col >> 2;  // one Philox call per 4 columns: call index = 2*(col/8) + (col%8)/4
    const Philox4 r = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32) ^ (uint32_t)(offset << 8), (uint32_t)row,
                                    (uint32_t)(offset >> 24), k0, k1);
    e_host[j] = exp1_from_bits(r.v[col & 3]);
  }
}
