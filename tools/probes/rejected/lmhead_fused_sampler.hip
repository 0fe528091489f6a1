// lm_head GEMM with the token sampler in its epilogue, for gfx950 (decode steps: <= 256 rows).
// Replaces, in ONE pass over the vocabulary matrix, ParallelLMHead.forward (nano-vllm layers/embed_head.py:56-66:
// F.linear(x, weight) -> [B, V] logits) followed by Sampler.forward (layers/sampler.py:7-12), as the reference
// runs them back to back at engine/model_runner.py:212-218. The [B, V] logits (39 MB at B = 131, bf16) are never
// written to or re-read from HBM: every workgroup reduces its 256 columns to one {sampling key, index} pair per
// row, and a tiny second kernel merges the ~600 pairs per row. Same arithmetic as nvl_sample on bf16-rounded
// logits (one-pass exponential race argmax_i l_i/T - log E_i, Philox keyed by (seed, offset, row, global column),
// T == 0 => argmax with lowest index on ties), so it also serves a vocabulary shard (col_offset) under TP.
//
// Decomposition ("wide tile": the opposite of gemm_decode.hip's, because here N = 151,936 gives 594 workgroups
// without splitting K, and x must NOT be re-read per 32 columns — that costs 1.4 GB of L2 traffic for 311 MB of
// weights, profiles/r02_gemm_deep_lm_head_kb2.json):
//   * workgroup = 8 waves = 256 vocabulary rows of W (NT = 2 sixteen-row MFMA tiles per wave) x ALL batch rows;
//     the waves split N, so there is no cross-wave reduction of accumulators.
//   * K advances in 128-wide steps. The x tile of a step ([batch, 128] bf16, 36 KiB at 144 rows) is staged ONCE per
//     workgroup into LDS (registers -> ds_write_b128, 16-byte XOR swizzle => conflict-free ds_read_b128 B fragments)
//     and read by all 8 waves; two LDS stages, one barrier per step.
//   * W fragments go HBM -> VGPR with non-temporal loads through a 3-deep register ring: the loads of step s + 2
//     are issued at the start of step s, so every wave keeps 16 KiB (the workgroup 128 KiB) of the weight stream in
//     flight. x loads of step s + 1 are issued BEFORE them (loads retire in order: x must not queue behind HBM).
//   * v_mfma_f32_16x16x32_bf16, A = W fragment, B = x fragment: lane (m = lane & 15, q = lane >> 4) ends up with 4
//     consecutive vocabulary columns of batch row m — exactly one Philox4x32 draw.
//   * 145-192 rows: 12 row tiles x NT = 1 (128 columns per workgroup) keeps the accumulators in registers and the
//     matrix streamed once; above 192 rows the caller keeps GEMM + nvl_sample (16 row tiles spill).
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kNW = 8;
constexpr int kBK = 128;                 // k per step
constexpr int kKB = kBK / 32;            // 32-wide MFMA k blocks per step

__host__ __device__ __forceinline__ float u01_bits(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

struct Best {
  float v;
  int idx;
};
__device__ __forceinline__ Best better(Best a, Best b) {
  if (b.v > a.v || (b.v == a.v && b.idx < a.idx)) return b;
  return a;
}

template <int MT, int NT, int SB>
__global__ __launch_bounds__(kNW * 64) void lmhead_sample_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ temps,
    uint32_t* __restrict__ partial, bf16_t* __restrict__ logits_out, int M, int V, int K, int64_t col_offset,
    uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev, const uint64_t* __restrict__ row_keys) {
  constexpr int kRows = MT * 16;
  constexpr int kCH = (kRows * 16 + kNW * 64 - 1) / (kNW * 64);   // 16-byte x chunks per thread per step
  // bytes per LDS stage: every thread writes its kCH chunks unconditionally (chunks past the last row land in the
  // stage's padding, never read) — a lane-divergent guard around the load / LDS write pair makes hipcc wait vmcnt(0)
  // inside the branch, which drains the weight prefetch ring every step
  constexpr int kStage = kCH * kNW * 64 * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int steps = K / kBK;
  const int n0 = (blockIdx.x * kNW + wave) * (NT * 16);     // this wave's first (local) vocabulary column

  const bf16_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int row = n0 + nt * 16 + l15;
    row = row < V ? row : V - 1;                            // ragged last workgroup: any valid row, masked below
    wrow[nt] = w + (int64_t)row * K + lq * 8;
  }
  // x staging: chunk c = tid + i * 512 -> row c >> 4, 16-byte column c & 15
  int x_src[kCH], x_dst[kCH];
#pragma unroll
  for (int i = 0; i < kCH; ++i) {
    const int c = tid + i * (kNW * 64);
    int row = c >> 4;
    const int col = c & 15;
    x_dst[i] = row * 256 + ((col ^ (row & 15)) << 4);       // rows >= kRows: the padding of the stage
    row = row < M ? row : M - 1;                            // padding rows read a valid row (never reported)
    x_src[i] = row * K + col * 8;
  }
  int frag_off[kKB];                                        // B fragment of k block kb: row l15 of a row tile
#pragma unroll
  for (int kb = 0; kb < kKB; ++kb) frag_off[kb] = l15 * 256 + (((kb * 4 + lq) ^ l15) << 4);

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ring depth: 3 sets (two steps ahead) while the accumulators leave room, 2 sets (one step ahead: still 64 KiB of
  // weights in flight per workgroup) for the 9 x 2 and 16 x 1 tilings, which would spill otherwise
  constexpr int RING = (MT * NT > 12) ? 2 : 3;
  u32x4_t wf[RING][NT][kKB];
  u32x4_t xr[kCH];
  auto wload = [&](u32x4_t (*dst)[kKB], int s) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int kb = 0; kb < kKB; ++kb)
        dst[nt][kb] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wrow[nt] + s * kBK + kb * 32));
  };
  auto xload = [&](int s) {
#pragma unroll
    for (int i = 0; i < kCH; ++i) xr[i] = *reinterpret_cast<const u32x4_t*>(x + x_src[i] + s * kBK);
  };
  auto xwrite = [&](int stage) {
#pragma unroll
    for (int i = 0; i < kCH; ++i) *reinterpret_cast<u32x4_t*>(smem + stage * kStage + x_dst[i]) = xr[i];
  };

  // K is walked in blocks of SB steps whose code is STRAIGHT-LINE (fully unrolled, compile-time guards): hipcc's
  // s_waitcnt insertion is exact in straight-line code, whereas at loop headers / branch joins it merges the
  // outstanding-load scoreboards conservatively and ends up waiting vmcnt(0..5) before every use — which silently
  // turns a 3-deep prefetch ring into a synchronous load (seen in the .s of the first version of this kernel:
  // 154 us instead of ~65 for the Qwen3-0.6B head at 131 rows). The ring drains at a block boundary (one HBM round
  // trip per SB steps; Qwen3-0.6B is a single block).
  for (int blk = 0; blk < steps / SB; ++blk) {
    const int s0 = blk * SB;
    xload(s0);
    wload(wf[0], s0);
    if constexpr (SB > 1 && RING > 2) wload(wf[1], s0 + 1);
    __builtin_amdgcn_sched_barrier(0);
    xwrite(0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      // next step's x first (L2), THEN the weights two steps ahead (HBM): loads retire in order
      if (i + 1 < SB) xload(s0 + i + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (i + RING - 1 < SB) wload(wf[(i + RING - 1) % RING], s0 + i + RING - 1);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* xs = smem + (i & 1) * kStage;
      {
        // fragments of row tile mt + 1 are read from LDS under the MFMAs of tile mt (pinned: hipcc otherwise
        // re-serialises read -> wait -> NT MFMAs through one register quad)
        u32x4_t f[2][kKB];
#pragma unroll
        for (int kb = 0; kb < kKB; ++kb) f[0][kb] = *reinterpret_cast<const u32x4_t*>(xs + frag_off[kb]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (mt + 1 < MT) {
#pragma unroll
            for (int kb = 0; kb < kKB; ++kb)
              f[(mt + 1) & 1][kb] = *reinterpret_cast<const u32x4_t*>(xs + (mt + 1) * 16 * 256 + frag_off[kb]);
          }
#pragma unroll
          for (int kb = 0; kb < kKB; ++kb)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[i % RING][nt][kb]),
                                                                    __builtin_bit_cast(bf16x8_t, f[mt & 1][kb]), acc[mt][nt], 0, 0, 0);
          if (mt + 1 < MT) {
#pragma unroll
            for (int j = 0; j < kKB; ++j) {
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read ...
              __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);  // ... per NT MFMAs
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < SB) xwrite((i + 1) & 1);                  // that stage was last read one barrier ago
      __syncthreads();
    }
  }

  // ---- epilogue: bf16-round the logits, (optionally store them,) reduce each row's sampling key ---------------
  const uint64_t off0 = offset + (offset_dev ? *offset_dev : 0ull);
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  Best* red = reinterpret_cast<Best*>(smem);                // [kNW][kRows] (the x stages are dead: barrier above)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + l15;
    const bool row_ok = m < M;
    // the row's identity in the draw: the batch row, or (sequence, position) when the caller passes row keys
    const uint64_t rk = (row_keys != nullptr && row_ok) ? row_keys[m] : (uint64_t)m;
    const uint32_t rowid = (uint32_t)rk;
    const uint64_t off = off0 + (rk >> 32);
    const float T = row_ok ? temps[m] : 0.f;
    const bool greedy = !(T > 0.f);
    const float invT = greedy ? 1.f : 1.f / T;
    Best best{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col0 = n0 + nt * 16 + lq * 4;               // local column of element 0 of this lane's quad
      const int64_t gcol0 = col_offset + col0;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = round_bf16(acc[mt][nt][r]);
      if (logits_out != nullptr && row_ok && col0 + 3 < V) {
        u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(logits_out + (int64_t)m * V + col0) = o;
      } else if (logits_out != nullptr && row_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col0 + r < V) logits_out[(int64_t)m * V + col0 + r] = (bf16_t)v[r];
      }
      float key[4];
      if (__all(greedy)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) key[r] = v[r];
      } else {
        const int64_t ctr = gcol0 >> 2;                      // the draw nvl_sample makes for these 4 columns
        const Philox4 rnd = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32) ^ (uint32_t)(off << 8), rowid,
                                          (uint32_t)(off >> 24), k0, k1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = -log_normal_f32(u01_bits(rnd.v[r]));
          e = e < 1e-10f ? 1e-10f : e;
          key[r] = greedy ? v[r] : v[r] * invT - log_normal_f32(e);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col0 + r < V) best = better(best, Best{key[r], (int)(gcol0 + r)});
    }
    // the 4 lanes l15, l15 + 16, + 32, + 48 hold the same row
    Best o1{__shfl_xor(best.v, 16, 64), __shfl_xor(best.idx, 16, 64)};
    best = better(best, o1);
    Best o2{__shfl_xor(best.v, 32, 64), __shfl_xor(best.idx, 32, 64)};
    best = better(best, o2);
    if (lq == 0) red[wave * kRows + m] = best;
  }
  __syncthreads();
  if (tid < kRows && tid < M) {
    Best b = red[tid];
#pragma unroll
    for (int ww = 1; ww < kNW; ++ww) b = better(b, red[ww * kRows + tid]);
    uint32_t* dst = partial + ((int64_t)blockIdx.x * M + tid) * 2;
    dst[0] = __float_as_uint(b.v);
    dst[1] = (uint32_t)b.idx;
  }
}

// One workgroup per batch row: merge `parts` packed {key bits, index} pairs (part p of row r at
// packed[(p * batch + r) * 2]) into the winning index (int64) and/or the packed winner of this shard.
__global__ __launch_bounds__(256) void lmhead_merge_kernel(const uint32_t* __restrict__ packed, int parts, int batch,
                                                            int64_t* __restrict__ out, uint32_t* __restrict__ out_packed) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int row = blockIdx.x;
  Best b{-INFINITY, 0x7fffffff};
  for (int p = threadIdx.x; p < parts; p += 256) {
    const uint32_t* q = packed + ((int64_t)p * batch + row) * 2;
    b = better(b, Best{__uint_as_float(q[0]), (int)q[1]});
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best other{__shfl_xor(b.v, o, 64), __shfl_xor(b.idx, o, 64)};
    b = better(b, other);
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = b.v;
    si[threadIdx.x >> 6] = b.idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Best r{sv[0], si[0]};
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) r = better(r, Best{sv[ww], si[ww]});
    if (out) out[row] = r.idx == 0x7fffffff ? 0 : (int64_t)r.idx;
    if (out_packed) {
      out_packed[row * 2] = __float_as_uint(r.v);
      out_packed[row * 2 + 1] = (uint32_t)r.idx;
    }
  }
}

struct LmPlan {
  int mt, nt, groups, sb;
};

bool lm_plan(int64_t batch, int64_t vocab, int k, LmPlan* p) {
  if (batch < 1 || batch > 192 || vocab < 16 || k < kBK || k % kBK) return false;   // 16 row tiles spill: not built
  const int steps = k / kBK;
  p->sb = steps % 8 == 0 ? 8 : (steps % 5 == 0 ? 5 : (steps % 3 == 0 ? 3 : 0));     // straight-line block length
  if (!p->sb) return false;
  const int mtiles = (int)((batch + 15) / 16);
  static const int kMT2[] = {1, 2, 3, 5, 7, 9};
  p->mt = 0;
  for (int c : kMT2)
    if (c >= mtiles) { p->mt = c; p->nt = 2; break; }
  if (!p->mt) { p->mt = 12; p->nt = 1; }
  const int cols = kNW * p->nt * 16;
  p->groups = (int)((vocab + cols - 1) / cols);
  return true;
}

template <int MT, int NT, int SB>
int launch_lm(const LmPlan& p, const void* x, const void* w, const float* temps, uint32_t* partial, void* logits,
              int64_t batch, int64_t vocab, int k, int64_t col_offset, uint64_t seed, uint64_t offset,
              const uint64_t* offset_dev, const uint64_t* row_keys, hipStream_t s) {
  const size_t lds_x = (size_t)2 * ((MT * 16 * 16 + kNW * 64 - 1) / (kNW * 64)) * kNW * 64 * 16;
  const size_t lds_red = (size_t)kNW * MT * 16 * sizeof(Best);
  const size_t lds = lds_x > lds_red ? lds_x : lds_red;
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& attr_set = attr_done[nvl_device_slot()];
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lmhead_sample_kernel<MT, NT, SB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      nvl_set_error("nvl_lmhead_sample: cannot reserve %zu B of LDS", lds);
      return NVL_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((lmhead_sample_kernel<MT, NT, SB>), dim3((unsigned)p.groups), dim3(kNW * 64), lds, s,
                     (const bf16_t*)x, (const bf16_t*)w, temps, partial, (bf16_t*)logits, (int)batch, (int)vocab, k,
                     col_offset, seed, offset, offset_dev, row_keys);
  return NVL_OK;
}

}  // namespace

extern "C" size_t nvl_lmhead_sample_workspace_bytes(int64_t batch, int64_t vocab_local, int k) {
  LmPlan p;
  if (!lm_plan(batch, vocab_local, k, &p)) return 0;         // 0 = shape not covered: keep GEMM + nvl_sample
  return (size_t)p.groups * batch * 2 * sizeof(uint32_t);
}

extern "C" int nvl_lmhead_sample(const void* x, const void* weight, const float* temperatures, int64_t* out,
                                 void* best_packed, void* logits_out, int64_t batch, int64_t vocab_local, int k,
                                 int64_t col_offset, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                 const uint64_t* row_keys, void* workspace, size_t workspace_bytes, void* stream) {
  NVL_REQUIRE(x && weight && temperatures && workspace && (out || best_packed), "nvl_lmhead_sample: null pointer");
  NVL_REQUIRE(((uintptr_t)x | (uintptr_t)weight | (uintptr_t)workspace | (uintptr_t)logits_out | (uintptr_t)best_packed) % 8 == 0 &&
                  ((uintptr_t)x | (uintptr_t)weight) % 16 == 0,
              "nvl_lmhead_sample: x / weight must be 16-byte aligned (others 8)");
  NVL_REQUIRE(col_offset >= 0 && col_offset % 8 == 0 && col_offset + vocab_local < (1ll << 31) - 8,
              "nvl_lmhead_sample: bad col_offset=%lld", (long long)col_offset);
  LmPlan p;
  if (!lm_plan(batch, vocab_local, k, &p)) {
    nvl_set_error("nvl_lmhead_sample: shape batch=%lld vocab=%lld k=%d not covered (batch <= 192, k / 128 a multiple of 8, 5 or 3)",
                  (long long)batch, (long long)vocab_local, k);
    return NVL_EUNSUPPORTED;
  }
  NVL_REQUIRE(workspace_bytes >= (size_t)p.groups * batch * 8, "nvl_lmhead_sample: workspace too small");
  NVL_REQUIRE(logits_out == nullptr || vocab_local % 4 == 0, "nvl_lmhead_sample: logits_out needs vocab %% 4 == 0");
  hipStream_t s = (hipStream_t)stream;
  uint32_t* partial = (uint32_t*)workspace;
  int rc = NVL_EINVAL;
#define NVL_LM_SB(MT_, NT_, SB_)                                                                                  \
  if (p.sb == SB_)                                                                                                \
    rc = launch_lm<MT_, NT_, SB_>(p, x, weight, temperatures, partial, logits_out, batch, vocab_local, k,         \
                                  col_offset, seed, offset, offset_dev, row_keys, s);
#define NVL_LM_CASE(MT_, NT_)                                                                                     \
  if (p.mt == MT_ && p.nt == NT_) { NVL_LM_SB(MT_, NT_, 8) NVL_LM_SB(MT_, NT_, 5) NVL_LM_SB(MT_, NT_, 3) }
  NVL_LM_CASE(1, 2) NVL_LM_CASE(2, 2) NVL_LM_CASE(3, 2) NVL_LM_CASE(5, 2) NVL_LM_CASE(7, 2) NVL_LM_CASE(9, 2)
  NVL_LM_CASE(12, 1)
#undef NVL_LM_CASE
#undef NVL_LM_SB
  if (rc != NVL_OK) {
    if (rc == NVL_EINVAL) nvl_set_error("nvl_lmhead_sample: internal plan error (mt=%d nt=%d)", p.mt, p.nt);
    return rc;
  }
  hipLaunchKernelGGL(lmhead_merge_kernel, dim3((unsigned)batch), dim3(256), 0, s, partial, p.groups, (int)batch, out,
                     (uint32_t*)best_packed);
  return nvl_check_launch("nvl_lmhead_sample");
}
