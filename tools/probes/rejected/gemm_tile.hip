// Tiled decode linear for gfx950 at 64 ... 256 rows: out[M, N] = x[M, K] . W[N, K]^T with BOTH operands staged through LDS
// by LDS-DMA and all four SIMDs of a CU on the matrix pipe. Second form of nvl_linear_wide (gemm_wide.hip) for the same
// reference call — F.linear in LinearBase.forward and its subclasses, nano-vllm layers/linear.py:54-156, with SiluAndMul
// (layers/activation.py:8-11) or the split-K slab hand-off as epilogue — on tile-packed weights.
//
// Why (round 5, profiles/r05_gemm_wide_streams.json). The streaming kernel of gemm_wide.hip keeps the weight stream in
// registers: 3 fat consumer waves own 3 SIMDs' whole register files, a 4th wave (the x loader) idles the 4th SIMD's
// matrix pipe, and a consumer is ALONE on its SIMD, so every ds_read -> MFMA and MFMA -> MFMA dependency is exposed.
// Switching its two memory streams off one at a time showed what that costs once there are many rows: with NO x stream
// and NO HBM weight stream the skeleton alone (MFMAs, fragment reads, barriers) takes 33 of the 47 us of the Qwen3-8B
// gate_up at 144 rows and 55 of 75 us at 256 rows — twice the 16 / 29 us its MFMAs need on three SIMDs. At these row
// counts the kernel is bound by its compute skeleton, not by HBM.
//
// This form:
//   * workgroup = 8 waves = 2 per SIMD (<= 256 registers each): wave (r, c) owns row half r (MTH row tiles of 16) x column
//     pair c (two 16-column tiles) of a [2 MTH x 16 rows] x [128 columns] output tile; every wave runs MFMAs, and a SIMD
//     always has a second wave to issue from while one waits for LDS or the matrix pipe.
//   * K advances in 64-column stages. A stage is an x tile [rows, 64] (128-byte rows, the 16-byte-slot XOR swizzle of
//     gemm_wide.hip's 64-column step: conflict-free ds_read_b128 B fragments) and a W tile of 8 column tiles x 2
//     k-blocks, each (tile, k-block) one contiguous KiB of the PACKED weight matrix in MFMA-A lane order — LDS-DMA lands
//     it lane-linear, the fragment read is lane x 16 bytes. No operand ever sits in a register across steps.
//   * loaders are the same waves: waves 0-3 stage the W tile (4 one-KiB pieces each per stage), waves 4-7 the x tile (MTH
//     pieces each) with global_load_lds_dwordx4. Two roles because a wave's loads retire in order: the W ring is as
//     deep as the LDS allows (4-8 stages: HBM latency), the x ring 3 stages (L2 latency), and each role waits with its
//     own counted vmcnt. The kernel has no compiler-visible global loads, so hipcc inserts no waits of its own.
//   * one barrier per stage: "stage s + 1 has landed in every wave's view" and "stage s may be overwritten" at once.
// v_mfma_f32_16x16x32_bf16, A = W fragment, B = x fragment: lane (l15, lq) ends up with out[16 mt + l15][tile + 4 lq .. + 3].
// Rounding points are the reference's (GEMM output rounded to bf16 before the activation), as in gemm_wide.hip.
#include "common.h"
#include "gemm_tile.h"
#include <stdlib.h>
#include <type_traits>

namespace {

enum { EPI_BF16 = 0, EPI_SILU = 1, EPI_PARTIAL = 2 };
constexpr int kLdsMax = 160 * 1024;

// Stage geometry. BK = k columns per stage (64: two 32-wide k-blocks; 32: one). x stage = [2 MTH x 16 rows][BK] bf16, W stage
// = 8 column tiles x BK / 32 k-blocks x 1 KiB. Rings: the x ring keeps NSX stages (L2 latency), the W ring whatever LDS
// is left (HBM latency: the more in flight the better), both as deep as 160 KiB allows — which is why many rows take the
// 32-column stage (a 256-row x stage of 64 columns is 32 KiB: three of them and four W stages fill the LDS with 32 KiB
// of weights in flight; at 32 columns 64 KiB of weights and 48 KiB of x are in flight).
__host__ __device__ constexpr int tile_xstage(int mth, int bk) { return 2 * mth * 16 * bk * 2; }
__host__ __device__ constexpr int tile_wstage(int bk) { return 8 * (bk / 32) * 1024; }
__host__ __device__ constexpr int tile_nsx(int mth, int bk) { return bk == 32 ? 5 : (mth >= 7 ? 3 : 4); }
__host__ __device__ constexpr int tile_nsw(int mth, int bk) {
  const int n = (kLdsMax - tile_nsx(mth, bk) * tile_xstage(mth, bk)) / tile_wstage(bk);
  return n > 16 ? 16 : n;
}
__host__ __device__ constexpr int tile_lds(int mth, int bk) {
  return tile_nsx(mth, bk) * tile_xstage(mth, bk) + tile_nsw(mth, bk) * tile_wstage(bk);
}

__device__ __forceinline__ float silu_f32(float g) { return g / (1.f + __expf(-g)); }

__device__ __forceinline__ void lds_dma_16(const bf16_t* src, unsigned dst) {
  unsigned keep;
  const unsigned d = __builtin_amdgcn_readfirstlane(dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(d)
               : "memory");
}

template <int MTH, int EPI, int BK>
__global__ __launch_bounds__(512) void linear_tile_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          void* __restrict__ out, int M, int N, int K, int steps, int dbg) {
  static_assert(BK == 32 || BK == 64, "k columns per stage");
  constexpr int KBS = BK / 32;                                     // k-blocks per stage
  constexpr int XB = tile_xstage(MTH, BK), WB = tile_wstage(BK);
  constexpr int NSX = tile_nsx(MTH, BK), NSW = tile_nsw(MTH, BK);
  constexpr int kRowB = BK * 2;                                    // bytes of an x row per stage
  constexpr int kPieceRows = 1024 / kRowB;                         // rows one 1-KiB LDS-DMA piece covers: 8 / 16
  constexpr int XP = 2 * MTH * 16 / kPieceRows;                    // x pieces per stage
  constexpr int NPX = (XP + 3) / 4, NPW = 2 * KBS;                 // pieces per stage of an x wave / a W wave
  static_assert(NSX >= 3 && NSW >= 3 && (NSW - 3) * NPW < 64 && (NSX - 3) * NPX < 64, "ring depths / vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = wave >> 2, c = wave & 3;
  const int l15 = lane & 15, lq = lane >> 4;
  const int out_cols = EPI == EPI_SILU ? N / 2 : N;
  const int ntiles = out_cols >> 4;
  // global 16-column tile of the workgroup's local tile t (0 .. 7): plain order, or (SiLU) gate tile / matching up tile
  // pairs; a ragged last workgroup reads any valid tile (never stored)
  auto gtile = [&](int t) {
    if (EPI == EPI_SILU) {
      int pair = (int)blockIdx.x * 4 + (t >> 1);
      pair = pair < ntiles ? pair : ntiles - 1;
      return (t & 1) ? ntiles + pair : pair;
    }
    const int tt = (int)blockIdx.x * 8 + t;
    return tt < ntiles ? tt : ntiles - 1;
  };
  const int64_t k0 = (int64_t)blockIdx.y * steps * BK;
  // workgroups start their K walk at different stages and wrap (rows of W are K * 2 bytes apart: lock-step walkers would
  // all sit on the same HBM channels), as in gemm_wide.hip
  const int rot = (int)(((unsigned)blockIdx.x + 3u * blockIdx.y) % (unsigned)steps);
  auto kstep = [&](int s) {
    s += rot;
    return s >= steps ? s - steps : s;
  };

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned xl0 = lds0, wl0 = lds0 + NSX * XB;
  // x swizzle: the 16-byte slot of chunk q in row R. 128-byte rows (BK = 64): q ^ ((R >> 1) & 7), two rows share a
  // 256-byte bank row; 64-byte rows (BK = 32): q ^ g[(R >> 2) & 3] with g = {0, 2, 3, 1}, four rows share one — in both
  // the 16 lanes ds_read_b128 serves together hit 16 different slots (tests/test_wide_gemm_lds_mapping.py).
  auto swz = [](int row) { return BK == 64 ? ((row >> 1) & 7) : ((0x1320 >> (((row >> 2) & 3) * 4)) & 3); };
  // ---- this wave's share of a stage: waves 0-3 stage the W tile, waves 4-7 the x tile ---------------------------------------
  const bf16_t* srcw[NPW];
  const bf16_t* srcx[NPX];
  if (r == 0) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = c + 4 * i, t = p / KBS, kb = p % KBS;          // piece = (local tile, k-block of the stage)
      srcw[i] = w + (int64_t)gtile(t) * 16 * K + ((k0 >> 5) + kb) * 512 + lane * 8;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      int j = c + 4 * i;                                           // piece = kPieceRows rows
      j = j < XP ? j : XP - 1;                                     // (a wave without a piece of its own repeats the last one)
      const int per_row = kRowB / 16;                              // 16-byte slots per row: 8 / 4
      const int row = kPieceRows * j + lane / per_row, slot = lane % per_row;
      const int grow = row < M ? row : M - 1;                      // padding rows read a valid row (never stored)
      srcx[i] = x + (int64_t)grow * K + k0 + ((slot ^ swz(row)) << 3);
    }
  }
  // (measurement switches, NVL_WIDE_DBG: bit 0 = no x stream after the prologue, bit 1 = no weight stream after it —
  //  each stream alone inside the real pipeline; results are garbage)
  auto issue = [&](int s) {                                        // logical stage s < steps
    const int ks = kstep(s);
    if (s >= (r == 0 ? NSW : NSX) - 1 && (dbg & (r == 0 ? 2 : 1))) return;
    if (r == 0) {
      const unsigned dst = wl0 + (unsigned)(s % NSW) * WB;
#pragma unroll
      for (int i = 0; i < NPW; ++i) lds_dma_16(srcw[i] + (int64_t)ks * (KBS * 512), dst + (unsigned)(c + 4 * i) * 1024);
    } else {
      const unsigned dst = xl0 + (unsigned)(s % NSX) * XB;
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int j = c + 4 * i < XP ? c + 4 * i : XP - 1;
        lds_dma_16(srcx[i] + ks * BK, dst + (unsigned)j * 1024);
      }
    }
  };
  // "stage s + 1 has landed" with stages s + 1 ... s + NS - 2 of this wave's role in flight: all but the oldest may stay
  auto wait_next = [&](bool full_window) {
    if (full_window) {
      if (r == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSW - 3) * NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSX - 3) * NPX) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  const int ns = r == 0 ? NSW : NSX;

  // ---- prologue: NS - 2 stages in flight (the other two ring slots are the one being read and the one about to be) ---------
  for (int s = 0; s < ns - 2 && s < steps; ++s) issue(s);
  wait_next(steps >= ns - 2);                                      // (stage 0: with a full window NS - 3 newer ones stay)
  __builtin_amdgcn_s_barrier();
  if (ns - 2 < steps) issue(ns - 2);

  // fragment offsets: x row l15 of a row tile at its swizzled slot; W lane x 16 inside the (tile, k-block) KiB
  int xoff[KBS];
#pragma unroll
  for (int kb = 0; kb < KBS; ++kb) xoff[kb] = (r * MTH * 16 + l15) * kRowB + (((kb * 4 + lq) ^ swz(l15)) << 4);
  const int woff = (2 * c) * KBS * 1024 + lane * 16;

  f32x4_t acc[MTH][2];
#pragma unroll
  for (int i = 0; i < MTH; ++i) acc[i][0] = acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Software pipeline over (stage, k-block) units: the fragments of unit u + 1 are read from LDS under the MFMAs of unit
  // u, also across the stage boundary — the barrier that opens stage s + 1 sits BEFORE the last k-block of stage s, so
  // no LDS round trip is ever exposed (the first version read, waited, computed: its skeleton alone — no memory stream
  // at all — ran the matrix pipe at ~50 %: profiles/r05_gemm_tile_streams.json).
  u32x4_t xf[2][MTH], wf[2][2];
  auto fread = [&](int buf, int xs, int ws, int kb) {
    const unsigned char* xt = smem + xs * XB;
    const unsigned char* wt = smem + NSX * XB + ws * WB + woff;
    wf[buf][0] = *reinterpret_cast<const u32x4_t*>(wt + kb * 1024);
    wf[buf][1] = *reinterpret_cast<const u32x4_t*>(wt + KBS * 1024 + kb * 1024);
#pragma unroll
    for (int i = 0; i < MTH; ++i) xf[buf][i] = *reinterpret_cast<const u32x4_t*>(xt + i * 16 * kRowB + xoff[kb]);
  };
  auto mfmas = [&](int buf) {
#pragma unroll
    for (int i = 0; i < MTH; ++i) {
      acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[buf][0]),
                                                          __builtin_bit_cast(bf16x8_t, xf[buf][i]), acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[buf][1]),
                                                          __builtin_bit_cast(bf16x8_t, xf[buf][i]), acc[i][1], 0, 0, 0);
    }
  };
  auto interleave = [&]() {                                        // MTH + 2 LDS reads among 2 MTH MFMAs, reads first
#pragma unroll
    for (int j = 0; j < MTH + 2; ++j) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * MTH - (MTH + 2), 0);
  };

  int xs = 0, ws = 0;                                              // ring slots of the current stage
  // one unit = one k-block held in fragment buffer BUF (a compile-time constant: the loops below alternate 0 / 1); it
  // reads the next unit into the other buffer. The last k-block of a stage first opens the next stage: every wave's
  // pieces of it have landed (own counted wait + barrier), and the slot of stage s - 1 is free for stage s + NS - 1.
  auto unit = [&](auto buf_c, int s, int kb) {
    constexpr int B = decltype(buf_c)::value;
    const bool last_kb = kb == KBS - 1;
    int nxs = xs, nws = ws, nkb = kb + 1;
    if (last_kb) {
      wait_next(s + ns - 2 < steps);
      __builtin_amdgcn_s_barrier();
      if (s + ns - 1 < steps) issue(s + ns - 1);
      nxs = xs + 1 == NSX ? 0 : xs + 1;
      nws = ws + 1 == NSW ? 0 : ws + 1;
      nkb = 0;
    }
    if (!last_kb || s + 1 < steps) {
      fread(B ^ 1, nxs, nws, nkb);
      mfmas(B);
      interleave();
    } else {
      mfmas(B);
    }
    xs = nxs;
    ws = nws;
    __builtin_amdgcn_sched_barrier(0);
  };
  fread(0, 0, 0, 0);
  if constexpr (KBS == 2) {
    for (int s = 0; s < steps; ++s) {
      unit(std::integral_constant<int, 0>{}, s, 0);
      unit(std::integral_constant<int, 1>{}, s, 1);
    }
  } else {
    for (int s = 0; s < steps; s += 2) {
      unit(std::integral_constant<int, 0>{}, s, 0);
      if (s + 1 < steps) unit(std::integral_constant<int, 1>{}, s + 1, 0);
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < MTH; ++i) {
    const int m = (r * MTH + i) * 16 + l15;
    if (m >= M) continue;
    if constexpr (EPI == EPI_SILU) {
      const int n = ((int)blockIdx.x * 4 + c) * 16;
      if (n >= out_cols) continue;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = silu_f32(round_bf16(acc[i][0][e])) * round_bf16(acc[i][1][e]);
      *reinterpret_cast<u32x2_t*>((bf16_t*)out + (int64_t)m * out_cols + n + lq * 4) =
          u32x2_t{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    } else {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int n = ((int)blockIdx.x * 8 + 2 * c + nt) * 16;
        if (n >= out_cols) continue;
        if constexpr (EPI == EPI_BF16) {
          *reinterpret_cast<u32x2_t*>((bf16_t*)out + (int64_t)m * N + n + lq * 4) =
              u32x2_t{pack_bf16x2(acc[i][nt][0], acc[i][nt][1]), pack_bf16x2(acc[i][nt][2], acc[i][nt][3])};
        } else {
          *reinterpret_cast<f32x4_t*>((float*)out + ((int64_t)blockIdx.y * M + m) * N + n + lq * 4) = acc[i][nt];
        }
      }
    }
  }
}

// k columns per stage by row count: 32 from 11 row tiles on (see tile_nsw); NVL_WIDE_TILE_BK=32|64 forces one (A/B)
int tile_bk(int mth) {
  static const int forced = [] { const char* e = getenv("NVL_WIDE_TILE_BK"); return e ? atoi(e) : 0; }();
  if (forced == 32 || forced == 64) return forced;
  return mth >= 6 ? 32 : 64;
}

template <int MTH, int EPI, int BK>
int launch_tile(const void* x, const void* w, void* out, int64_t m, int n, int k, int split, hipStream_t s) {
  constexpr int lds = tile_lds(MTH, BK);
  static_assert(lds <= kLdsMax, "stage rings exceed the LDS");
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& attr_set = attr_done[nvl_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_tile_kernel<MTH, EPI, BK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      nvl_set_error("nvl_linear_wide (tiled form): cannot reserve %d B of LDS", lds);
      return NVL_ELAUNCH;
    }
    attr_set = true;
  }
  static const int dbg = [] { const char* e = getenv("NVL_WIDE_DBG"); return e ? atoi(e) : 0; }();
  const int out_cols = EPI == EPI_SILU ? n / 2 : n;
  const int per_wg = EPI == EPI_SILU ? 64 : 128;
  const unsigned gx = (unsigned)((out_cols + per_wg - 1) / per_wg);
  hipLaunchKernelGGL((linear_tile_kernel<MTH, EPI, BK>), dim3(gx, split), dim3(512), lds, s, (const bf16_t*)x,
                     (const bf16_t*)w, out, (int)m, n, k, k / BK / split, dbg);
  return NVL_OK;
}

template <int EPI>
int dispatch_tile(int mth, const void* x, const void* w, void* out, int64_t m, int n, int k, int split, hipStream_t s) {
  const int bk = tile_bk(mth);
#define NVL_T_CASE(V)                                                                     \
  case V:                                                                                 \
    return bk == 32 ? launch_tile<V, EPI, 32>(x, w, out, m, n, k, split, s)              \
                    : launch_tile<V, EPI, 64>(x, w, out, m, n, k, split, s);
  switch (mth) {
    NVL_T_CASE(2) NVL_T_CASE(3) NVL_T_CASE(4) NVL_T_CASE(5) NVL_T_CASE(6) NVL_T_CASE(7) NVL_T_CASE(8)
  }
#undef NVL_T_CASE
  nvl_set_error("nvl_linear_wide (tiled form): internal plan error (mth=%d)", mth);
  return NVL_EINVAL;
}

}  // namespace

bool nvl_tile_covers(int64_t m, int n, int k, int mode) {
  if (m < 33 || m > 256 || k % 64 || k < 256) return false;             // 3 ... 16 row tiles; >= 4 stages of 64 columns
  return mode == EPI_SILU ? n % 32 == 0 : n % 16 == 0;
}

int nvl_tile_workgroups(int n, int mode) { return mode == EPI_SILU ? (n / 2 + 63) / 64 : (n + 127) / 128; }

int nvl_tile_launch(const void* x, const void* w_packed, void* out, int64_t m, int n, int k, int mode, int split,
                    void* stream) {
  const int mth = (int)((m + 31) / 32) < 2 ? 2 : (int)((m + 31) / 32);
  hipStream_t s = (hipStream_t)stream;
  if (mode == EPI_BF16) return dispatch_tile<EPI_BF16>(mth, x, w_packed, out, m, n, k, split, s);
  if (mode == EPI_SILU) return dispatch_tile<EPI_SILU>(mth, x, w_packed, out, m, n, k, split, s);
  return dispatch_tile<EPI_PARTIAL>(mth, x, w_packed, out, m, n, k, split, s);
}
