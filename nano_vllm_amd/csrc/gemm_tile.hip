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

namespace {

enum { EPI_BF16 = 0, EPI_SILU = 1, EPI_PARTIAL = 2 };
constexpr int kBK = 64;                      // k columns per stage
constexpr int kWStage = 8 * 2 * 1024;        // 8 column tiles x 2 k-blocks x 1 KiB
constexpr int kNSX = 3;                      // x stages (two in flight)
constexpr int kLdsMax = 160 * 1024;

__host__ __device__ constexpr int tile_xstage(int mth) { return 2 * mth * 16 * kBK * 2; }
__host__ __device__ constexpr int tile_nsw(int mth) {
  const int n = (kLdsMax - kNSX * tile_xstage(mth)) / kWStage;
  return n > 8 ? 8 : n;
}
__host__ __device__ constexpr int tile_lds(int mth) { return kNSX * tile_xstage(mth) + tile_nsw(mth) * kWStage; }

__device__ __forceinline__ float silu_f32(float g) { return g / (1.f + __expf(-g)); }

__device__ __forceinline__ void lds_dma_16(const bf16_t* src, unsigned dst) {
  unsigned keep;
  const unsigned d = __builtin_amdgcn_readfirstlane(dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(d)
               : "memory");
}

template <int MTH, int EPI>
__global__ __launch_bounds__(512) void linear_tile_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          void* __restrict__ out, int M, int N, int K, int steps) {
  constexpr int XB = tile_xstage(MTH);
  constexpr int NSW = tile_nsw(MTH);
  constexpr int AX = kNSX - 1, AW = NSW - 1;                       // stages in flight per role
  static_assert(NSW >= 3 && (AW - 1) * 4 < 64 && (AX - 1) * MTH < 64, "ring depths / vmcnt is a 6-bit counter");
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
  const int64_t k0 = (int64_t)blockIdx.y * steps * kBK;
  // workgroups start their K walk at different stages and wrap (rows of W are K * 2 bytes apart: lock-step walkers would
  // all sit on the same HBM channels), as in gemm_wide.hip
  const int rot = (int)(((unsigned)blockIdx.x + 3u * blockIdx.y) % (unsigned)steps);
  auto kstep = [&](int s) {
    s += rot;
    return s >= steps ? s - steps : s;
  };

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned xl0 = lds0, wl0 = lds0 + kNSX * XB;
  // ---- this wave's share of a stage: W pieces c, c + 4, c + 8, c + 12 (r == 0) or x pieces c, c + 4, ... (r == 1) --------
  constexpr int NPW = 4, NPX = MTH;
  const bf16_t* srcw[NPW];
  const bf16_t* srcx[NPX];
  if (r == 0) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = c + 4 * i, t = p >> 1, kb = p & 1;             // piece = (local tile, k-block of the stage)
      srcw[i] = w + (int64_t)gtile(t) * 16 * K + ((k0 >> 5) + kb) * 512 + lane * 8;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int j = c + 4 * i;                                     // piece = rows 8 j .. 8 j + 7 (128 bytes each)
      const int row = 8 * j + (lane >> 3), slot = lane & 7;
      const int grow = row < M ? row : M - 1;                      // padding rows read a valid row (never stored)
      srcx[i] = x + (int64_t)grow * K + k0 + ((slot ^ ((row >> 1) & 7)) << 3);
    }
  }
  auto issue = [&](int s) {                                        // logical stage s < steps
    const int ks = kstep(s);
    if (r == 0) {
      const unsigned dst = wl0 + (unsigned)(s % NSW) * kWStage;
#pragma unroll
      for (int i = 0; i < NPW; ++i) lds_dma_16(srcw[i] + (int64_t)ks * 1024, dst + (unsigned)(c + 4 * i) * 1024);
    } else {
      const unsigned dst = xl0 + (unsigned)(s % kNSX) * XB;
#pragma unroll
      for (int i = 0; i < NPX; ++i) lds_dma_16(srcx[i] + ks * kBK, dst + (unsigned)(c + 4 * i) * 1024);
    }
  };
  // "the next stage has landed": at most the newest A - 1 stages of this wave's role are still in flight
  auto wait_next = [&](bool full_window) {
    if (full_window) {
      if (r == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AW - 1) * NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AX - 1) * NPX) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  // ---- prologue: fill the rings, wait for stage 0 -----------------------------------------------------------------------
  const int ahead = r == 0 ? AW : AX;
  for (int s = 0; s < ahead && s < steps; ++s) issue(s);
  wait_next(steps >= ahead);
  __builtin_amdgcn_s_barrier();

  // fragment offsets inside a stage: x row l15 of a row tile, 16-byte slot (4 kb + lq) ^ ((l15 >> 1) & 7); W lane x 16
  int xoff[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) xoff[kb] = (r * MTH * 16 + l15) * (kBK * 2) + (((kb * 4 + lq) ^ ((l15 >> 1) & 7)) << 4);
  const int woff = (2 * c) * 2 * 1024 + lane * 16;

  f32x4_t acc[MTH][2];
#pragma unroll
  for (int i = 0; i < MTH; ++i) acc[i][0] = acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int xs = 0, ws = 0;                                              // ring slots of the current stage
  for (int s = 0; s < steps; ++s) {
    if (s + ahead < steps) issue(s + ahead);                       // (its slot was read during stage s - 1: free since the barrier)
    const unsigned char* xt = smem + xs * XB;
    const unsigned char* wt = smem + kNSX * XB + ws * kWStage + woff;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const bf16x8_t w0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wt + kb * 1024));
      const bf16x8_t w1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wt + 2048 + kb * 1024));
#pragma unroll
      for (int i = 0; i < MTH; ++i) {
        const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(xt + i * 16 * (kBK * 2) + xoff[kb]));
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xf, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xf, acc[i][1], 0, 0, 0);
      }
    }
    wait_next(s + ahead < steps);
    __builtin_amdgcn_s_barrier();
    xs = xs + 1 == kNSX ? 0 : xs + 1;
    ws = ws + 1 == NSW ? 0 : ws + 1;
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

template <int MTH, int EPI>
int launch_tile(const void* x, const void* w, void* out, int64_t m, int n, int k, int split, hipStream_t s) {
  constexpr int lds = tile_lds(MTH);
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& attr_set = attr_done[nvl_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_tile_kernel<MTH, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      nvl_set_error("nvl_linear_wide (tiled form): cannot reserve %d B of LDS", lds);
      return NVL_ELAUNCH;
    }
    attr_set = true;
  }
  const int out_cols = EPI == EPI_SILU ? n / 2 : n;
  const int per_wg = EPI == EPI_SILU ? 64 : 128;
  const unsigned gx = (unsigned)((out_cols + per_wg - 1) / per_wg);
  hipLaunchKernelGGL((linear_tile_kernel<MTH, EPI>), dim3(gx, split), dim3(512), lds, s, (const bf16_t*)x,
                     (const bf16_t*)w, out, (int)m, n, k, k / kBK / split);
  return NVL_OK;
}

template <int EPI>
int dispatch_tile(int mth, const void* x, const void* w, void* out, int64_t m, int n, int k, int split, hipStream_t s) {
  switch (mth) {
    case 2: return launch_tile<2, EPI>(x, w, out, m, n, k, split, s);
    case 3: return launch_tile<3, EPI>(x, w, out, m, n, k, split, s);
    case 4: return launch_tile<4, EPI>(x, w, out, m, n, k, split, s);
    case 5: return launch_tile<5, EPI>(x, w, out, m, n, k, split, s);
    case 6: return launch_tile<6, EPI>(x, w, out, m, n, k, split, s);
    case 7: return launch_tile<7, EPI>(x, w, out, m, n, k, split, s);
    case 8: return launch_tile<8, EPI>(x, w, out, m, n, k, split, s);
  }
  nvl_set_error("nvl_linear_wide (tiled form): internal plan error (mth=%d)", mth);
  return NVL_EINVAL;
}

}  // namespace

bool nvl_tile_covers(int64_t m, int n, int k, int mode) {
  if (m < 33 || m > 256 || k % kBK || k < 4 * kBK) return false;       // 3 ... 16 row tiles; >= 4 stages
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
