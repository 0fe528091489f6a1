// Skinny ("decode") linear layers for gfx950: out[M, N] = x[M, K] . W[N, K]^T with M <= a few
// hundred rows (one row per running sequence), bf16 in, fp32 accumulate on MFMA.
// Replaces F.linear as called from LinearBase.forward (nano-vllm layers/linear.py:54-156) on the
// decode step, where the library GEMM spends 8-15 us per call on 4-13 MB of weights, with the
// reference's activation / residual glue folded into the epilogues.
//
// What bounds this shape on MI355X is not HBM (weights stream once, non-temporal) but the
// L2 -> CU ingest of the activations: every workgroup needs all M rows of its K range, and a CU
// ingests at most ~50 B/clk (tools/probes/l2_read_probe.hip). So the decomposition minimises bytes
// per CU at ~256 workgroups: a workgroup = 8 waves owns NT = 1-2 sixteen-column output tiles for
// one M-group (<= 8-16 row tiles); its 8 waves split the K range (no weight byte is loaded twice
// inside a workgroup, every x fragment feeds NT MFMAs), keep their W fragments in registers, and
// merge their partial accumulators through LDS at the end. x is fetched in whole contiguous row
// segments (KB*64 B per row; fragment-shaped 16 x 64 B loads ingest 3-4x slower) into a per-wave
// 2-slot LDS ring and read back as MFMA B fragments with ds_read_b128. N = 1024 projections
// (o_proj, down_proj) have too few column tiles, so they are additionally split over K across
// workgroups and emit fp32 partial slabs; the consumer (nvl_add_rmsnorm_splitk) sums the slabs in
// its prologue — "reduce at the launch boundary", cdna_hip_programming.md §5 — so no in-launch
// cross-workgroup hand-off is needed.
//
// MFMA: v_mfma_f32_16x16x32_bf16 with A = W fragment (A[i][k] = W[n0+i][k0+k]) and B = x fragment
// (B[k][j] = x[m0+j][k0+k]): lane l holds A[l&15][8*(l>>4)..+7], B[8*(l>>4)..+7][l&15] and
// D[4*(l>>4)+r][l&15], i.e. 4 consecutive output columns of one row per lane => 8-byte bf16 /
// 16-byte fp32 stores.
//
// Epilogues (rounding points are the reference's: the GEMM output is rounded to bf16 first):
//   BF16    out[m, n]            = bf16(acc)
//   SILU    out[m, j]            = bf16(silu(bf16(acc[gate j])) * bf16(acc[up j]))   (activation.py:8-11)
//   PARTIAL part[split][m, n]    = acc (fp32), summed and rounded by the consumer
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kNW = 8;          // waves per workgroup (K-split inside the workgroup), 2 per SIMD
enum { EPI_BF16 = 0, EPI_SILU = 1, EPI_PARTIAL = 2 };

__device__ __forceinline__ float silu_f32(float g) { return g / (1.f + __expf(-g)); }

// LDS image of one 16-row x (KB*64-byte) x-slice tile: row stride padded by 16 B so the 16 rows of
// a ds_read_b128 fragment read fall on 16 different 16-byte slots (<= 2-way conflicts).
template <int KB>
struct XTile {
  static constexpr int kRowBytes = KB * 64;
  static constexpr int kStride = kRowBytes + 16;
  static constexpr int kBytes = 16 * kStride;
  static constexpr int kLanesPerRow = KB * 4;      // 16-byte chunks per row
};

template <int MT, int KB, int NT, int EPI>
__global__ __launch_bounds__(kNW * 64) void linear_decode_kernel(const bf16_t* __restrict__ x,
                                                                  const bf16_t* __restrict__ w,
                                                                  void* __restrict__ out, int M, int N, int K,
                                                                  int ko_iters, int packed) {
  static_assert(EPI != EPI_SILU || NT == 2, "SILU pairs a gate tile with an up tile");
  constexpr int T = MT * NT;
  // NT == 1: two accumulation chains per tile (even / odd k blocks) hide the MFMA dependency latency;
  // NT == 2 already has two independent chains.
  constexpr bool kTwoChains = NT == 1;
  constexpr int KH = kTwoChains ? (KB + 1) / 2 : KB;
  using XT = XTile<KB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4_t* red = reinterpret_cast<f32x4_t*>(smem_raw);   // [kNW / 2][T][64]   (after the main loop)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int n_tile = blockIdx.x, split = blockIdx.y;
  const int m_base = blockIdx.z * (MT * 16);
  const int out_cols = EPI == EPI_SILU ? N / 2 : N;
  unsigned char* xlds = smem_raw + wave * (2 * XT::kBytes);   // this wave's private 2-slot tile ring

  // Weight fragments. Row-major W: lane (l15, lq) reads 16 B of row l15 — every 16-lane group of the wave touches 16
  // different 128-byte lines, which the CU's address path retires at ~15 B/clk. `packed` (nvl_pack_weight_tiles: the
  // 16 x 32 sub-matrix of a (tile, k-block) stored as ONE contiguous KiB in lane order): the same fragment is a
  // contiguous wave load, a quarter of the address-path time. Both are affine in the k block: base + kw * kws + kb * kbs.
  const int kbs = packed ? 512 : 32;               // elements between consecutive 32-wide k blocks
  const int kws = packed ? 16 : 1;                 // elements per unit of k
  const bf16_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    // SILU: tile 0 = gate columns, tile 1 = the matching up columns; otherwise NT adjacent tiles
    const int tile = EPI == EPI_SILU ? nt * (out_cols >> 4) + n_tile : n_tile * NT + nt;
    wrow[nt] = packed ? w + (int64_t)tile * 16 * K + lane * 8 : w + (int64_t)(tile * 16 + l15) * K + lq * 8;
  }

  // x staging geometry: a tile is 64*KB 16-byte chunks; load instruction i moves chunks i*64 + lane, i.e.
  // whole contiguous row segments (KB*64 B per row) — 3-4x the L2->CU rate of fragment-shaped 16 x 64 B
  // loads on this chip (tools/probes/l2_read_probe.hip: 45-60 vs 15 B/clk/CU).
  int xoff[KB], wr_off[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) {
    const int c = i * 64 + lane;
    const int r = c / XT::kLanesPerRow;
    const int ccol = c - r * XT::kLanesPerRow;
    wr_off[i] = r * XT::kStride + ccol * 16;
    xoff[i] = r * K + ccol * 8;                    // element offset inside the tile's slice
  }
  const int rd_off = l15 * XT::kStride + lq * 16;

  f32x4_t acc[MT][NT], acc2[kTwoChains ? MT : 1][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (kTwoChains) acc2[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

  const int64_t xlimit = (int64_t)M * K - 8;       // rows >= M (ragged last tile) read any valid address

  const int64_t k_wg = (int64_t)split * ko_iters * (kNW * KB * 32);
  for (int ko = 0; ko < ko_iters; ++ko) {
    const int64_t kw = k_wg + (int64_t)(ko * kNW + wave) * (KB * 32);
    u32x4_t wf[NT][KB];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        wf[nt][kb] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wrow[nt] + kw * kws + kb * kbs));

    auto mfma_tile = [&](int mt, const u32x4_t* f) {
#pragma unroll
      for (int kb = 0; kb < KH; ++kb)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[nt][kb]),
                                                                __builtin_bit_cast(bf16x8_t, f[kb]), acc[mt][nt], 0, 0,
                                                                0);
          if (kTwoChains && kb + KH < KB)
            acc2[kTwoChains ? mt : 0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[nt][kb + KH]),
                                                                   __builtin_bit_cast(bf16x8_t, f[kb + KH]),
                                                                   acc2[kTwoChains ? mt : 0][nt], 0, 0, 0);
        }
    };
    u32x4_t g[2][KB];
    auto gload = [&](int mt, u32x4_t* dst) {
      const int64_t base = (int64_t)(m_base + mt * 16) * K + kw;
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        int64_t off = base + xoff[i];
        off = off < xlimit ? off : xlimit;
        dst[i] = *reinterpret_cast<const u32x4_t*>(x + off);
      }
    };
    auto lwrite = [&](int slot, const u32x4_t* src) {
#pragma unroll
      for (int i = 0; i < KB; ++i) *reinterpret_cast<u32x4_t*>(xlds + slot * XT::kBytes + wr_off[i]) = src[i];
    };

    // software pipeline (order pinned with sched_barrier: hipcc otherwise sinks every load to its use):
    //   global loads run 2-3 tiles ahead in registers, the LDS image 1 tile ahead, and within an
    //   iteration the fragment reads of tile mt are issued BEFORE the LDS writes of tile mt+1.
    gload(0, g[0]);
    if (MT > 1) gload(1, g[1]);
    __builtin_amdgcn_sched_barrier(0);
    lwrite(0, g[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (MT > 2) gload(2, g[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      u32x4_t f[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        f[kb] = *reinterpret_cast<const u32x4_t*>(xlds + (mt & 1) * XT::kBytes + rd_off + kb * 64);
      __builtin_amdgcn_sched_barrier(0);
      if (mt + 1 < MT) lwrite((mt + 1) & 1, g[(mt + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (mt + 3 < MT) gload(mt + 3, g[(mt + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_tile(mt, f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      if (kTwoChains) acc[mt][nt] += acc2[mt][nt];
  __syncthreads();   // all waves are done with their tile rings: the LDS is reused for the merge

  // merge the 8 waves' partial tiles through LDS in two stages (4T KiB of LDS instead of 8T):
  // waves 4-7 hand their tiles to waves 0-3, then wave w finishes M-tiles mt == w (mod 8).
  constexpr int HW = kNW / 2;
  if (wave >= HW) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) red[((wave - HW) * T + mt * NT + nt) * 64 + lane] = acc[mt][nt];
  }
  __syncthreads();
  if (wave < HW) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4_t* slot = red + (wave * T + mt * NT + nt) * 64 + lane;
        *slot = acc[mt][nt] + *slot;
      }
  }
  __syncthreads();

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if ((mt % kNW) != wave) continue;
    f32x4_t v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      v[nt] = red[(mt * NT + nt) * 64 + lane];
#pragma unroll
      for (int ww = 1; ww < HW; ++ww) v[nt] += red[(ww * T + mt * NT + nt) * 64 + lane];
    }
    const int m = m_base + mt * 16 + l15;
    if (m >= M) continue;
    if constexpr (EPI == EPI_BF16) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = (n_tile * NT + nt) * 16 + lq * 4;
        u32x2_t o = {pack_bf16x2(v[nt][0], v[nt][1]), pack_bf16x2(v[nt][2], v[nt][3])};
        *reinterpret_cast<u32x2_t*>((bf16_t*)out + (int64_t)m * N + n) = o;
      }
    } else if constexpr (EPI == EPI_SILU) {
      const int n = n_tile * 16 + lq * 4;
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = silu_f32(round_bf16(v[0][i])) * round_bf16(v[1][i]);
      u32x2_t ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
      *reinterpret_cast<u32x2_t*>((bf16_t*)out + (int64_t)m * out_cols + n) = ov;
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = (n_tile * NT + nt) * 16 + lq * 4;
        *reinterpret_cast<f32x4_t*>((float*)out + ((int64_t)split * M + m) * N + n) = v[nt];
      }
    }
  }
}


struct Plan {
  int kb, ko, split, nt, mt, mgroups;
};
thread_local int g_packed = 0;       // weight layout of the call being dispatched (set by nvl_linear_decode)

// Deep-K shapes (several passes of the per-wave K range: Qwen3-8B / 32B projections) are reported as NOT covered: with
// 16-32 output columns per workgroup every workgroup re-reads all of x (768 workgroups x 144 x 4096 x 2 B = 906 MB of
// L2 traffic for 201 MB of weights on the 8B gate_up), which a multi-pass variant of this kernel ran at 1.6-2.2 TB/s
// (profiles/r02_gemm_deep_*.json, removed). Those shapes belong to the wide-tile kernel (gemm_wide.hip).
// K range per wave = K / (split * 8) must be KB * KO * 32 with KB in {1, 2, 3, 4}.
bool make_plan(int64_t m, int n, int k, int mode, Plan* p) {
  const int out_cols = mode == EPI_SILU ? n / 2 : n;
  if (m < 1 || m > 4096 || n < 16 || k < 256 || out_cols % 16 || (mode == EPI_SILU && n % 32)) return false;
  const int tiles = out_cols / 16;
  const int mtiles = (int)((m + 15) / 16);
  if (k > 4096 || (k > 1024 && mode != EPI_PARTIAL) || (k > 3072 && mode == EPI_PARTIAL)) return false;   // deep K
  // two column tiles per workgroup (every x fragment feeds 2 MFMAs) and M split in >= 2 groups once there
  // are enough rows: halves the x bytes a CU ingests at the same workgroup count
  // (round 4 sweep of every (NT, row groups, split) decomposition on the four Qwen3-0.6B projections at 64 / 131 / 208 rows,
  // GEMM + the add-RMSNorm that sums its slabs: tools/gemm_skinny_sweep.py, profiles/r04_gemm_skinny_sweep.jsonl — the
  // two-tile / two-row-group form already pays from 4 row tiles on: qkv 5.5 -> 4.9 us, down 7.5 -> 7.1, o 6.7 -> 6.5 at 64 rows)
  int nt = 1;
  if (mode == EPI_SILU || (mtiles >= 4 && tiles % 2 == 0)) nt = 2;
  int mgroups = (mtiles + 15) / 16;
  const int col_wgs = mode == EPI_SILU ? tiles : tiles / nt;
  if (nt == 2 && mode != EPI_SILU) {
    // few column workgroups (layer projections): a second row group halves the x bytes per CU at the same
    // workgroup count; many (lm_head: 4,748): the matrix must not be streamed twice — one group of <= 9 row tiles
    if (col_wgs < 1024) { if (mgroups < 2) mgroups = 2; }
    else mgroups = (mtiles + 8) / 9;
  }
  // (measured, round 4: a second row group for the SiLU launch as well — 384 workgroups of 5 row tiles instead of 192
  // of 9 — is SLOWER, 12.0 vs 9.6 us at 131 rows: profiles/r04_gemm_silu_two_row_groups.json)
  // o_proj-like slab outputs (K <= 2048) from 9 row tiles on: THREE row groups x a 2-way K split (192 workgroups, two
  // slabs for the norm to sum) beat two groups x 4-way (256 workgroups, four slabs): 8.2 -> 7.7 us at 131 rows, 9.6 -> 9.2
  // at 208 (same sweep); deeper K (down_proj, 3072) has no single-pass 2-way split and stays as it was
  // (only while three groups keep a group within the 16 instantiated row tiles; beyond 48 row tiles the ceil(mtiles / 16)
  // groups of the general rule stand)
  if (mode == EPI_PARTIAL && nt == 2 && col_wgs < 1024 && mtiles >= 9 && mtiles <= 48 && k <= 2048 && k % (2 * kNW * 32) == 0)
    mgroups = 3;
  int split = 1;
  if (mode == EPI_PARTIAL) {                      // fill the chip: ~256 workgroups
    while (split < 8 && col_wgs * mgroups * split * 2 <= 256 && k % (split * 2 * kNW * 32) == 0) split *= 2;
  }
  // A/B override for tools/gemm_skinny_sweep.py: NVL_SKINNY_PLAN="nt,mgroups,split" (0 = keep the rule's value). Read ONCE
  // per process (this function is on the decode hot path, and a caller may have sized its slab scratch from an earlier
  // answer); a forced value that breaks an invariant of the rule makes the shape "not covered" instead of a bad launch.
  static const struct Forced { int nt = 0, mg = 0, sp = 0; Forced() { if (const char* e = getenv("NVL_SKINNY_PLAN")) sscanf(e, "%d,%d,%d", &nt, &mg, &sp); } } forced;
  if (forced.nt == 1 || forced.nt == 2) {
    if (mode == EPI_SILU && forced.nt != 2) return false;
    if (forced.nt == 2 && mode != EPI_SILU && tiles % 2) return false;
    nt = forced.nt;
  }
  if (forced.mg > 0) mgroups = forced.mg;
  if (forced.sp > 0 && mode == EPI_PARTIAL) {
    if (forced.sp != 1 && forced.sp != 2 && forced.sp != 4 && forced.sp != 8) return false;
    split = forced.sp;
  }
  if (mgroups > mtiles) return false;
  if (k % (split * kNW * 32)) return false;
  const int kw = k / (split * kNW * 32);          // 32-wide k blocks per wave
  int kb = 0;
  for (int c : {4, 3, 2, 1})
    if (kw % c == 0) { kb = c; break; }
  if (!kb) return false;
  if (kw / kb > 1) return false;                  // (shallow shapes are single-pass by construction)
  p->kb = kb;
  p->ko = kw / kb;
  p->split = split;
  p->nt = nt;
  p->mgroups = mgroups;
  p->mt = (mtiles + mgroups - 1) / mgroups;
  // one column tile per workgroup exists for <= 3 row tiles (where it is the rule's pick); an ODD number of column tiles
  // with more rows is reported as not covered (no model shape: the caller's next kernel takes it)
  if (nt == 1 && p->mt > 3) return false;
  if (p->mt > 16 || (mgroups - 1) * p->mt >= mtiles) return false;   // 16 row tiles are instantiated; no empty row group
  return true;
}

template <int MT, int KB, int NT, int EPI>
int launch(const void* x, const void* w, void* out, int64_t m, int n, int k, const Plan& p, hipStream_t s) {
  const size_t lds_red = (size_t)(kNW / 2) * MT * NT * 64 * sizeof(f32x4_t);
  const size_t lds_ring = (size_t)kNW * 2 * XTile<KB>::kBytes;
  const size_t lds = lds_red > lds_ring ? lds_red : lds_ring;
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& attr_set = attr_done[nvl_device_slot()];
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_decode_kernel<MT, KB, NT, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      nvl_set_error("nvl_linear_decode: cannot reserve %zu B of LDS", lds);
      return NVL_ELAUNCH;
    }
    attr_set = true;
  }
  const int out_cols = EPI == EPI_SILU ? n / 2 : n;
  const int col_wgs = EPI == EPI_SILU ? out_cols / 16 : out_cols / 16 / NT;
  hipLaunchKernelGGL((linear_decode_kernel<MT, KB, NT, EPI>), dim3(col_wgs, p.split, p.mgroups), dim3(kNW * 64), lds,
                     s, (const bf16_t*)x, (const bf16_t*)w, out, (int)m, n, k, p.ko, g_packed);
  return nvl_check_launch("nvl_linear_decode");
}

template <int KB, int NT, int EPI>
int dispatch_mt(const void* x, const void* w, void* out, int64_t m, int n, int k, const Plan& p, hipStream_t s) {
#define NVL_MT_CASE(V) \
  case V:              \
    return launch<V, KB, NT, EPI>(x, w, out, m, n, k, p, s);
  switch (p.mt) { NVL_MT_CASE(1) NVL_MT_CASE(2) NVL_MT_CASE(3) }
  if constexpr (NT == 2) {      // one column tile per workgroup is the plan of <= 3 row tiles only (make_plan)
    switch (p.mt) {
      NVL_MT_CASE(4) NVL_MT_CASE(5) NVL_MT_CASE(6) NVL_MT_CASE(7) NVL_MT_CASE(8) NVL_MT_CASE(9) NVL_MT_CASE(10)
      NVL_MT_CASE(11) NVL_MT_CASE(12) NVL_MT_CASE(13) NVL_MT_CASE(14) NVL_MT_CASE(15) NVL_MT_CASE(16)
    }
  }
#undef NVL_MT_CASE
  nvl_set_error("nvl_linear_decode: internal plan error (mt=%d nt=%d)", p.mt, NT);
  return NVL_EINVAL;
}

template <int NT, int EPI>
int dispatch_kb(const void* x, const void* w, void* out, int64_t m, int n, int k, const Plan& p, hipStream_t s) {
  switch (p.kb) {
    case 1: return dispatch_mt<1, NT, EPI>(x, w, out, m, n, k, p, s);
    case 2: return dispatch_mt<2, NT, EPI>(x, w, out, m, n, k, p, s);
    case 3: return dispatch_mt<3, NT, EPI>(x, w, out, m, n, k, p, s);
    case 4: return dispatch_mt<4, NT, EPI>(x, w, out, m, n, k, p, s);
  }
  nvl_set_error("nvl_linear_decode: internal plan error (kb=%d)", p.kb);
  return NVL_EINVAL;
}

// s = bf16(sum_s part[s][row]) + residual; residual <- bf16(s); y = bf16(s * rstd * w)
// One workgroup per row, ONE 8-element chunk per thread round (T = hidden/8 threads when that is <= 256):
// every load of a thread — S slab pieces, residual, weight — is independent and issued up front, so the
// kernel is one memory round trip + a block reduction long (it is latency-bound: <= a few MB in flight).
template <int T, int S>
__global__ __launch_bounds__(T) void add_rmsnorm_splitk_kernel(const float* __restrict__ part, int splits_rt,
                                                                int64_t split_stride, bf16_t* __restrict__ residual,
                                                                const bf16_t* __restrict__ weight,
                                                                bf16_t* __restrict__ y, int hidden, float eps) {
  constexpr int kMaxChunks = 4;
  __shared__ float red[(T + NVL_WAVE - 1) / NVL_WAVE];
  const int splits = S > 0 ? S : splits_rt;
  const int64_t row = blockIdx.x;
  const float* pr = part + row * hidden;
  bf16_t* rr = residual + row * hidden;
  bf16_t* yr = y + row * hidden;
  const int nchunks = hidden >> 3;
  float v[kMaxChunks][8];
  u32x4_t wraw[kMaxChunks], rraw[kMaxChunks];
  f32x4_t pa[kMaxChunks], pb[kMaxChunks];
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int chunk = threadIdx.x + c * T;
    if (chunk < nchunks) {
      wraw[c] = *reinterpret_cast<const u32x4_t*>(weight + chunk * 8);
      rraw[c] = *reinterpret_cast<const u32x4_t*>(rr + chunk * 8);
      pa[c] = *reinterpret_cast<const f32x4_t*>(pr + chunk * 8);
      pb[c] = *reinterpret_cast<const f32x4_t*>(pr + chunk * 8 + 4);
      if constexpr (S > 0) {
        f32x4_t ta[S > 1 ? S - 1 : 1], tb[S > 1 ? S - 1 : 1];
#pragma unroll
        for (int sidx = 1; sidx < S; ++sidx) {
          ta[sidx - 1] = *reinterpret_cast<const f32x4_t*>(pr + sidx * split_stride + chunk * 8);
          tb[sidx - 1] = *reinterpret_cast<const f32x4_t*>(pr + sidx * split_stride + chunk * 8 + 4);
        }
#pragma unroll
        for (int sidx = 1; sidx < S; ++sidx) {   // same summation order as the runtime-S loop: s = 0, 1, 2, ...
          pa[c] += ta[sidx - 1];
          pb[c] += tb[sidx - 1];
        }
      } else {
        for (int sidx = 1; sidx < splits; ++sidx) {
          pa[c] += *reinterpret_cast<const f32x4_t*>(pr + sidx * split_stride + chunk * 8);
          pb[c] += *reinterpret_cast<const f32x4_t*>(pr + sidx * split_stride + chunk * 8 + 4);
        }
      }
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int chunk = threadIdx.x + c * T;
    if (chunk < nchunks) {
      float r[8];
      unpack8(rraw[c], r);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[c][i] = round_bf16(pa[c][i]) + r[i];
        v[c][i + 4] = round_bf16(pb[c][i]) + r[i + 4];
      }
      *reinterpret_cast<u32x4_t*>(rr + chunk * 8) = pack8(v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += v[c][i] * v[c][i];
    }
  }
  ss = wave_allreduce_sum(ss);
  if constexpr (T > NVL_WAVE) {
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = ss;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < T / NVL_WAVE; ++ww) t += red[ww];
    ss = t;
  }
  const float rstd = rsqrtf(ss / (float)hidden + eps);
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int chunk = threadIdx.x + c * T;
    if (chunk < nchunks) {
      float wf[8], o[8];
      unpack8(wraw[c], wf);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = v[c][i] * rstd * wf[i];
      *reinterpret_cast<u32x4_t*>(yr + chunk * 8) = pack8(o);
    }
  }
}

template <int T>
void launch_add_rmsnorm_splitk(const float* partials, int splits, int64_t split_stride, bf16_t* residual,
                               const bf16_t* weight, bf16_t* y, int64_t rows, int hidden, float eps, hipStream_t s) {
#define NVL_SPLITK_CASE(SS)                                                                                       \
  hipLaunchKernelGGL((add_rmsnorm_splitk_kernel<T, SS>), dim3((unsigned)rows), dim3(T), 0, s, partials, splits, \
                     split_stride, residual, weight, y, hidden, eps)
  switch (splits) {
    case 1: NVL_SPLITK_CASE(1); break;
    case 2: NVL_SPLITK_CASE(2); break;
    case 4: NVL_SPLITK_CASE(4); break;
    case 8: NVL_SPLITK_CASE(8); break;
    default: NVL_SPLITK_CASE(0); break;
  }
#undef NVL_SPLITK_CASE
}

}  // namespace

extern "C" int nvl_linear_decode_splits(int64_t m, int n, int k, int mode) {
  Plan p;
  if (mode < 0 || mode > 2 || !make_plan(m, n, k, mode, &p)) return 0;
  return p.split;
}

extern "C" int nvl_linear_decode(const void* x, const void* weight, void* out, int64_t m, int n, int k, int mode,
                                 int weight_layout, void* stream) {
  NVL_REQUIRE(x && weight && out, "nvl_linear_decode: null pointer");
  NVL_REQUIRE(mode >= 0 && mode <= 2, "nvl_linear_decode: mode=%d (0 bf16, 1 silu*mul, 2 split-K fp32 partials)", mode);
  NVL_REQUIRE(weight_layout == 0 || weight_layout == 1, "nvl_linear_decode: weight_layout=%d (0 row-major [N, K], 1 tile-packed)", weight_layout);
  g_packed = weight_layout;
  NVL_REQUIRE(((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out) % 16 == 0,
              "nvl_linear_decode: pointers must be 16-byte aligned");
  Plan p;
  if (!make_plan(m, n, k, mode, &p)) {
    nvl_set_error("nvl_linear_decode: shape m=%lld n=%d k=%d mode=%d not covered (query nvl_linear_decode_splits first)",
                  (long long)m, n, k, mode);
    return NVL_EUNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  if (mode == EPI_SILU) return dispatch_kb<2, EPI_SILU>(x, weight, out, m, n, k, p, s);
  if (mode == EPI_BF16)
    return p.nt == 2 ? dispatch_kb<2, EPI_BF16>(x, weight, out, m, n, k, p, s)
                     : dispatch_kb<1, EPI_BF16>(x, weight, out, m, n, k, p, s);
  return p.nt == 2 ? dispatch_kb<2, EPI_PARTIAL>(x, weight, out, m, n, k, p, s)
                   : dispatch_kb<1, EPI_PARTIAL>(x, weight, out, m, n, k, p, s);
}

extern "C" int nvl_add_rmsnorm_splitk(const float* partials, int splits, void* residual, const void* weight, void* y,
                                      int64_t rows, int hidden, float eps, void* stream) {
  NVL_REQUIRE(partials && residual && weight && y, "nvl_add_rmsnorm_splitk: null pointer");
  NVL_REQUIRE(splits >= 1 && splits <= 64, "nvl_add_rmsnorm_splitk: splits=%d out of range", splits);
  NVL_REQUIRE(rows >= 0 && rows < (1ll << 31), "nvl_add_rmsnorm_splitk: bad rows=%lld", (long long)rows);
  NVL_REQUIRE(hidden > 0 && hidden % 8 == 0 && hidden <= 256 * 8 * 4,
              "nvl_add_rmsnorm_splitk: hidden=%d must be a multiple of 8 and <= %d", hidden, 256 * 8 * 4);
  NVL_REQUIRE(((uintptr_t)partials | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)residual) % 16 == 0,
              "nvl_add_rmsnorm_splitk: pointers must be 16-byte aligned");
  if (rows == 0) return NVL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t split_stride = rows * (int64_t)hidden;
  bf16_t* rp = (bf16_t*)residual;
  const bf16_t* wp = (const bf16_t*)weight;
  bf16_t* yp = (bf16_t*)y;
  if (hidden <= 64 * 8) {
    launch_add_rmsnorm_splitk<64>(partials, splits, split_stride, rp, wp, yp, rows, hidden, eps, s);
  } else if (hidden <= 128 * 8) {
    launch_add_rmsnorm_splitk<128>(partials, splits, split_stride, rp, wp, yp, rows, hidden, eps, s);
  } else {
    launch_add_rmsnorm_splitk<256>(partials, splits, split_stride, rp, wp, yp, rows, hidden, eps, s);
  }
  return nvl_check_launch("nvl_add_rmsnorm_splitk");
}
