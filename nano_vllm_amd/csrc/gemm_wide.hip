// Wide-tile streaming linear layers for gfx950: out[M, N] = x[M, K] . W[N, K]^T for decode steps (M <= a few hundred
// rows) on DEEP reductions — the Qwen3-8B / 32B projections (K = 4096 ... 25,600), full width and per-rank TP shapes.
// Replaces F.linear as called from LinearBase.forward (nano-vllm layers/linear.py:54-156) on those shapes, with the
// reference's activation glue (SiluAndMul, layers/activation.py:8-11) and the split-K slab hand-off to
// nvl_add_rmsnorm_splitk in the epilogues, like nvl_linear_decode does for K <= 1024.
//
// Why a second decomposition. gemm_decode.hip gives a workgroup 16-32 output columns and splits K over its 8
// waves: every workgroup ingests all M rows of x for those few columns, which is fine at K = 1024 (268 KB) and
// hopeless at K = 4096+ (x traffic L2 -> CU = M / 32 = 4.5x the weight bytes at 144 rows;
// profiles/r02_gemm_deep_*.json: 1.6-2.2 TB/s, below hipBLASLt). Here the ratio is M / (columns per workgroup):
//   * workgroup = NW = 3 or 4 CONSUMER waves + 1 LOADER wave, one workgroup per CU. Consumer wave w owns NT (1 or 2)
//     sixteen-column MFMA tiles x ALL row tiles of its row group (a workgroup covers 48 ... 128 output columns, no
//     cross-wave reduction). Few fat waves instead of 8 thin ones: every x fragment read from LDS feeds NT MFMAs
//     and only NW waves read it (8 waves with NT = 1 are LDS-bound at 144 rows: 288 ds_read_b128 per step).
//   * K advances in 128-wide steps. The loader wave stages the x tile of step s + 2 ([rows, 128] bf16, 36 KiB at 144
//     rows) with LDS-DMA (row-contiguous global_load_lds_dwordx4 from L2 into 16-byte XOR-swizzled slots =>
//     conflict-free ds_read_b128 B fragments) while the consumers work on step s; three LDS stages, one barrier
//     per step.
//   * Consumers stream their W fragments HBM -> VGPR with non-temporal loads through a RING-deep register ring
//     (the loads of step s + RING - 1 are issued during step s, a few per row-tile group between the MFMAs):
//     (RING - 1) x NT x 4 KiB in flight per wave. The x loads live in a DIFFERENT wave because a wave's loads retire in order: in the first version every
//     wave loaded x chunks too, and waiting for the x of step s + 1 (an L2 hit) also waited for every older weight
//     load, i.e. the ring was drained to its newest set at every step (~18 GB/s per CU whatever RING was).
//   * The K loop's body is RING steps of straight-line code with UNCONDITIONAL (index-clamped) prefetches and
//     unconditional uses, so both edges into the loop header carry the same outstanding-load pattern and hipcc's
//     s_waitcnt stays counted; a load whose only use sits under a branch is sunk into it and becomes synchronous
//     (both seen in the .s of earlier versions). The steps % RING tail runs after the loop on the ring's loaded sets.
//   * What the time is made of (measured): a CU's memory path is the saturated resource; it
//     moves ~18.6 GB/s of fragment-shaped weight loads and ~67 GB/s of row-contiguous x tiles, and the two ADD:
//     T = W bytes per CU / 18.6 GB/s + x bytes per CU / 67 GB/s (8B gate_up at 144 rows: 42 + 18 us). Hence ~256 equal
//     workgroups first (every CU streams), then as many columns per workgroup as that allows (fewer x re-reads).
//   * Workgroups start their K walk at different steps (kstep): rows of W are K * 2 bytes apart, lock-step walkers
//     would all hit the same offset of an 8-16 KiB stride at once.
//   * Small-N shapes (qkv / o / down, and everything per-rank under TP) do not have ~256 column tiles: K is split
//     over workgroups as well (grid.y) and the partial tiles leave as fp32 slabs [split][M][N]. For o_proj / down_proj
//     the consumer (nvl_add_rmsnorm_splitk) sums the slabs in its prologue ("reduce at the launch boundary"); for
//     bf16 / SiLU outputs a small reduce kernel follows (slab_reduce_kernel).
//   * v_mfma_f32_16x16x32_bf16, A = W fragment, B = x fragment: lane (l15 = lane & 15, lq = lane >> 4) ends up with
//     out[row = 16 mt + l15][col = tile + 4 lq .. + 3] => 8-byte bf16 / 16-byte fp32 stores.
//   * SiLU: a wave's two tiles are a gate tile and the matching up tile, paired in registers.
//   * PACKED weights (round 3, the default the engine uses): a fragment-shaped load of a ROW-MAJOR matrix makes every
//     16-lane group of the wave touch 16 different 128-byte lines (16 rows x 16 B), and the CU's address / L1 path
//     retires such a group at ~1 line per clock: 15 B/clk/CU measured (tools/probes/l2_read_probe.hip) against 45-60
//     for row-contiguous loads — which is why the weight and the x terms of the time model above ADD: together they
//     saturate that path. nvl_pack_weight_tiles stores W once, at model-load time, in the consumer waves' fragment
//     order — [N/16 tiles][K/32 k-blocks][64 lanes][8 elements]: the 16 x 32 sub-matrix of a (tile, k-block) is ONE
//     contiguous KiB in lane order, a wave's whole K walk over a tile one contiguous run — so a wave instruction reads
//     1 KiB contiguous (each 16-lane group 2 lines) and the weight stream costs a quarter of the address-path time.
// Rounding points are the reference's: the GEMM output is rounded to bf16 before the activation.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kBK = 128;                 // k per step (the 13-16 row-tile form steps by 64: see wide_bk)
constexpr int kKB = kBK / 32;            // 32-wide MFMA k blocks per step
enum { EPI_BF16 = 0, EPI_SILU = 1, EPI_PARTIAL = 2 };

__device__ __forceinline__ float silu_f32(float g) { return g / (1.f + __expf(-g)); }

// k columns per step. 128 (a 256-byte x row per step: one LDS-DMA instruction = 4 rows) up to 12 row tiles. At 13-16 row
// tiles (ONE row group for 193-256 rows) a 128-wide x tile is 64 KiB: only two of them fit the LDS, the loader can run
// just ONE step ahead, and every step then waits for a whole L2 round trip of its x tile (round 3: 8B gate_up at 256
// rows 75.9 us, behind the library's 68.4). Stepping by 64 columns halves the tile (32 KiB: three stages again, the
// loader two steps = the same 128 columns ahead), at one more barrier per 128 columns. Measured at 208 / 256 rows on
// the SAME decompositions (profiles/r04_gemm_wide_m256_bk64_vs_bk128.json): with one consumer wave per SIMD (NW = 3)
// 8B gate_up 74.0 -> 61.9 us (library + SiLU 68.1), 32B qkv 54.4 -> 47.7, 32B o 40.8 -> 37.3, 32B gate_up 239 -> 209; it
// LOSES with four thin consumer waves + two loaders (NW = 4: 8B down 49.9 -> 63.2) and on very deep per-workgroup K
// ranges (32B down, 6400 columns per workgroup: 131 -> 141) — hence the rule in wide_plan. NVL_WIDE_BK=128 keeps the
// round-3 form everywhere (A/B measurements).
__host__ __device__ constexpr bool wide_bk64_pays(int mt, int nw, int k_per_wg) { return mt > 12 && nw == 3 && k_per_wg <= 5120; }
__host__ __device__ constexpr int wide_stages(int mt, int bk = 128) { return 3 * mt * 16 * bk * 2 <= 144 * 1024 ? 3 : 2; }

// Loader waves per workgroup. One wave keeps at most 63 loads (63 KiB of 1-KiB LDS-DMA instructions) in flight — the
// vmcnt counter is 6 bits — which at the L2 latency seen under load is ~30 GB/s: in the x-heavy decompositions (few
// columns per workgroup, many rows) staging the x tile, not streaming the weights, set the step time
// (profiles/r03_gemm_wide_sweep_m144.jsonl: 36 KiB of x per 1.2-1.3 us whatever the weight bytes). The 5-wave
// configurations (NW = 4: <= 256 registers per wave, so a sixth wave fits a SIMD next to a consumer) take TWO loader
// waves, each staging every other 1-KiB piece of the tile; the 4-wave ones (NW = 3) own their SIMD's whole register
// file and cannot host another wave.
__host__ __device__ constexpr int wide_loaders(int nw) { return nw == 4 ? 2 : 1; }

// The hand-scheduled consumer K loop (inline asm, generated: tools/gen_wide_asm.py describes the schedule) for the
// one-wave-per-SIMD shape NT = 2, NW = 3, 64-column k steps, tile-packed weights, 12 or 16 row tiles.
#include "gemm_wide_core.inc"
#include "gemm_tile4_core.inc"
#ifdef NVL_PROBES
#include "gemm_wide_core_probes.inc"       // (generated by the probe build: the loop without its reads / without its MFMAs)
#endif
__host__ __device__ constexpr bool wide_core_shape(int mt, int nt, int nw, int bk, bool packed) {
  return (mt == 16 || mt == 12) && nt == 2 && nw == 3 && bk == 64 && packed;
}
// Loader waves of a workgroup. The core's consumers stay within 256 registers, so a FIFTH wave fits a SIMD next to one of
// them: two loaders, each staging every other 1-KiB piece of the x tile — one wave's 63 loads in flight (the 6-bit vmcnt)
// are ~30 GB/s at the L2 latency seen under load, and a 256-row x tile is 32 KiB per 64-column step: with ONE loader the
// x stream, not the matrix pipe and not the weight stream, set the step time (round 6: hipcc's consumer loop and the
// hand-scheduled one measured the same 66-68 us on the 8B gate_up at 256 rows).
__host__ __device__ constexpr int wide_loaders_of(int nw, bool core) { return core ? 2 : wide_loaders(nw); }

template <int MT, int NT, int NW, int EPI, int RING, bool PACKED, int BK = 128, bool CORE = false>
__global__ __launch_bounds__((NW + wide_loaders_of(NW, CORE)) * 64) void linear_wide_kernel(const bf16_t* __restrict__ x,
                                                                     const bf16_t* __restrict__ w,
                                                                     void* __restrict__ out, int M, int N, int K,
                                                                     int steps, int paired_tiles, int dbg) {
  static_assert(BK == 128 || BK == 64, "k columns per step");
  constexpr int kKB = BK / 32;                                    // (shadows the file-scope constants: per-step geometry)
  constexpr int kBK = BK;
  constexpr int kRowB = BK * 2;                                   // bytes of one x row per step
  constexpr int kRows = MT * 16;
  constexpr int kStage = kRows * kRowB;                           // bytes per LDS stage
  // x stages: three (the loader runs two steps ahead) while they fit the 160 KiB of LDS — up to 12 row tiles at 128
  // columns per step, 16 at 64; two (one step ahead) otherwise
  // CORE (hand-scheduled consumer loop): four stages, the loader three steps ahead, and a stage is published one step
  // EARLY — at the barrier that ends step s the tiles of steps s + 1 AND s + 2 have landed — so that the consumers can
  // request the first fragments of step s + 1 before that barrier (their matrix pipe does not drain at step boundaries)
  static_assert(!CORE || wide_core_shape(MT, NT, NW, BK, PACKED), "the generated core exists for this shape only");
  static_assert(!CORE || RING == NVL_WIDE_CORE_RING, "ring depth of the generated core");
  constexpr int NS = CORE ? NVL_WIDE_CORE_STAGES : wide_stages(MT, BK);
  constexpr int AHEAD = NS - 1;
  constexpr int LAND = CORE ? 2 : 1;                              // steps ahead that have LANDED at a step's barrier
  static_assert(EPI != EPI_SILU || NT == 2, "SiLU: a wave holds a gate tile and its up tile");
  constexpr int GT = EPI == EPI_SILU ? 1 : NT;                    // output tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  // Which (column tile, row group)? Plain: (blockIdx.x, blockIdx.z). With TWO row groups (145-288 rows) both groups of
  // a column tile stream the SAME weights: `paired_tiles` (= the number of column tiles) folds the row group into
  // blockIdx.x so that the two partners are dispatched back to back ON THE SAME XCD (workgroup id mod 8 picks the XCD;
  // ids 16 g + c and 16 g + 8 + c are the two row groups of tile 8 g + c): whichever of the two reaches a weight line
  // first brings it into that XCD's L2 and the other one hits it — the matrix leaves HBM once instead of twice.
  int tile_x = blockIdx.x, rgroup = blockIdx.z;
  if (paired_tiles > 0) {
    tile_x = (int)(blockIdx.x >> 4) * 8 + (int)(blockIdx.x & 7);
    rgroup = (int)(blockIdx.x >> 3) & 1;
    if (tile_x >= paired_tiles) return;                           // padding of the grid to a multiple of 16 (whole workgroup)
  }
  const int m_base = rgroup * kRows;
  const int out_cols = EPI == EPI_SILU ? N / 2 : N;
  const int64_t k0 = (int64_t)blockIdx.y * steps * kBK;
  const int last = steps - 1;
  // Workgroups walk their K range from different starting steps (wrapping around): rows of W are K * 2 bytes apart, so
  // workgroups marching in lock-step would all touch the same offset inside an 8-16 KiB stride at the same time
  // (the same few HBM channels).
  const int rot = (int)(((unsigned)tile_x + 3u * blockIdx.y) % (unsigned)steps);
  auto kstep = [&](int s) {                                       // logical step (prefetches past the end clamp) -> k step
    s = (s < last ? s : last) + rot;
    return s >= steps ? s - steps : s;
  };
  // (measurement switches, NVL_WIDE_DBG: bit 0 = the loader stages step 0 only, bit 1 = every weight load re-reads step 0's
  //  lines — each stream alone inside the real pipeline; results are garbage)
  //  — compiled in only by a probe build (NVL_PROBES=1 python -m nano_vllm_amd.build): the shipped library has no such switch
#ifdef NVL_PROBES
  const bool dbg_no_x = dbg & 1, dbg_no_w = dbg & 2;
#else
  constexpr bool dbg_no_x = false, dbg_no_w = false;
  (void)dbg;
#endif

  if (wave >= NW) {
    // ---- loader wave(s): x tile of step s + 2 -> LDS stage (s + 2) % 3 while the consumers work on step s --------
    // LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land at M0 + 16 lane, no staging registers), two steps in
    // flight: staging a step is one L2 round trip long, and with a single step in flight that round trip was the
    // step time of the whole workgroup (first loader version: 1.7 us per step at 16 rows). Instruction i covers
    // LDS chunks 64 i .. 64 i + 63 = rows 4 i + lq, slots l15; the XOR swizzle is applied on the SOURCE side (slot
    // l15 of row r holds chunk l15 ^ (r & 15)), so an instruction still reads 4 rows x 256 contiguous bytes.
    // The loads are inline asm (hipcc neither counts them nor keeps M0), so this wave's waits are explicit.
    constexpr int NL = wide_loaders_of(NW, CORE);
    constexpr int kPieceRows = 1024 / kRowB;                      // rows one 1-KiB LDS-DMA instruction covers: 4 (8 at BK = 64)
    constexpr int kPieces = kRows / kPieceRows;                   // 1-KiB pieces of a tile
    constexpr int kLC = kPieces / NL;                             // ... staged by THIS loader wave
    static_assert(kPieces % NL == 0 && kLC * (AHEAD - LAND) < 64, "vmcnt is a 6-bit counter");
    const int lw = wave - NW;                                     // which loader: pieces lw, lw + NL, ...
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const bf16_t* xb = x + k0;
    int x_src[kLC];                                               // element offsets (M * K < 2^31)
#pragma unroll
    for (int i = 0; i < kLC; ++i) {
      // lane -> (row inside the piece, 16-byte slot of that row) in the LANE-LINEAR image LDS-DMA writes; the chunk the
      // slot must hold is the consumers' swizzle run backwards (BK = 128: slot ^ (row & 15); BK = 64: slot ^ ((row >> 1) & 7),
      // two rows share a 256-byte bank row, see frag_off)
      const int prow = BK == 128 ? lq : lane >> 3, slot = BK == 128 ? l15 : lane & 7;
      const int row = kPieceRows * (i * NL + lw) + prow;
      int grow = m_base + row;
      grow = grow < M ? grow : M - 1;                             // padding rows read a valid row (never stored)
      const int chunk = BK == 128 ? (slot ^ (row & 15)) : (slot ^ ((row >> 1) & 7));
      x_src[i] = grow * K + (chunk << 3);
    }
    auto issue = [&](int s) {
      if (dbg_no_x && s > 0) return;
      const unsigned dst = lds0 + (unsigned)(s % NS) * kStage + (unsigned)lw * 1024;
      const bf16_t* src = xb + kstep(s) * kBK;
#pragma unroll
      for (int i = 0; i < kLC; ++i) {
        unsigned keep;
        const unsigned d = __builtin_amdgcn_readfirstlane(dst + i * (NL * 1024));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src + x_src[i]), "s"(d)
                     : "memory");
      }
    };
    // AHEAD steps in flight; "steps up to s + LAND have landed" = at most the newest AHEAD - LAND steps' loads are still
    // outstanding
    if (steps >= AHEAD) {
#pragma unroll
      for (int a = 0; a < AHEAD; ++a) issue(a);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLC * (AHEAD - LAND)) : "memory");  // steps 0 .. LAND - 1 have landed
    } else {
      for (int a = 0; a < steps; ++a) issue(a);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < steps; ++s) {
      // stage (s + AHEAD) % NS was last read during step s - 1: free since the previous barrier
      if (s + AHEAD <= last) {
        issue(s + AHEAD);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLC * (AHEAD - LAND)) : "memory"); // steps up to s + LAND have landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ---- consumer waves ----------------------------------------------------------------------------------------------
  // first output column of this wave's tiles, and the W rows feeding them (SiLU: tile 0 = gate, tile 1 = up)
  const int n0 = (tile_x * NW + wave) * (GT * 16);
  // element strides of this wave's weight pointer per k step / per 32-wide k block: row-major rows advance by k;
  // packed tiles advance by whole KiB blocks (512 elements) of the tile's contiguous run
  constexpr int kWStep = PACKED ? kKB * 512 : kBK;
  constexpr int kWBlk = PACKED ? 512 : 32;
  const bf16_t* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if constexpr (PACKED) {
      int tile = (n0 >> 4) + (nt % GT);
      const int ntiles = out_cols >> 4;
      tile = tile < ntiles ? tile : ntiles - 1;                   // ragged last workgroup: any valid tile, masked below
      if (EPI == EPI_SILU && nt >= GT) tile += ntiles;
      wrow[nt] = w + (int64_t)tile * 16 * K + (k0 >> 5) * 512 + lane * 8;
    } else {
      int row = n0 + (nt % GT) * 16 + l15;
      row = row < out_cols ? row : out_cols - 1;                  // ragged last workgroup: any valid row, masked below
      if (EPI == EPI_SILU && nt >= GT) row += out_cols;
      wrow[nt] = w + (int64_t)row * K + k0 + lq * 8;
    }
  }
  int frag_off[kKB];                                              // B fragment of k block kb: row l15 of a row tile
#pragma unroll
  for (int kb = 0; kb < kKB; ++kb)
    frag_off[kb] = BK == 128 ? l15 * 256 + (((kb * 4 + lq) ^ l15) << 4)
                             // 128-byte rows: rows 2j and 2j + 1 share a 256-byte bank row, so the 16 lanes of a fragment
                             // read (fixed chunk, rows 0-15) are conflict-free with slot = chunk ^ j
                             : l15 * 128 + (((kb * 4 + lq) ^ ((l15 >> 1) & 7)) << 4);

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if constexpr (CORE) {
    // ---- the whole K loop of this wave: one inline-asm statement (gemm_wide_core.inc) ------------------------------
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint64_t wb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const uint64_t p = (uint64_t)(uintptr_t)(wrow[nt] - lane * 8);           // the tile's run at this workgroup's k range
      // (readfirstlane returns int: without the unsigned casts the low half would be SIGN-extended over the high one)
      wb[nt] = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) << 32) |
               (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)p);
    }
    auto& flat = reinterpret_cast<f32x4_t(&)[MT * NT]>(acc);
    const int steps_s = __builtin_amdgcn_readfirstlane(steps), rot_s = __builtin_amdgcn_readfirstlane(rot);
    const int xa0 = (int)(lds0 + frag_off[0]), xa1 = (int)(lds0 + frag_off[1]);
    const int wstep = dbg_no_w ? 0 : kKB * 1024;                   // bytes of a tile's packed run per k step
#ifdef NVL_PROBES
    if (MT == 16 && (dbg & 4)) wide_core_mt16_noread(flat, xa0, xa1, lane * 16, wb[0], wb[1], steps_s, rot_s, wstep);
    else if (MT == 16 && (dbg & 8)) wide_core_mt16_nomfma(flat, xa0, xa1, lane * 16, wb[0], wb[1], steps_s, rot_s, wstep);
    else
#endif
    if constexpr (MT == 16) wide_core_mt16(flat, xa0, xa1, lane * 16, wb[0], wb[1], steps_s, rot_s, wstep);
    else wide_core_mt12(flat, xa0, xa1, lane * 16, wb[0], wb[1], steps_s, rot_s, wstep);
  } else {
  u32x4_t wf[RING][NT][kKB];
  auto wload = [&](u32x4_t (*dst)[kKB], int s) {
    s = dbg_no_w ? 0 : kstep(s);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int kb = 0; kb < kKB; ++kb)
        dst[nt][kb] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wrow[nt] + s * kWStep + kb * kWBlk));
  };

  // one k step: the MFMAs of every row tile against the weight set `wfs`, x fragments from LDS stage `xs`; when
  // `wdst` is given, the weight loads of logical step `sl` are issued INTERLEAVED with the MFMAs (a few per row-tile
  // group) instead of in one burst at the top of the step: a wave executes in order, so a burst that finds the
  // memory queue full stalls the MFMAs behind it, and a long MFMA phase leaves the queue unfed.
  // Row tiles go through the matrix pipe in groups of PT (2 independent accumulation chains per group either way);
  // the fragments of group g + 1 are read from LDS under the MFMAs of group g (pinned: hipcc otherwise re-serialises
  // read -> wait -> 2 MFMAs through one register quad).
  auto compute = [&](auto load_tag, const unsigned char* xs, const u32x4_t (*wfs)[kKB], u32x4_t (*wdst)[kKB], int sl) {
    constexpr bool LOAD = decltype(load_tag)::value;
    constexpr int PT = NT == 1 ? 2 : 1;
    constexpr int NG = (MT + PT - 1) / PT;
    constexpr int L = NT * kKB;                                   // weight loads per step
    const int ks = LOAD && !dbg_no_w ? kstep(sl) : 0;
    u32x4_t f[2][PT][kKB];
    auto fread = [&](int g, u32x4_t (*dst)[kKB]) {
#pragma unroll
      for (int t = 0; t < PT; ++t)
        if (g * PT + t < MT) {
#pragma unroll
          for (int kb = 0; kb < kKB; ++kb)
            dst[t][kb] = *reinterpret_cast<const u32x4_t*>(xs + (g * PT + t) * 16 * kRowB + frag_off[kb]);
        }
    };
    fread(0, f[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int j0 = g * L / NG, j1 = (g + 1) * L / NG;           // this group's share of the step's weight loads
      if constexpr (LOAD) {
#pragma unroll
        for (int j = 0; j < L; ++j)
          if (j >= j0 && j < j1)
            wdst[j / kKB][j % kKB] = __builtin_nontemporal_load(
                reinterpret_cast<const u32x4_t*>(wrow[j / kKB] + ks * kWStep + (j % kKB) * kWBlk));
      }
      if (g + 1 < NG) fread(g + 1, f[(g + 1) & 1]);
#pragma unroll
      for (int kb = 0; kb < kKB; ++kb)
#pragma unroll
        for (int t = 0; t < PT; ++t)
          if (g * PT + t < MT) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[g * PT + t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8_t, wfs[nt][kb]), __builtin_bit_cast(bf16x8_t, f[g & 1][t][kb]),
                  acc[g * PT + t][nt], 0, 0, 0);
          }
      if constexpr (LOAD) {                                       // the group's loads first
#pragma unroll
        for (int j = 0; j < L; ++j)
          if (j >= j0 && j < j1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      if (g + 1 < NG) {
#pragma unroll
        for (int j = 0; j < PT * kKB; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read ...
          __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);  // ... per NT MFMAs
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Pipeline invariant at the top of step s (ring slot i = s mod RING): LDS stage s % 3 holds x of step s (loader),
  // wf[i] .. wf[i + RING - 2] hold (or have in flight) the weights of steps s .. s + RING - 2.
#pragma unroll
  for (int r = 0; r < RING - 1; ++r) wload(wf[r], r);
  __syncthreads();                                                // stage 0 is staged
  const int nblk = steps / RING;
  for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
    for (int i = 0; i < RING; ++i) {
      const int s = blk * RING + i;
      // the weights RING - 1 steps ahead are requested inside compute()
      compute(std::true_type{}, smem + (s % NS) * kStage, wf[i], wf[(i + RING - 1) % RING], s + RING - 1);
      __syncthreads();                                            // stage (s + 1) % 3 is staged, stage s % 3 is free
    }
  }
#pragma unroll
  for (int i = 0; i < RING - 1; ++i) {
    const int s = nblk * RING + i;
    if (s < steps) {
      compute(std::false_type{}, smem + (s % NS) * kStage, wf[i], nullptr, 0);
      __syncthreads();
    }
  }
  }  // !CORE

  // ---- epilogue ------------------------------------------------------------------------------------------------
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_base + mt * 16 + l15;
    if (m >= M) continue;
#pragma unroll
    for (int gt = 0; gt < GT; ++gt) {
      const int n = n0 + gt * 16;
      if (n >= out_cols) continue;
      if constexpr (EPI == EPI_SILU) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu_f32(round_bf16(acc[mt][0][r])) * round_bf16(acc[mt][1][r]);
        u32x2_t ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        *reinterpret_cast<u32x2_t*>((bf16_t*)out + (int64_t)m * out_cols + n + lq * 4) = ov;
      } else if constexpr (EPI == EPI_BF16) {
        u32x2_t o = {pack_bf16x2(acc[mt][gt][0], acc[mt][gt][1]), pack_bf16x2(acc[mt][gt][2], acc[mt][gt][3])};
        *reinterpret_cast<u32x2_t*>((bf16_t*)out + (int64_t)m * N + n + lq * 4) = o;
      } else {
        *reinterpret_cast<f32x4_t*>((float*)out + ((int64_t)blockIdx.y * M + m) * N + n + lq * 4) = acc[mt][gt];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// FOUR-consumer tile kernel (round 6) for 145-256 rows on tile-packed weights: every SIMD of the CU runs a matrix wave.
// The one-wave-per-SIMD kernel above leaves a SIMD to its loader wave and streams W through the consumers' registers; its
// matrix work (3 SIMDs, 17 cycles per 16x16x32 MFMA at the ~1.9 GHz the chip holds under this load) is 37 us on the 8B
// gate_up at 256 rows before a byte is waited for (profiles/r06_gemm_core_streams_*.txt: 42 us measured for the MFMA-only
// loop, 48 with the fragment reads, 60 with both streams). Here:
//   waves 0-3  consumers: wave q owns RT row tiles (rows 64 q ...) x ALL CT column tiles of the workgroup; both operands
//              come from LDS (gemm_tile4_core.inc, generated by tools/gen_wide_asm.py — its GenTile docstring is the
//              schedule); <= 256 registers, so seven waves fit the CU;
//   waves 4-5  x loaders: the [RT x 64 rows][64 columns] tile of step s + 4 by LDS-DMA into a 4-stage ring, every other
//              1-KiB piece each (one wave's 63 loads in flight are ~30 GB/s);
//   waves 6-7  W loaders: CT x 2 KiB of packed fragments per step (wave 6 the first k block of every tile, wave 7 the
//              second), HBM -> their OWN register rings (6-8 steps deep: that is where the HBM latency is hidden — 160 KiB
//              of LDS cannot hold four x stages AND a deep W ring) -> ds_write_b128 into a 2-stage LDS ring one step ahead
//              of the consumers.
// Two barriers per 64-column step (an s_barrier costs the matrix pipe nothing when the MFMAs queue behind it:
// tools/probes/mfma_issue_probe.hip): a(s) after the first k block = "x(s + 1), W(s + 1) are in LDS", b(s) after the second
// = "every read of x(s) / W(s) has returned".
constexpr int kT4XStages = 4, kT4WStages = 2;
__host__ __device__ constexpr int tile4_wring(int ct) { return ct <= 6 ? 8 : 6; }   // W register ring (k steps) of a W loader: <= 192 registers

template <int RT, int CT, int EPI>
__global__ __launch_bounds__(8 * 64) void linear_tile4_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                              void* __restrict__ out, int M, int N, int K, int steps, int dbg) {
#ifdef NVL_PROBES
  const bool dbg_no_x = dbg & 1, dbg_no_w = dbg & 2;              // (probe builds: NVL_WIDE_DBG, as linear_wide_kernel)
#else
  constexpr bool dbg_no_x = false, dbg_no_w = false;
  (void)dbg;
#endif
  constexpr int kRows = RT * 64;                                  // rows of an x stage (four waves x RT row tiles)
  constexpr int kXStage = kRows * 128;                            // bytes: 64 columns per step
  constexpr int kWStage = CT * 2048;
  constexpr int HC = CT / 2;                                      // SiLU: gate tiles 0 .. HC - 1, their up tiles HC .. CT - 1
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wlds = smem + kT4XStages * kXStage;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int tile_x = blockIdx.x;
  const int out_cols = EPI == EPI_SILU ? N / 2 : N;
  const int ntiles = out_cols >> 4;
  const int64_t k0 = (int64_t)blockIdx.y * steps * 64;
  const int last = steps - 1;
  const int rot = (int)(((unsigned)tile_x + 3u * blockIdx.y) % (unsigned)steps);   // (as linear_wide_kernel: staggered K walks)
  auto kstep = [&](int s) {
    s = (s < last ? s : last) + rot;
    return s >= steps ? s - steps : s;
  };
  // W tile (16 weight rows) behind column tile c of this workgroup; past the ragged end: any valid tile, masked at the store
  auto wtile = [&](int c) {
    int t = EPI == EPI_SILU ? tile_x * HC + (c % HC) : tile_x * CT + c;
    t = t < ntiles ? t : ntiles - 1;
    return EPI == EPI_SILU && c >= HC ? t + ntiles : t;
  };

  if (wave >= 6) {
    // ---- W loaders: wave 6 + kb takes k block kb of every tile ---------------------------------------------------------
    constexpr int RW = tile4_wring(CT), P = CT;                   // ring depth (steps), 1-KiB pieces per step and loader
    const int kb = wave - 6;
    const bf16_t* src[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) src[c] = w + (int64_t)wtile(c) * 16 * K + ((k0 >> 5) + kb) * 512 + lane * 8;
    u32x4_t ring[RW][P];
    auto load = [&](u32x4_t* dst, int s) {
      const int ks = dbg_no_w ? 0 : kstep(s);
#pragma unroll
      for (int c = 0; c < CT; ++c)
        dst[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src[c] + ks * 1024));
    };
    auto park = [&](const u32x4_t* set, int s) {                  // registers -> LDS stage s % 2, lane-linear (fragment order)
#ifdef NVL_PROBES
      if (dbg & 4) return;                                        // (probe: no ds_writes)
#endif
      unsigned char* dst = wlds + (s & 1) * kWStage + kb * 1024 + lane * 16;
#pragma unroll
      for (int c = 0; c < CT; ++c) *reinterpret_cast<u32x4_t*>(dst + c * 2048) = set[c];
    };
#pragma unroll
    for (int r = 0; r < RW - 1; ++r) load(ring[r], r);
    park(ring[0], 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the ds_writes have landed (NOT vmcnt: the ring stays in flight)
    __builtin_amdgcn_s_barrier();
    // RW steps of straight-line code per iteration (unconditional, index-clamped loads: hipcc's vmcnt stays counted, as in
    // linear_wide_kernel's consumer loop); the steps % RW tail runs on the sets already requested
    const int nblk = steps / RW;
    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int s = blk * RW + i;
        park(ring[(i + 1) % RW], s + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // a(s): W(s + 1) is in LDS
        load(ring[(i + RW - 1) % RW], s + RW - 1);                // (requests go out in the half step nobody waits for this wave)
        __builtin_amdgcn_s_barrier();                             // b(s)
      }
    }
#pragma unroll
    for (int i = 0; i < RW - 1; ++i) {
      if (nblk * RW + i < steps) {
        park(ring[(i + 1) % RW], nblk * RW + i + 1);              // (past the last step: a clamped set, never read)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }
  if (wave >= 4) {
    // ---- x loaders (as linear_wide_kernel's, 64-column steps): pieces lw, lw + 2, ... of the tile ----------------------
    constexpr int kPieces = kRows / 8, kLC = kPieces / 2;
    static_assert(kPieces % 2 == 0 && 3 * kLC < 64, "vmcnt is a 6-bit counter");
    const int lw = wave - 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const bf16_t* xb = x + k0;
    int x_src[kLC];
#pragma unroll
    for (int i = 0; i < kLC; ++i) {
      const int prow = lane >> 3, slot = lane & 7;
      const int row = 8 * (i * 2 + lw) + prow;
      int grow = row < M ? row : M - 1;                           // padding rows read a valid row (never stored)
      x_src[i] = grow * K + ((slot ^ ((row >> 1) & 7)) << 3);
    }
    // pieces [i0, i1) of this loader's share of tile s
    auto issue = [&](int s, int i0, int i1) {                     // (wave-uniform bounds; the piece loop stays unrolled)
      if (dbg_no_x && s > 0) return;
      const unsigned dst = lds0 + (unsigned)(s % kT4XStages) * kXStage + (unsigned)lw * 1024;
      const bf16_t* srcp = xb + kstep(s) * 64;
#pragma unroll
      for (int i = 0; i < kLC; ++i) {
        if (i < i0 || i >= i1) continue;
#ifdef NVL_PROBES
        if ((dbg & 32) && (i & 1)) continue;                      // (probe: half the pieces — is the loader ISSUE-bound?)
#endif
        unsigned keep;
        const unsigned d = __builtin_amdgcn_readfirstlane(dst + i * 2048);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(srcp + x_src[i]), "s"(d)
                     : "memory");
      }
    };
    static_assert(kLC % 2 == 0, "a tile is issued in two halves");
    // Tile s + 3 goes out in TWO halves, one behind each barrier of step s (its stage, (s - 1) % 4, is free since b(s - 1)):
    // issuing a whole tile between b(s) and a(s + 1) — half a step — made the loaders the last to arrive at a(s + 1)
    // (an LDS-DMA instruction costs its wave 30-100 cycles of issue; 16 of them are most of half a step).
    const int first = steps < 3 ? steps : 3;
    for (int a = 0; a < first; ++a) issue(a, 0, kLC);
    if (first == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kLC) : "memory");           // x(0) has landed
    else if (first == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < steps; ++s) {
      const bool more = s + 3 <= last;
      if (more) issue(s + 3, 0, kLC / 2);
      // x(s + 1) has landed = only what was requested behind it (x(s + 2) and the half of x(s + 3)) may still be on its way
      const int rem = last - (s + 1);                             // tiles behind x(s + 1) that exist
      if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLC + kLC / 2) : "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLC) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                               // a(s)
      if (more) issue(s + 3, kLC / 2, kLC);
      __builtin_amdgcn_s_barrier();                               // b(s)
    }
    return;
  }

  // ---- consumers ---------------------------------------------------------------------------------------------------
  f32x4_t acc[RT * CT];
#pragma unroll
  for (int i = 0; i < RT * CT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned wl0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wlds;
    // B fragment of k block kb: row l15 of a row tile, 16-byte slot swizzled with the row pair (the loaders' image)
    const int f0 = l15 * 128 + (((0 * 4 + lq) ^ ((l15 >> 1) & 7)) << 4), f1 = l15 * 128 + (((1 * 4 + lq) ^ ((l15 >> 1) & 7)) << 4);
    const int xrow0 = wave * RT * 2048;                           // this wave's first row tile inside a stage
    const int steps_s = __builtin_amdgcn_readfirstlane(steps);
#define NVL_T4_CORE(R_, C_) if constexpr (RT == R_ && CT == C_) tile4_core_r##R_##c##C_(acc, (int)(lds0 + xrow0 + f0), (int)(lds0 + xrow0 + f1), (int)(wl0 + lane * 16), steps_s);
#ifdef NVL_PROBES
    if (RT == 4 && CT == 6 && (dbg & 24)) {
      if constexpr (RT == 4 && CT == 6) {
        if (dbg & 8) tile4_core_r4c6_noread(acc, (int)(lds0 + xrow0 + f0), (int)(lds0 + xrow0 + f1), (int)(wl0 + lane * 16), steps_s);
        else tile4_core_r4c6_nomfma(acc, (int)(lds0 + xrow0 + f0), (int)(lds0 + xrow0 + f1), (int)(wl0 + lane * 16), steps_s);
      }
    } else
#endif
    {
    NVL_T4_CORE(4, 6) NVL_T4_CORE(4, 8) NVL_T4_CORE(4, 4) NVL_T4_CORE(3, 6) NVL_T4_CORE(3, 8) NVL_T4_CORE(3, 4)
    }
#undef NVL_T4_CORE
  }
  // ---- epilogue: lane (l15, lq) holds out[row = 16 rt + l15][col = tile + 4 lq .. + 3] ---------------------------------
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int m = (wave * RT + r) * 16 + l15;
    if (m >= M) continue;
#pragma unroll
    for (int c = 0; c < (EPI == EPI_SILU ? HC : CT); ++c) {
      const int n = (EPI == EPI_SILU ? tile_x * HC + c : tile_x * CT + c) * 16;
      if (n >= out_cols) continue;
      if constexpr (EPI == EPI_SILU) {
        const f32x4_t g = acc[r * CT + c], u = acc[r * CT + c + HC];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = silu_f32(round_bf16(g[j])) * round_bf16(u[j]);
        u32x2_t ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        *reinterpret_cast<u32x2_t*>((bf16_t*)out + (int64_t)m * out_cols + n + lq * 4) = ov;
      } else if constexpr (EPI == EPI_BF16) {
        const f32x4_t v = acc[r * CT + c];
        u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>((bf16_t*)out + (int64_t)m * N + n + lq * 4) = o;
      } else {
        *reinterpret_cast<f32x4_t*>((float*)out + ((int64_t)blockIdx.y * M + m) * N + n + lq * 4) = acc[r * CT + c];
      }
    }
  }
}

// out = epilogue(sum_s part[s]) for bf16 / SiLU outputs whose GEMM was split over K: 4 output columns per thread,
// every slab piece requested before the first add (one memory round trip); slabs are summed in split order.
template <int EPI>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ part, int splits,
                                                           int64_t split_stride, bf16_t* __restrict__ out, int M,
                                                           int N) {
  const int out_cols = EPI == EPI_SILU ? N / 2 : N;
  const int qpr = out_cols >> 2;                                  // quads per row
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)M * qpr) return;
  const int row = (int)(q / qpr), c = (int)(q - (int64_t)row * qpr) * 4;
  const float* p = part + (int64_t)row * N + c;
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
  f32x4_t b = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == EPI_SILU) b = *reinterpret_cast<const f32x4_t*>(p + out_cols);
  for (int s = 1; s < splits; ++s) {
    a += *reinterpret_cast<const f32x4_t*>(p + s * split_stride);
    if constexpr (EPI == EPI_SILU) b += *reinterpret_cast<const f32x4_t*>(p + s * split_stride + out_cols);
  }
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = EPI == EPI_SILU ? silu_f32(round_bf16(a[r])) * round_bf16(b[r]) : a[r];
  u32x2_t ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
  *reinterpret_cast<u32x2_t*>(out + (int64_t)row * out_cols + c) = ov;
}

// W [N, K] row-major -> packed [N/16][K/32][64 lanes][8]: thread = one 16-byte chunk of the destination.
__global__ __launch_bounds__(256) void pack_weight_tiles_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ packed,
                                                                 int64_t chunks, int K) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= chunks) return;
  const int lane = (int)(c & 63);
  const int64_t blk = c >> 6;                                     // (tile, k-block)
  const int kblocks = K >> 5;
  const int64_t tile = blk / kblocks;
  const int kb = (int)(blk - tile * kblocks);
  const int64_t row = tile * 16 + (lane & 15);
  const int k = kb * 32 + (lane >> 4) * 8;
  *reinterpret_cast<u32x4_t*>(packed + c * 8) = *reinterpret_cast<const u32x4_t*>(w + row * K + k);
}

// ---- host: plan ---------------------------------------------------------------------------------------------------
struct WidePlan {
  int mt, nt, nw, mgroups, tiles, split, steps;   // tiles = workgroups along N; steps = bk-wide k steps per workgroup
  int bk;                                         // k columns per step: 128, or 64 (wide_bk)
};

// Weight ring depth. 5 waves (NW = 4) share 4 SIMDs, so those kernels live in 256 registers: one set less at 7+ row
// tiles x 2 column tiles.
constexpr int ring_of128(int nt, int nw, int mt) {
  if (nw == 3) return nt == 1 ? 8 : (mt > 9 ? 4 : 6);   // 4 waves, one per SIMD: 512 registers per wave
  if (mt > 9) return mt == 12 ? 3 : 4;                  // (only NT = 1 is instantiated above 9 row tiles with 5 waves; 12 row
                                                        //  tiles keep three x stages and spill with a deeper ring)
  return nt == 1 ? 6 : (mt >= 7 ? 3 : 4);
}
// in steps of `bk` columns: a 64-column step holds half the weight bytes, so the ring is twice as deep for the same
// bytes in flight (and the same registers)
constexpr int ring_of(int nt, int nw, int mt, int bk = 128) { return ring_of128(nt, nw, mt) * (128 / bk); }

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

// Row-tile counts that are instantiated (a batch is rounded up to the next one; the x rows past M are clamped reads)
int round_mt(int mtiles, int nt) {
  static const int kMT[] = {1, 2, 3, 5, 7, 9, 12, 16};
  for (int c : kMT)
    if (c >= mtiles) return c;
  return 0;
}

// Model of one launch (us) for a candidate decomposition, fitted to a sweep of every (NT, NW, split) plan on the
// Qwen3-8B / 32B / 32B-per-rank shapes at 144 rows with tile-packed weights (tools/gemm_wide_sweep.py,
// profiles/r03_gemm_wide_sweep_m144.jsonl: mean |log error| 9 %, the plan it picks is within 3 % of the best measured
// one on average). A CU streams ~28 GB/s of packed weights and ~67 GB/s of x tiles (L2 -> LDS); the two overlap, but
// not perfectly (a quarter of the shorter one shows); time ~ (rounds of 256 workgroups) x (steps per workgroup) x
// (time of a step on one CU) plus the pipeline fill per workgroup, the launch, and the slab traffic when K is split.
double wide_cost(int64_t m, int n, int k, int mode, const WidePlan& p) {
  const int wgs = p.tiles * p.split * p.mgroups;
  const double rounds = (double)((wgs + 255) / 256);
  const double wbytes_step = (double)p.nw * p.nt * 16 * 256.0;
  const double t_w = wbytes_step / 28.0e3;                                        // us at 28 GB/s per CU
  // (the model works in 128-column units; a 64-column plan makes two barriers per unit)
  const double t_mfma = (double)p.mt * p.nt * kKB * 16.0 / 2400.0;               // 16 clk per MFMA at 2.4 GHz
  double t_x = (double)p.mt * 16 * 256.0 / 67.0e3;                                // L2 -> LDS staging of the x tile
  if (wide_stages(p.mt, p.bk) == 2 && t_x < 1.1) t_x = 1.1;                      // one step ahead only: an L2 round trip per step
  double t_step = t_w;
  if (t_mfma > t_step) t_step = t_mfma;
  if (t_x > t_step) t_step = t_x;
  t_step += 0.25 * (t_w < t_x ? t_w : t_x) + 0.05 * (128 / p.bk);                // imperfect overlap + the barrier(s)
  // the whole chip cannot exceed ~5.6 TB/s either
  const double active = wgs < 256 ? wgs : 256;
  const double chip = active * wbytes_step / 5.6e6;
  if (chip > t_step) t_step = chip;
  double t = rounds * (p.steps * (p.bk / 128.0) * t_step + 2.0) + 2.5;
  if (p.split > 1 || mode == EPI_PARTIAL) {
    const double slab = (double)p.split * m * n * 4.0;
    t += slab / 4.0e6;                                                            // written here ...
    if (mode != EPI_PARTIAL) t += slab / 4.0e6 + 3.5;                             // ... re-read by the reduce kernel
    else t += (p.split - 1) * (double)m * n * 4.0 / 6.0e6;                        // ... or by the consumer's prologue
  }
  return t;
}

// Measured-best decompositions where the model above mis-ranks them (round-4 sweeps of every (NT, NW, split) plan at 32 ...
// 256 rows, profiles/r04_gemm_wide_sweep_*.jsonl). The model was fitted at 144 rows on the round-3 kernel; it
// under-estimates the 13-16-row-tile form by 7-35 % (a blanket penalty on that form sends other shapes to worse plans) and,
// for the deep-K slab projections, prefers four thin waves x one column tile where two column tiles per wave (half the x
// fragment reads per MFMA) with a deeper K split are 10-17 % faster. us, planner -> tuned:
//   Qwen3-8B  down [4096, 12288]  64-144 rows: 24.0 / 32.7 / 33.2 -> 21.9 / 27.3 / 28.1 (at 64 / 131 / 144)
//                                 208 / 256 rows: 49.4 / 50.2 -> 37.4 / 45.2 (two paired row groups)
//   Qwen3-32B down [5120, 25600]  96 rows 72.7 -> 66.0;  112-144 rows: 80.5 / 87.8 / 93.0 -> 69.3 / 74.9 / 78.2;  256: 130 -> 112.5
//   Qwen3-32B qkv  [10240, 5120]  32-144 rows: 25.0 / 32.9 / 39.2 / 39.8 -> 23.2 / 29.5 / 36.3 / 38.0;  256: 47.7 -> 45.6
//   Qwen3-8B  o    [4096, 4096]   208 rows: 15.8 -> 14.6
// NVL_WIDE_TUNED=0 = the model's picks everywhere (A/B).
struct TunedPlan { int n, k, mode, mtiles_lo, mtiles_hi, nt, nw, split; };
constexpr TunedPlan kTuned[] = {
    {4096, 12288, EPI_PARTIAL, 4, 9, 2, 4, 8},
    {4096, 12288, EPI_PARTIAL, 13, 16, 2, 4, 4},
    {5120, 25600, EPI_PARTIAL, 6, 6, 2, 4, 4},
    {5120, 25600, EPI_PARTIAL, 7, 9, 2, 3, 8},
    {5120, 25600, EPI_PARTIAL, 13, 16, 2, 4, 8},
    {10240, 5120, EPI_BF16, 2, 9, 2, 3, 2},
    {10240, 5120, EPI_BF16, 13, 16, 2, 3, 1},
    {4096, 4096, EPI_PARTIAL, 13, 13, 2, 4, 4},
    // per-rank qkv of Qwen3-32B at TP = 8 / TP = 4 as fp32 slabs for the fused decode attention (round 5: the attention
    // prologue sums them, so FEW slabs matter more than the last microsecond of the GEMM — every wave of the attention
    // grid reads (G + 2) x 512 B per slab): sweep profiles/r05_gemm_wide_sweep_tp_rank_shapes.jsonl, us incl. the reduce
    // launch these plans no longer need: TP8 split 4 (nt 1, nw 3) 14.6 vs the model's split 8 14.3 at 144 rows; 13-16
    // row tiles split 4 (nt 1, nw 4) 15.8; TP4 split 4 (nt 1, nw 4) 17.4 vs split 5 17.9, 13-16 row tiles nt 2 nw 3 split 4 21.3
    {1280, 5120, EPI_PARTIAL, 1, 9, 1, 3, 4},
    {1280, 5120, EPI_PARTIAL, 10, 16, 1, 4, 4},
    {2560, 5120, EPI_PARTIAL, 1, 9, 1, 4, 4},
    {2560, 5120, EPI_PARTIAL, 10, 16, 2, 3, 4},
    // per-rank gate_up / down at TP = 8 / TP = 4 (same sweep; us, model's pick -> measured best): TP8 gate_up 64-144 rows
    // split 5 -> 4 (22.3 -> 21.5 / 28.0 -> 27.4); TP4 down 64 rows split 2 -> nt 2 nw 4 split 5 (19.1 -> 16.4)
    {6400, 5120, EPI_SILU, 2, 9, 2, 4, 4},
    {5120, 6400, EPI_PARTIAL, 2, 9, 2, 4, 5},
};

bool wide_plan(int64_t m, int n, int k, int mode, WidePlan* best) {
  if (m < 1 || m > 1024 || n < 16 || k < kBK || k % kBK) return false;
  if (m * (int64_t)k >= (1ll << 31) || (int64_t)n * k >= (1ll << 40)) return false;   // 32-bit x element offsets in the loader
  if (mode == EPI_SILU ? n % 32 : n % 16) return false;
  const int out_cols = mode == EPI_SILU ? n / 2 : n;
  const int mtiles = (int)((m + 15) / 16);
  const int force_bk = env_int("NVL_WIDE_BK", 0) == 128 ? 128 : 0;
  int force_nt = env_int("NVL_WIDE_NT", 0), force_nw = env_int("NVL_WIDE_NW", 0);
  int force_split = env_int("NVL_WIDE_SPLIT", 0);
  if (!force_nt && !force_nw && !force_split && env_int("NVL_WIDE_TUNED", 1))
    for (const TunedPlan& t : kTuned)
      if (t.n == n && t.k == k && t.mode == mode && mtiles >= t.mtiles_lo && mtiles <= t.mtiles_hi) {
        force_nt = t.nt; force_nw = t.nw; force_split = t.split;
      }
  double best_t = 1e30;
  for (int nt : {2, 1}) {
    if (force_nt && nt != force_nt) continue;
    if (mode == EPI_SILU && nt == 1) continue;
    for (int nw : {4, 3}) {
      if (force_nw && nw != force_nw) continue;
      // Row tiles per row group: <= 9 everywhere; 10-16 (ONE row group up to 256 rows: every weight byte is streamed
      // once, where two groups stream it twice) only in the decompositions whose register file holds 16 row tiles of
      // accumulators next to the weight ring — 4 waves with one SIMD each (NW = 3), or one column tile per wave.
      // More rows = more row groups, whose workgroups are paired on one XCD (see the kernel).
      const bool big_ok = nw == 3 || nt == 1;
      for (int mt_max : {16, 9}) {
        if (mt_max > 9 && (!big_ok || mtiles <= 9)) continue;
        WidePlan p;
        p.nt = nt;
        p.nw = nw;
        p.mgroups = (mtiles + mt_max - 1) / mt_max;
        p.mt = round_mt((mtiles + p.mgroups - 1) / p.mgroups, nt);
        if (!p.mt || (p.mt > 9 && !big_ok)) continue;
        // the decomposition is chosen by the round-3 model in 128-column steps (every plan it picks for the model shapes
        // has been measured); the k step is refined afterwards, below
        p.bk = 128;
        const int ksteps = k / p.bk;
        const int cols = nw * (mode == EPI_SILU ? 1 : nt) * 16;
        p.tiles = (out_cols + cols - 1) / cols;
        for (int split = 1; split <= 32; ++split) {
          if (ksteps % split) continue;
          if (force_split && split != force_split) continue;
          if (split > 1 && ksteps / split < 2 * (128 / p.bk)) break;
          p.split = split;
          p.steps = ksteps / split;
          const double t = wide_cost(m, n, k, mode, p);
          if (t < best_t) { best_t = t; *best = p; }
        }
      }
    }
  }
  if (best_t < 1e30 && !force_bk && wide_bk64_pays(best->mt, best->nw, best->steps * 128)) {
    best->bk = 64;
    best->steps *= 2;
  }
#ifdef NVL_PROBES
  if (best_t < 1e30 && env_int("NVL_WIDE_DEBUG", 0))
    fprintf(stderr, "nvl_linear_wide plan m=%lld n=%d k=%d mode=%d: nt=%d nw=%d mt=%d groups=%d tiles=%d split=%d steps=%d x %d -> %d wgs, model %.1f us\n",
            (long long)m, n, k, mode, best->nt, best->nw, best->mt, best->mgroups, best->tiles, best->split, best->steps, best->bk,
            best->tiles * best->split * best->mgroups, best_t);
#endif
  return best_t < 1e30;
}

template <int MT, int NT, int NW, int EPI, bool PACKED, int BK, bool CORE = false>
int launch_wide_l(const WidePlan& p, const void* x, const void* w, void* out, int64_t m, int n, int k, hipStream_t s) {
  constexpr int RING = CORE ? NVL_WIDE_CORE_RING : ring_of(NT, NW, MT, BK);
  const size_t lds = (size_t)(CORE ? NVL_WIDE_CORE_STAGES : wide_stages(MT, BK)) * MT * 16 * BK * 2;
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& attr_set = attr_done[nvl_device_slot()];
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_wide_kernel<MT, NT, NW, EPI, RING, PACKED, BK, CORE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      nvl_set_error("nvl_linear_wide: cannot reserve %zu B of LDS", lds);
      return NVL_ELAUNCH;
    }
    attr_set = true;
  }
  // two row groups: pair them on one XCD (see the kernel); NVL_WIDE_PAIR=0 keeps the plain (tile, split, group) grid
  static const bool pair_ok = env_int("NVL_WIDE_PAIR", 1) != 0;
#ifdef NVL_PROBES
  static const int dbg = env_int("NVL_WIDE_DBG", 0);
#else
  constexpr int dbg = 0;
#endif
  if (p.mgroups == 2 && pair_ok) {
    const unsigned gx = (unsigned)((p.tiles + 7) / 8) * 16;
    hipLaunchKernelGGL((linear_wide_kernel<MT, NT, NW, EPI, RING, PACKED, BK, CORE>), dim3(gx, p.split, 1),
                       dim3((NW + wide_loaders_of(NW, CORE)) * 64), lds, s, (const bf16_t*)x, (const bf16_t*)w, out, (int)m, n, k, p.steps, p.tiles, dbg);
    return NVL_OK;
  }
  hipLaunchKernelGGL((linear_wide_kernel<MT, NT, NW, EPI, RING, PACKED, BK, CORE>), dim3(p.tiles, p.split, p.mgroups),
                     dim3((NW + wide_loaders_of(NW, CORE)) * 64), lds, s, (const bf16_t*)x, (const bf16_t*)w, out, (int)m, n, k, p.steps, 0, dbg);
  return NVL_OK;
}

thread_local bool g_packed = false;      // weight layout of the launch being dispatched (set by nvl_linear_wide)
thread_local int g_mode = 0;             // ... and its output mode (a split-K bf16 / SiLU call dispatches the PARTIAL kernels)

// ---- four-consumer tile kernel: when, and with how many column tiles per workgroup ---------------------------------------
// One row group of 10-16 row tiles on tile-packed weights. The K split is the plan's (nvl_linear_wide_plan reports it without
// knowing the layout); the column tiles per workgroup are chosen for whole rounds of 256 workgroups: fewer, fatter workgroups
// re-read x less often but a ragged last round costs a full one (Qwen3-32B gate_up: 534 workgroups of 6 tiles = 3 rounds,
// 400 of 8 = 2).
// Where it is used (round-6 A/B of every Qwen3-8B / 14B / 32B projection at 160 / 192 / 208 / 256 rows against the
// one-wave-per-SIMD kernel, profiles/r06_gemm_tile4_ab_all_shapes.json): bf16 and SiLU outputs — at 145-192 rows always
// (+2 ... +25 %: 32B gate_up 150 vs 190 us, per-rank TP gate_up 33 vs 38), at 193-256 rows when its whole-round count beats
// the 96-column tiling's (32B gate_up 191 vs 222 us, qkv 48.5 vs 49.4); the fp32-slab projections (o / down: small N, deep
// split K) are a wash or lose (8B down at 192 rows 48.8 vs 41.2) and keep the kernel above. NVL_WIDE_TILE4=0 / 2 = never /
// wherever it can run; NVL_WIDE_CT forces the column tiles.
int tile4_ct(const WidePlan& p, int mode, int n, int k) {
  const int on = env_int("NVL_WIDE_TILE4", 1), force_ct = env_int("NVL_WIDE_CT", 0);   // (host side of a launch: read per call)
  if (!on || !g_packed || p.mgroups != 1 || (p.mt != 12 && p.mt != 16) || k % 64) return 0;
  if (on != 2 && g_mode == EPI_PARTIAL) return 0;
  const int units = (mode == EPI_SILU ? n / 2 : n) / 16;           // column tiles of the output (SiLU: gate / up pairs)
  int best = 0;
  double best_t = 1e30;
  for (int ct : {8, 6, 4}) {
    if (force_ct && ct != force_ct) continue;
    const int per_wg = mode == EPI_SILU ? ct / 2 : ct;
    const int64_t wgs = (int64_t)((units + per_wg - 1) / per_wg) * p.split;
    const double t = (double)((wgs + 255) / 256) * (0.15 + 0.075 * ct);   // us per 64-column step of a workgroup, roughly
    if (t < best_t) { best_t = t; best = ct; }
  }
  if (on != 2 && p.mt == 16) {
    // ... against the shipped tiling of the same plan (nw x nt x 16 columns per workgroup, ~ a 6-tile workgroup's step)
    const double t_wide = (double)(((int64_t)p.tiles * p.split + 255) / 256) * 0.57;
    if (best_t >= t_wide) return 0;
  }
  return best;
}

template <int RT, int CT, int EPI>
int launch_tile4(const WidePlan& p, int mode, const void* x, const void* w, void* out, int64_t m, int n, int k, hipStream_t s) {
  const size_t lds = (size_t)kT4XStages * RT * 64 * 128 + (size_t)kT4WStages * CT * 2048;
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& attr_set = attr_done[nvl_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_tile4_kernel<RT, CT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      nvl_set_error("nvl_linear_wide: cannot reserve %zu B of LDS", lds);
      return NVL_ELAUNCH;
    }
    attr_set = true;
  }
  const int units = (EPI == EPI_SILU ? n / 2 : n) / 16, per_wg = EPI == EPI_SILU ? CT / 2 : CT;
#ifdef NVL_PROBES
  static const int dbg = env_int("NVL_WIDE_DBG", 0);
#else
  constexpr int dbg = 0;
#endif
  hipLaunchKernelGGL((linear_tile4_kernel<RT, CT, EPI>), dim3((unsigned)((units + per_wg - 1) / per_wg), p.split, 1), dim3(8 * 64), lds, s,
                     (const bf16_t*)x, (const bf16_t*)w, out, (int)m, n, k, (k / 64) / p.split, dbg);
  return NVL_OK;
}

template <int EPI>
int dispatch_tile4(int ct, const WidePlan& p, int mode, const void* x, const void* w, void* out, int64_t m, int n, int k, hipStream_t s) {
#define NVL_T4(R_, C_) if (p.mt == R_ * 4 && ct == C_) return launch_tile4<R_, C_, EPI>(p, mode, x, w, out, m, n, k, s);
  NVL_T4(4, 8) NVL_T4(4, 6) NVL_T4(4, 4) NVL_T4(3, 8) NVL_T4(3, 6) NVL_T4(3, 4)
#undef NVL_T4
  nvl_set_error("nvl_linear_wide: internal plan error (tile4 mt=%d ct=%d)", p.mt, ct);
  return NVL_EINVAL;
}

template <int MT, int NT, int NW, int EPI>
int launch_wide(const WidePlan& p, const void* x, const void* w, void* out, int64_t m, int n, int k, hipStream_t s) {
  if constexpr (MT > 12) {                                         // the 64-column step exists for the 13-16 row-tile form only
    if constexpr (MT == 16 && NT == 2 && NW == 3) {                 // ... whose consumer loop is the hand-scheduled core
      static const bool core_on = env_int("NVL_WIDE_CORE", 1) != 0;  // (0: hipcc's schedule of the same decomposition, A/B)
      if (p.bk == 64 && g_packed && core_on) return launch_wide_l<MT, NT, NW, EPI, true, 64, true>(p, x, w, out, m, n, k, s);
    }
    if (p.bk == 64)
      return g_packed ? launch_wide_l<MT, NT, NW, EPI, true, 64>(p, x, w, out, m, n, k, s)
                      : launch_wide_l<MT, NT, NW, EPI, false, 64>(p, x, w, out, m, n, k, s);
  }
  return g_packed ? launch_wide_l<MT, NT, NW, EPI, true, 128>(p, x, w, out, m, n, k, s)
                  : launch_wide_l<MT, NT, NW, EPI, false, 128>(p, x, w, out, m, n, k, s);
}

template <int NT, int NW, int EPI>
int dispatch_mt(const WidePlan& p, const void* x, const void* w, void* out, int64_t m, int n, int k, hipStream_t s) {
  switch (p.mt) {
    case 1: return launch_wide<1, NT, NW, EPI>(p, x, w, out, m, n, k, s);
    case 2: return launch_wide<2, NT, NW, EPI>(p, x, w, out, m, n, k, s);
    case 3: return launch_wide<3, NT, NW, EPI>(p, x, w, out, m, n, k, s);
    case 5: return launch_wide<5, NT, NW, EPI>(p, x, w, out, m, n, k, s);
    case 7: return launch_wide<7, NT, NW, EPI>(p, x, w, out, m, n, k, s);
    case 9: return launch_wide<9, NT, NW, EPI>(p, x, w, out, m, n, k, s);
  }
  if constexpr (NW == 3 || NT == 1) {
    switch (p.mt) {
      case 12: return launch_wide<12, NT, NW, EPI>(p, x, w, out, m, n, k, s);
      case 16: return launch_wide<16, NT, NW, EPI>(p, x, w, out, m, n, k, s);
    }
  }
  nvl_set_error("nvl_linear_wide: internal plan error (mt=%d nt=%d nw=%d)", p.mt, p.nt, p.nw);
  return NVL_EINVAL;
}

template <int EPI>
int dispatch_wide(const WidePlan& p, const void* x, const void* w, void* out, int64_t m, int n, int k, hipStream_t s) {
  if (const int ct = tile4_ct(p, EPI, n, k)) return dispatch_tile4<EPI>(ct, p, EPI, x, w, out, m, n, k, s);
#define NVL_W_CASE(NT_, NW_) \
  if (p.nt == NT_ && p.nw == NW_) return dispatch_mt<NT_, NW_, EPI>(p, x, w, out, m, n, k, s);
  if constexpr (EPI != EPI_SILU) { NVL_W_CASE(1, 3) NVL_W_CASE(1, 4) }
  NVL_W_CASE(2, 3) NVL_W_CASE(2, 4)
#undef NVL_W_CASE
  nvl_set_error("nvl_linear_wide: internal plan error (nt=%d nw=%d)", p.nt, p.nw);
  return NVL_EINVAL;
}

}  // namespace

extern "C" int nvl_linear_wide_plan(int64_t m, int n, int k, int mode, int* splits, size_t* workspace_bytes) {
  WidePlan p;
  if (mode < 0 || mode > 2 || !wide_plan(m, n, k, mode, &p)) return 0;
  if (splits) *splits = p.split;
  if (workspace_bytes) *workspace_bytes = mode != EPI_PARTIAL && p.split > 1 ? (size_t)p.split * m * n * sizeof(float) : 0;
  return 1;
}

extern "C" int nvl_pack_weight_tiles(const void* weight, void* packed, int64_t n, int64_t k, void* stream) {
  NVL_REQUIRE(weight && packed && weight != packed, "nvl_pack_weight_tiles: null or aliased pointers");
  NVL_REQUIRE(n > 0 && k > 0 && n % 16 == 0 && k % 32 == 0 && k < (1ll << 31), "nvl_pack_weight_tiles: n=%lld must be a multiple of 16, k=%lld of 32",
              (long long)n, (long long)k);
  NVL_REQUIRE(((uintptr_t)weight | (uintptr_t)packed) % 16 == 0, "nvl_pack_weight_tiles: pointers must be 16-byte aligned");
  const int64_t chunks = n * k / 8;
  NVL_REQUIRE((chunks + 255) / 256 < (1ll << 31), "nvl_pack_weight_tiles: matrix too large");
  hipLaunchKernelGGL(pack_weight_tiles_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)weight, (bf16_t*)packed, chunks, (int)k);
  return nvl_check_launch("nvl_pack_weight_tiles");
}

extern "C" int nvl_linear_wide(const void* x, const void* weight, void* out, int64_t m, int n, int k, int mode,
                               int weight_layout, void* workspace, size_t workspace_bytes, void* stream) {
  NVL_REQUIRE(x && weight && out, "nvl_linear_wide: null pointer");
  NVL_REQUIRE(mode >= 0 && mode <= 2, "nvl_linear_wide: mode=%d (0 bf16, 1 silu*mul, 2 split-K fp32 partials)", mode);
  NVL_REQUIRE(weight_layout == 0 || weight_layout == 1, "nvl_linear_wide: weight_layout=%d (0 row-major [N, K], 1 tile-packed)", weight_layout);
  g_packed = weight_layout == 1;
  g_mode = mode;
  NVL_REQUIRE(((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)workspace) % 16 == 0,
              "nvl_linear_wide: pointers must be 16-byte aligned");
  WidePlan p;
  if (!wide_plan(m, n, k, mode, &p)) {
    nvl_set_error("nvl_linear_wide: shape m=%lld n=%d k=%d mode=%d not covered (query nvl_linear_wide_plan first)",
                  (long long)m, n, k, mode);
    return NVL_EUNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  if (mode == EPI_PARTIAL) {
    const int rc = dispatch_wide<EPI_PARTIAL>(p, x, weight, out, m, n, k, s);
    return rc != NVL_OK ? rc : nvl_check_launch("nvl_linear_wide");
  }
  if (p.split == 1) {
    const int rc = mode == EPI_SILU ? dispatch_wide<EPI_SILU>(p, x, weight, out, m, n, k, s)
                                    : dispatch_wide<EPI_BF16>(p, x, weight, out, m, n, k, s);
    return rc != NVL_OK ? rc : nvl_check_launch("nvl_linear_wide");
  }
  const size_t need = (size_t)p.split * m * n * sizeof(float);
  NVL_REQUIRE(workspace && workspace_bytes >= need, "nvl_linear_wide: workspace %zu B < required %zu B", workspace_bytes,
              need);
  // split-K with a bf16 / SiLU output: the GEMM runs as if its columns were independent (a SiLU pair is formed by
  // the reduce kernel), slabs go to the workspace
  WidePlan q = p;
  if (mode == EPI_SILU) {   // plain column tiling over all N = gate | up columns
    const int cols = q.nw * q.nt * 16;
    q.tiles = (n + cols - 1) / cols;
  }
  int rc = dispatch_wide<EPI_PARTIAL>(q, x, weight, workspace, m, n, k, s);
  if (rc != NVL_OK) return rc;
  const int out_cols = mode == EPI_SILU ? n / 2 : n;
  const int64_t quads = m * (int64_t)(out_cols / 4);
  const unsigned blocks = (unsigned)((quads + 255) / 256);
  if (mode == EPI_SILU)
    hipLaunchKernelGGL(slab_reduce_kernel<EPI_SILU>, dim3(blocks), dim3(256), 0, s, (const float*)workspace, p.split,
                       (int64_t)m * n, (bf16_t*)out, (int)m, n);
  else
    hipLaunchKernelGGL(slab_reduce_kernel<EPI_BF16>, dim3(blocks), dim3(256), 0, s, (const float*)workspace, p.split,
                       (int64_t)m * n, (bf16_t*)out, (int)m, n);
  return nvl_check_launch("nvl_linear_wide");
}
