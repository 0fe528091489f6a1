// Varlen causal (bottom-right aligned) GQA prefill attention for gfx950 on the matrix cores.
// Replaces flash_attn_varlen_func as used by nano-vllm layers/attention.py:64-70, both K/V
// sources: packed [sum Lk, Hkv, 128] tensors, or the paged cache + block table (prefix cache /
// chunked-prefill continuation).
//
// Structure (flash-style, one workgroup = 4 waves = 128 query rows of one (sequence, q-head)):
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, operands swapped so that each LANE owns one
//     query column: the softmax row-reduction is 31 in-lane max/adds + one cross-half exchange.
//   * K tile (64 keys x 128) staged in LDS with a 16-byte XOR swizzle -> conflict-free
//     ds_read_b128 A-fragments; V tile staged row-major (320-byte row stride) and fed to the
//     P.V MFMA through ds_read_b64_tr_b16 (hardware transpose read).
//   * O^T = V^T . P^T accumulates in registers (lane = query column, so the online-softmax
//     rescale is lane-local). The MFMA k-slot <-> key mapping of the P.V product is permuted
//     to match the S^T accumulator layout, so P never moves between lanes.
//   * online softmax in fp32 (base-2), P rounded to bf16 before P.V, fp32 accumulation.
//   * K/V tiles are double-buffered in LDS and staged through registers with the issue-early /
//     write-late split (cdna_hip_programming.md T14): the global loads of tile t+1 are issued after
//     the QK^T of tile t and written to the other LDS buffer after its P.V, so HBM/L2 latency hides
//     under the softmax / P.V phase and there is ONE workgroup barrier per tile.
//   * a wave skips the MFMA work of key tiles that lie entirely above its 32 rows' causal frontier.
// The (sequence, q-block) of a workgroup is found on device from cu_seqlens_q — in registers (shuffle scan + ballot +
// readlane, one global round trip, no barrier) for launches of up to 64 sequences, through an LDS prefix + binary
// search beyond that — so no host-side tile list is needed.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kQBlk = 128;   // query rows per workgroup (4 waves x 32; with 8 waves: of two q-heads)
constexpr int kKBlk = 64;    // keys per tile
constexpr int kKRowB = 256;  // K tile row bytes in LDS
constexpr int kVRowB = 320;  // V tile row bytes in LDS (256 + 64 pad: conflict-free tr reads)
constexpr float kNegBig = -1.0e30f;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4_t& w) { return __builtin_bit_cast(bf16x8_t, w); }

struct SeqTile {
  int seq, qblk;
};

constexpr int kTileBytes = kKBlk * (kKRowB + kVRowB);   // one K + V tile pair in LDS

// KV8 (PAGED only): the cache holds OCP fp8 e4m3 (128 bytes per (token, head) row); a staged tile is converted to
// bf16 on its way into LDS (exact), everything downstream is unchanged.
// NW = 4: one workgroup = 4 waves = 128 query rows of ONE q-head (two such workgroups per CU).
// NW = 8: one workgroup = 8 waves = the same 128 query rows of TWO q-heads of one kv group (waves 0-3 / 4-7): the
//         K/V tile is fetched from L2/HBM and written to LDS once for both heads — half the staging loads, LDS
//         writes and prologue work per MFMA, at the price of an 8-wave barrier domain (one workgroup per CU).
// (Round 6, two more shapes built, correct on this file's whole test suite, measured slower and retired with their numbers —
// tools/probes/rejected/attn_prefill_pingpong_8wave.hip.txt: the two halves of the 8-wave workgroup one barrier interval apart,
// matrix interval beside vector interval, 0.92 vs 0.94 PF; attn_prefill_w64_one_wave_per_simd.hip.txt: 64 rows per wave, O and Q
// in AGPRs, asm MFMAs in source order between compiler-generated softmax slices, 0.84 vs 0.95 PF. What they established
// (profiles/r06_prefill_pp_*.txt): under this kernel's MFMA load the chip clocks at ~1.56 GHz (an MFMA-only skeleton of the
// loop reaches 1.42 PF = 87 % of what 1.56 GHz allows), the wave that loses a SIMD's issue arbitration runs at 0.3-0.4 of its
// rate beside its partner, and a lone wave needs <= 7 issue slots per MFMA where hipcc-generated slices cost 12.)
// (A PERSISTENT form — workgroups walking the longest-first item list in a snake, the next item's Q rows and first K/V
// tile requested during the last tile of the current one — was built and measured in round 4: +13 % on equal-length
// batches, -8 ... -12 % on the bench's ragged ones, a dynamic per-XCD ticket queue slower than both. Retired:
// tools/probes/rejected/attn_prefill_persistent_walk.hip.txt, profiles/r04_prefill_persist_ab.json.)
template <bool PAGED, bool KV8, int NW>
__global__ __launch_bounds__(NW * 64, 2) void prefill_attn_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t k_tok_stride,
    int64_t v_tok_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, bf16_t* __restrict__ out, int num_seqs, int hq,
    int hkv, int block_size, float scale_log2e, int xcd_map, float* __restrict__ lse, float rescale_thr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // two {K: 64 x 256 B, V: 64 x 320 B} tile buffers, then the tile-lookup scratch
  constexpr int NT = NW * 64;                            // threads per workgroup
  int* wsum = reinterpret_cast<int*>(smem + 2 * kTileBytes);
  int* pre = wsum + 8;                                  // [num_seqs + 1]

  const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
  const int wave = wave_all & 3;                        // 32-row block of the 128-row q tile
  const int hsub = wave_all >> 2;                       // which of the workgroup's q-heads (NW == 8)
  const int qcol = lane & 31, hi = lane >> 5;

  // Longest-first dispatch: workgroups are launched in block-id order and a q-block's work grows with its
  // index (causal), so tiles are taken from the END of the list and the head index varies fastest — the
  // short tiles fill the tail instead of the 64-tile ones (measured 1.2-1.4x on 4 x 4096 / 1 x 16384). (A batch-wide
  // longest-first order — every sequence's top block, then every next one — was measured too: +6 % at 16 x 1024 but
  // -4 % on ragged bench batches and -10 % at 8 x 2048 / G = 8: neighbouring workgroups then stream DIFFERENT sequences'
  // K/V and the working set of the workgroups in flight outgrows the L2s. profiles/r03_prefill_ab_s1_l{4,0}.json.)
  // Which (q-head, tile)? The G = hq / hkv query heads of a kv group stream the SAME K/V tiles. MI355X has 8 XCDs
  // with private L2s and hands workgroup b to XCD b % 8, so with the plain order (head fastest) the heads of a
  // group land on G different XCDs and every one of them pulls the tiles through its own L2. xcd_map: workgroups
  // are numbered so that the G heads of a (tile, kv-head) group occupy CONSECUTIVE slots of ONE XCD — they run
  // side by side at the same pace and all but the first hit that XCD's L2 (cdna_hip_programming.md T1; placement
  // only changes speed, never results).
  int head, tile_rank;
  // block id -> (q-head, rank of the tile in the longest-first list) under the XCD-aware numbering
  auto xcd_decode = [&](int b, int& head_, int& rank_) {
    const int G = hq / hkv;
    const int xcd = b & 7, slot = b >> 3;
    const int gi = slot / G, g = slot - gi * G;
    const int j = gi * 8 + xcd;                    // (tile, kv-head) group, longest tiles first
    rank_ = j / hkv;
    head_ = (j - rank_ * hkv) * G + g;
  };
  if (xcd_map) {
    xcd_decode(blockIdx.x, head, tile_rank);
  } else {
    head = NW == 8 ? blockIdx.x * 2 + hsub : blockIdx.x;
    tile_rank = blockIdx.y;
  }
  // ---- which (sequence, q-block) is this workgroup? ------------------------------------
  int seq, qblk, q0, lq, k0, lk;     // wave-uniform
  // the tile list in registers (launches of <= 64 sequences): lane i holds sequence i's bounds and tile prefix
  int a0 = 0, a1 = 0, b0 = 0, b1 = 0, val = 0, sc = 0, total = 0;
  auto reg_locate = [&](int rank_, int& seq_, int& qblk_, int& q0_, int& lq_, int& k0_, int& lk_) {
    const int tile = total - 1 - rank_;            // longest-first, see above
    seq_ = __popcll(__ballot(sc <= tile));         // inclusive prefixes <= tile: the sequences before ours
    qblk_ = tile - __builtin_amdgcn_readlane(sc - val, seq_);
    q0_ = __builtin_amdgcn_readlane(a0, seq_);
    lq_ = __builtin_amdgcn_readlane(a1, seq_) - q0_;
    k0_ = __builtin_amdgcn_readlane(b0, seq_);
    lk_ = __builtin_amdgcn_readlane(b1, seq_) - k0_;
  };
  if (num_seqs <= 64) {
    // Up to 64 sequences (every prefill batch of the bench): each WAVE derives the tile list by itself, in
    // registers — lane i holds sequence i's cu_seqlens entries, a shuffle scan gives the tile prefix, a ballot finds
    // the sequence, readlane fetches its bounds. One global round trip, no LDS, no workgroup barrier, no dependent
    // re-load of cu_seqlens[seq] (the LDS form below costs three barriers, a binary search out of LDS and a second
    // dependent round trip per workgroup: ~1 us of a short sequence's ~5 us workgroup).
    if (lane < num_seqs) {
      a0 = cu_q[lane]; a1 = cu_q[lane + 1];
      b0 = cu_k[lane]; b1 = cu_k[lane + 1];
    }
    val = (a1 - a0 + kQBlk - 1) / kQBlk;
    sc = val;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(sc, o, 64);
      if (lane >= o) sc += n;
    }
    total = __builtin_amdgcn_readlane(sc, 63);
    if (tile_rank >= total) return;  // grid is an upper bound
    reg_locate(tile_rank, seq, qblk, q0, lq, k0, lk);
  } else {
    int carry = 0;
    if (tid == 0) pre[0] = 0;
    for (int base = 0; base < num_seqs; base += NT) {
      const int i = base + tid;
      int val = 0;
      if (i < num_seqs) val = (cu_q[i + 1] - cu_q[i] + kQBlk - 1) / kQBlk;
      int s = val;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(s, o, 64);
        if (lane >= o) s += n;
      }
      if (lane == 63) wsum[wave_all] = s;
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int t = wsum[w];
        if (w < wave_all) woff += t;
        tot += t;
      }
      if (i < num_seqs) pre[i + 1] = carry + woff + s;
      carry += tot;
      __syncthreads();
    }
    __syncthreads();
    if (tile_rank >= pre[num_seqs]) return;  // grid is an upper bound
    const int tile = pre[num_seqs] - 1 - tile_rank;
    int lo = 0, hi_s = num_seqs;
    while (hi_s - lo > 1) {
      const int mid = (lo + hi_s) >> 1;
      if (pre[mid] <= tile) lo = mid; else hi_s = mid;
    }
    // wave-uniform by construction, but read out of LDS: tell the compiler (scalar registers, scalar loads of
    // cu_seqlens / block tables, SGPR-based K/V addressing whose VGPR offsets stay live across the loop)
    seq = __builtin_amdgcn_readfirstlane(lo);
    qblk = __builtin_amdgcn_readfirstlane(tile - pre[seq]);
    q0 = cu_q[seq]; lq = cu_q[seq + 1] - q0;
    k0 = cu_k[seq]; lk = cu_k[seq + 1] - k0;
  }
  // Both branches produce wave-uniform values, but after the merge hipcc no longer KNOWS that: left alone it keeps the
  // K/V buffer descriptors in VGPRs and wraps every tile load in a readfirstlane "waterfall" loop (~10 instructions
  // per load, loads serialised — cdna_hip_programming.md T20).
  seq = __builtin_amdgcn_readfirstlane(seq);
  qblk = __builtin_amdgcn_readfirstlane(qblk);
  q0 = __builtin_amdgcn_readfirstlane(q0);
  lq = __builtin_amdgcn_readfirstlane(lq);
  k0 = __builtin_amdgcn_readfirstlane(k0);
  lk = __builtin_amdgcn_readfirstlane(lk);
  head = __builtin_amdgcn_readfirstlane(head);   // (8-wave shape: derived from the wave id, uniform per wave)
  // geometry of the workgroup's item
  int kvh, off, qi, qi_c, kv_end;
  bool q_valid;
  auto item_geometry = [&]() {
    kvh = head / (hq / hkv);
    off = lk - lq;  // bottom-right alignment: query i sees keys j <= i + off
    qi = qblk * kQBlk + wave * 32 + qcol;
    q_valid = qi < lq;
    qi_c = q_valid ? qi : lq - 1;
    kv_end = min(lk, qblk * kQBlk + kQBlk + off);  // keys visible to the block's last query
  };
  item_geometry();

  // ---- Q fragments: B operand of S^T = K.Q^T : lane (query, hi) holds d = ds*16 + 8*hi .. +8
  bf16x8_t qf[8];
  auto load_q = [&](int q0_, int lq_, int qblk_, int head_) {
    int row = qblk_ * kQBlk + wave * 32 + qcol;
    row = row < lq_ ? row : lq_ - 1;
    const bf16_t* qp = q + ((int64_t)(q0_ + row) * hq + head_) * 128 + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = as_bf16x8(*reinterpret_cast<const u32x4_t*>(qp + ds * 16));
  };
  load_q(q0, lq, qblk, head);
  // Keep the Q loads ahead of the first tile's staging loads: the prologue's wait for that tile then also
  // retires them. (If hipcc sinks them to the loop head, its waitcnt pass keeps per-fragment vmcnt waits
  // inside the loop, which drain the in-flight staging loads in the middle of every QK^T phase.)
  __builtin_amdgcn_sched_barrier(0);

  f32x16_t oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = kNegBig, l_run = 0.f;

  int kmax_vis = qi_c + off;  // last key this query may see
  // loop-invariant LDS byte offsets of this lane's K fragments (row qcol, swizzled slot) and V transpose reads
  int kslot[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) kslot[ds] = qcol * kKRowB + (((ds * 2 + hi) ^ (qcol & 15)) << 4);
  const int i16 = lane & 15;
  const int vlane = (4 * hi + (i16 >> 2)) * kVRowB + (16 * ((lane >> 4) & 1) + (i16 & 3) * 4) * 2;
  const f32x16_t kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- K/V staging: each thread moves 4 16-byte chunks of K and of V per tile ------------------
  u32x4_t kreg[4], vreg[4];
  // this thread's chunk n of a tile: row (tid >> 4) + 16 n, 16-byte column tid & 15; element offsets from
  // the tile's first row are loop-invariant (no 64-bit multiplies in the loop)
  const int64_t kstride = PAGED ? 128 : k_tok_stride, vstride = PAGED ? 128 : v_tok_stride;
  constexpr int kCh = 1024 / NT;                         // 16-byte chunks of a bf16 K (or V) tile per thread
  constexpr int kCh8 = 512 / NT;                         // ... of an fp8 tile
  const int srow = tid >> 4, sc16 = tid & 15;
  unsigned int koff[4], voff[4];   // BYTE offsets, unsigned: the loads use the SGPR-base + 32-bit VGPR offset form
#pragma unroll
  for (int n = 0; n < kCh; ++n) {
    koff[n] = ((unsigned int)((srow + n * (NT / 16)) * kstride) + sc16 * 8) * 2u;
    voff[n] = ((unsigned int)((srow + n * (NT / 16)) * vstride) + sc16 * 8) * 2u;
  }
  // Loads go through buffer descriptors (SGPR base + SGPR tile offset + the loop-invariant VGPR byte
  // offsets above): no per-tile address VALU, nothing the loads depend on is rewritten while they are in
  // flight, and rows past the end of the sequence / block come back as zeros from the hardware range check
  // (they are masked anyway).
  // issue the global loads of the tile starting at key kt of the item (sequence seq_, kv-head kvh_, keys k0_ .. k0_ + lk_)
  auto stage_load_of = [&](int seq_, int kvh_, int k0_, int lk_, int kt) {
    __amdgpu_buffer_rsrc_t krs, vrs;
    int ksoff, vsoff;
    if constexpr (PAGED) {
      const int blk = block_tables[(int64_t)seq_ * bt_stride + kt / block_size];
      const int64_t base = (((int64_t)blk * hkv + kvh_) * block_size + (kt % block_size)) * 128;
      if constexpr (KV8) {
        krs = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)k + base), 0, kKBlk * 128, 0x00020000);
        vrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)v + base), 0, kKBlk * 128, 0x00020000);
      } else {
        krs = __builtin_amdgcn_make_buffer_rsrc((void*)(k + base), 0, kKBlk * 256, 0x00020000);
        vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(v + base), 0, kKBlk * 256, 0x00020000);
      }
      ksoff = vsoff = 0;
    } else {
      krs = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)k0_ * k_tok_stride + kvh_ * 128), 0,
                                              (int)(lk_ * kstride * 2), 0x00020000);
      vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)k0_ * v_tok_stride + kvh_ * 128), 0,
                                              (int)(lk_ * vstride * 2), 0x00020000);
      ksoff = (int)(kt * kstride * 2);
      vsoff = (int)(kt * vstride * 2);
    }
    if constexpr (KV8) {
      // 64 keys x 128 B = 512 16-byte chunks (chunk = tid + NT n -> row chunk >> 3, 16 elements)
#pragma unroll
      for (int n = 0; n < kCh8; ++n) {
        kreg[n] = __builtin_amdgcn_raw_buffer_load_b128(krs, (tid + n * NT) * 16, ksoff, 0);
        vreg[n] = __builtin_amdgcn_raw_buffer_load_b128(vrs, (tid + n * NT) * 16, vsoff, 0);
      }
    } else {
#pragma unroll
      for (int n = 0; n < kCh; ++n) {
        kreg[n] = __builtin_amdgcn_raw_buffer_load_b128(krs, koff[n], ksoff, 0);
        vreg[n] = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff[n], vsoff, 0);
      }
    }
  };
  auto stage_load = [&](int kt) { stage_load_of(seq, kvh, k0, lk, kt); };
  auto stage_write = [&](int buf) {     // registers -> LDS tile buffer `buf`
    unsigned char* kl = smem + buf * kTileBytes;
    unsigned char* vl = kl + kKBlk * kKRowB;
    if constexpr (KV8) {
#pragma unroll
      for (int n = 0; n < kCh8; ++n) {
        const int chunk = tid + n * NT;
        const int row = chunk >> 3, c16 = (chunk & 7) * 2;          // two bf16 16-byte chunks per fp8 chunk
        u32x4_t a, b;
        fp8x16_to_bf16(kreg[n], &a, &b);
        *reinterpret_cast<u32x4_t*>(kl + row * kKRowB + ((c16 ^ (row & 15)) << 4)) = a;
        *reinterpret_cast<u32x4_t*>(kl + row * kKRowB + (((c16 + 1) ^ (row & 15)) << 4)) = b;
        fp8x16_to_bf16(vreg[n], &a, &b);
        *reinterpret_cast<u32x4_t*>(vl + row * kVRowB + (c16 << 4)) = a;
        *reinterpret_cast<u32x4_t*>(vl + row * kVRowB + ((c16 + 1) << 4)) = b;
      }
    } else {
#pragma unroll
      for (int n = 0; n < kCh; ++n) {
        const int chunk = tid + n * NT;
        const int row = chunk >> 4, c16 = chunk & 15;
        *reinterpret_cast<u32x4_t*>(kl + row * kKRowB + ((c16 ^ (row & 15)) << 4)) = kreg[n];
        *reinterpret_cast<u32x4_t*>(vl + row * kVRowB + (c16 << 4)) = vreg[n];
      }
    }
  };
  // keys this WAVE's 32 rows can see: tiles starting above wave_kmax carry no work for it
  // (a wave whose 32 rows all lie past the end of the sequence — the tail of the last q-block — has no work at all:
  // it only helps staging the tiles; +3-4 % on batches of short sequences)
  int wave_kmax_s, wave_kmin_s;         // wave-uniform copies on the scalar side (scalar branches instead of exec masks)
  auto wave_frontier = [&]() {
    const bool wave_has_rows = qblk * kQBlk + wave * 32 < lq;
    const int wave_kmax = wave_has_rows ? min(qblk * kQBlk + wave * 32 + 31, lq - 1) + off : -1;
    const int wave_kmin = min(qblk * kQBlk + wave * 32, lq - 1) + off;   // ... and all of them see keys <= wave_kmin
    wave_kmax_s = __builtin_amdgcn_readfirstlane(wave_kmax);
    wave_kmin_s = __builtin_amdgcn_readfirstlane(wave_kmin);
  };
  wave_frontier();

  // kv_end >= 1 always (lq >= 1, off >= 0): unconditional, with an explicit vmcnt(0) — hipcc's waitcnt pass then KNOWS
  // the Q fragment loads have landed before the loop (with a conditional prologue it assumes they may be pending at the
  // loop head and puts vmcnt waits on their first uses inside QK^T, which drain whatever tile loads are in flight).
  stage_load(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  // a (no-op) use of the Q fragments HERE: without it LLVM sinks their loads into the loop preheader, behind this wait
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) asm volatile("" : "+v"(qf[ds]));
  stage_write(0);
  __syncthreads();


  // ---- S^T tile: 2 key blocks x 32 keys; lane = query column ---------------------------------
  auto qk = [&](f32x16_t (&sacc)[2], const unsigned char* k_lds) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // A operand: lane (key row kb*32 + qcol, hi) holds d = ds*16 + 8*hi .. +8 (swizzled 16-byte slot)
      const unsigned char* kr = k_lds + kb * 32 * kKRowB;
      sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
          as_bf16x8(*reinterpret_cast<const u32x4_t*>(kr + kslot[0])), qf[0], kZero16, 0, 0, 0);
#pragma unroll
      for (int ds = 1; ds < 8; ++ds)
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            as_bf16x8(*reinterpret_cast<const u32x4_t*>(kr + kslot[ds])), qf[ds], sacc[kb], 0, 0, 0);
    }
    // K fragment reads pinned three ahead of their MFMA (16 ds_read_b128, 16 MFMAs)
    // (five ahead, and the exp / sum pairs as v_pk_fma_f32 / v_pk_add_f32: both within noise, profiles/README.md round 6)
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  };
  auto softmax_pv = [&](f32x16_t (&sacc)[2], int kmin_w, const int kt, const unsigned char* v_lds) {
    // ---- online softmax (base 2; the softmax scale is folded into the exponent's FMA) --------------
    // Only tiles that straddle this wave's causal frontier need the per-element mask.
    if (kt + kKBlk - 1 > kmin_w) {
      // key(kb, r) = kt + 4 hi + c, c = kb*32 + (r & 3) + 8 (r >> 2) a compile-time constant: one subtraction, then a
      // compare-with-immediate + select per score
      const int lim = kmax_vis - kt - 4 * hi;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sacc[kb][r] = (kb * 32 + (r & 3) + 8 * (r >> 2)) <= lim ? sacc[kb][r] : kNegBig;
    }
    float mx = sacc[0][0], mx1 = sacc[1][0];   // two chains: half the dependent depth
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      mx = fmaxf(mx, sacc[0][r]);
      mx1 = fmaxf(mx1, sacc[1][r]);
    }
    mx = fmaxf(mx, mx1);
    {   // the other half-wave holds the same query's other 32 keys: one v_permlane32_swap (VALU) instead of a
        // ds_bpermute round trip through the LDS in the middle of every tile's dependent chain
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float m_new = fmaxf(m_run, mx * scale_log2e);
    // Deferred rescale (cdna_hip_programming.md T13): O and l are brought to the new maximum only when some row's maximum
    // grew by more than rescale_thr (log2 units); until then the tile's P are taken against the STALE maximum (P <= 2^thr
    // instead of <= 1: the same relative rounding, no overflow at thr = 8) and nothing else changes — O / l and LSE = m +
    // log2 l are the same quantities. The order is the safe one: the previous tile's P.V is complete, the tile's P are
    // exponentiated after the decision, l and O take the same factor. (rescale_thr = 0: rescale whenever a maximum moves,
    // as flash-attn does.) Measured +3-6 % on long prompts: on random data the branch is taken in ~40 % of the tiles of a
    // 16 k prompt at thr = 0 (33 register-pair multiplies each), in the first tile only at thr = 8
    // (profiles/r06_prefill_deferred_rescale_ab.json).
    if (__any(m_new > m_run + rescale_thr)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      m_run = m_new;
    }
    float psum[2] = {0.f, 0.f};
    bf16x8_t pf[2][2];  // [kb][r0]: P^T fragment (B operand), k-slot (hi, e) <-> acc reg r0*8 + e
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r0 = 0; r0 < 2; ++r0)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r0 * 8 + e], scale_log2e, -m_run));
          psum[0] += p;
          pf[kb][r0][e] = (bf16_t)p;
        }
    l_run += psum[0] + psum[1];

    // ---- O^T += V^T . P^T : A operand lane (d = lane&31, hi) needs V[key(hi, e)][d] ----------
    // key(hi, e) = kb*32 + 16*r0 + 4*hi + (e & 3) + 8*(e >> 2): two transpose reads of 4 keys.
    const unsigned char* vb = v_lds + vlane;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r0 = 0; r0 < 2; ++r0) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const unsigned char* p0 = vb + (kb * 32 + 16 * r0) * kVRowB + db * 64;   // compile-time offset
          const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(p0));
          const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(p0 + 8 * kVRowB));
          const s16x8_t a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), pf[kb][r0],
                                                             oacc[db], 0, 0, 0);
        }
      }
  };
  // One tile step, with the LDS buffer a COMPILE-TIME constant (the loop below is unrolled by the two buffers): every
  // ds_read / ds_write address is then a loop-invariant lane offset + an immediate, instead of ~25 VALU adds per tile
  // re-basing them on the buffer of the moment.
  auto tile_step = [&](auto buf_c, const int kt) {
    constexpr int buf = decltype(buf_c)::value;
    const bool more = kt + kKBlk < kv_end;
    const unsigned char* k_lds = smem + buf * kTileBytes;
    const unsigned char* v_lds = k_lds + kKBlk * kKRowB;
    // QK^T | issue the next tile's loads | softmax + P.V. The score registers are written and read under the same
    // (scalar) test and deliberately left unset on the skipped path: initialised, hipcc zero-fills all 32 of them at
    // every loop head (49 moves per tile; this form measured +9-12 % — profiles/r03_prefill_ab_*.json). The loads sit
    // after QK^T: issued at the top of the iteration (a full tile of latency cover) they measured 1-2 % slower.
    f32x16_t sacc[2];
    const bool active = kt <= wave_kmax_s;
    if (active) qk(sacc, k_lds);
    __builtin_amdgcn_sched_barrier(0);
    if (more) stage_load(kt + kKBlk);
    __builtin_amdgcn_sched_barrier(0);
    if (active) softmax_pv(sacc, wave_kmin_s, kt, v_lds);
    __builtin_amdgcn_sched_barrier(0);
    if (more) stage_write(buf ^ 1);     // the other buffer was last read one barrier ago
    __syncthreads();
  };
  // (Tried on top of this and dropped: a two-score-tile pipeline — QK^T of tile t+1 beside the softmax of tile t,
  // cdna_hip_programming.md T15 — with the interleave pinned block by block: 253-255 registers, 4-5 % SLOWER at
  // 1 x 16,384 than this loop; the rolled loop, 1-3 % slower. profiles/r03_prefill_ab_var{0,1,2}.json.)
  for (int kt = 0; kt < kv_end; kt += 2 * kKBlk) {
    tile_step(std::integral_constant<int, 0>{}, kt);
    if (kt + kKBlk >= kv_end) break;
    tile_step(std::integral_constant<int, 1>{}, kt + kKBlk);
  }

  // ---- epilogue: normalise and store O[query][d] ------------------------------------------------
  float l_tot;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = 1.f / l_tot;
  // optional log-sum-exp of the scaled scores per (query, head), natural log (flash-attn's softmax_lse): the running
  // max / sum live in the log2 domain here
  if (lse != nullptr && q_valid && hi == 0) lse[(int64_t)(q0 + qi) * hq + head] = 0.6931471805599453f * (m_run + log2f(l_tot));
  // A lane holds d = db*32 + 8*rg + 4*hi + (0..3) of its query row: 8 bytes per (db, rg), the other half-wave the 8
  // bytes next to them. One v_permlane32_swap per dword trades halves between two neighbouring groups, after which
  // every lane owns 16 contiguous bytes: 8 dwordx4 stores per lane instead of 16 dwordx2 (cdna_hip_programming.md T21;
  // the store tail of a short sequence's workgroup is issue-bound).
  bf16_t* op = out + ((int64_t)(q0 + qi_c) * hq + head) * 128 + 8 * hi;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
      const int r0 = rp * 8;
      const unsigned int ax = pack_bf16x2(oacc[db][r0 + 0] * inv, oacc[db][r0 + 1] * inv);
      const unsigned int ay = pack_bf16x2(oacc[db][r0 + 2] * inv, oacc[db][r0 + 3] * inv);
      const unsigned int bx = pack_bf16x2(oacc[db][r0 + 4] * inv, oacc[db][r0 + 5] * inv);
      const unsigned int by = pack_bf16x2(oacc[db][r0 + 6] * inv, oacc[db][r0 + 7] * inv);
      const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      // lanes 0-31: [own group 2rp | upper half's group 2rp] = d db*32 + 16 rp + 0..7; lanes 32-63: the next 8
      if (q_valid) *reinterpret_cast<u32x4_t*>(op + db * 32 + 16 * rp) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
    }
}

}  // namespace

// attn_prefill64.hip: the 64-rows-per-wave shape (one wave per SIMD, generated asm main loop) for long bf16 prompts; returns
// 1 when the launch does not fit it
int nvl_prefill_w64_launch(const void* q, const void* k, const void* v, int64_t k_tok_stride, int64_t v_tok_stride,
                           const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, const int32_t* block_tables, int64_t bt_stride,
                           int block_size, void* out, int64_t total_q, int num_seqs, int num_q_heads, int num_kv_heads,
                           float scale_log2e, float* lse, float rescale_thr, hipStream_t s);

extern "C" int nvl_attn_prefill_varlen(const void* q, const void* k, const void* v, int64_t k_tok_stride,
                                       int64_t v_tok_stride, const int32_t* cu_seqlens_q,
                                       const int32_t* cu_seqlens_k, const int32_t* block_tables, int64_t bt_stride,
                                       void* out, int64_t total_q, int num_seqs, int max_seqlen_q, int num_q_heads,
                                       int num_kv_heads, int block_size, int64_t num_blocks, float softmax_scale,
                                       int kv_dtype, float* lse, void* stream) {
  NVL_REQUIRE(q && k && v && cu_seqlens_q && cu_seqlens_k && out, "nvl_attn_prefill_varlen: null pointer");
  NVL_REQUIRE(kv_dtype == NVL_KV_BF16 || (kv_dtype == NVL_KV_FP8 && block_tables != nullptr),
              "nvl_attn_prefill_varlen: kv_dtype=%d (0 bf16; 1 fp8 e4m3 only with a paged cache)", kv_dtype);
  NVL_REQUIRE(total_q >= 0 && num_seqs >= 0 && num_seqs <= 32768, "nvl_attn_prefill_varlen: bad sizes (total_q=%lld, num_seqs=%d)", (long long)total_q, num_seqs);
  NVL_REQUIRE(num_kv_heads > 0 && num_q_heads > 0 && num_q_heads % num_kv_heads == 0 && num_q_heads <= 65535,
              "nvl_attn_prefill_varlen: Hq=%d must be a positive multiple of Hkv=%d", num_q_heads, num_kv_heads);
  NVL_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0,
              "nvl_attn_prefill_varlen: pointers must be 16-byte aligned");
  const bool paged = block_tables != nullptr;
  if (paged) {
    NVL_REQUIRE(block_size > 0 && block_size % kKBlk == 0 && num_blocks > 0 && bt_stride > 0,
                "nvl_attn_prefill_varlen: paged K/V needs block_size %% %d == 0 (got %d)", kKBlk, block_size);
  } else {
    NVL_REQUIRE(k_tok_stride % 8 == 0 && v_tok_stride % 8 == 0 && k_tok_stride >= (int64_t)num_kv_heads * 128 &&
                    v_tok_stride >= (int64_t)num_kv_heads * 128,
                "nvl_attn_prefill_varlen: bad K/V token strides");
  }
  if (total_q == 0 || num_seqs == 0) return NVL_OK;
  const int64_t tiles = (total_q + kQBlk - 1) / kQBlk + num_seqs;  // upper bound on sum ceil(Lq/128)
  NVL_REQUIRE(tiles < (1ll << 31), "nvl_attn_prefill_varlen: too many query tiles");
  const size_t lds = (size_t)2 * kTileBytes + 8 * sizeof(int) + (size_t)(num_seqs + 1) * sizeof(int);
  const float sl2 = softmax_scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  // XCD-aware workgroup numbering (see the kernel). Round 2's kernel, measured A/B on MI355X
  // (profiles/r02_prefill_xcd{0,1}.json): 4 x 4096 +1.5 %, 8 x 2048 / G = 8 +1.5 %, 16 x 1024 +0.7 %, bench-like
  // 29 x 561 -3 %, 1 x 16384 (16 / 8 heads) -12 %: co-locating a group's heads helps less than it hurts the
  // longest-first balance across XCDs.
  // Workgroup shape (see the kernel): 8 waves (two q-heads share the staged K/V tile) when the launch has long
  // sequences, 4 waves otherwise — measured (profiles/r02_prefill_waves{4,8}.json, TFLOP/s 4 / 8 waves): 1 x 16384
  // 908 / 931 (16/8 heads), 819 / 856 (8/1); 4 x 4096 693 / 718; 8 x 2048 G = 8 741 / 743; but 16 x 1024 497 / 422
  // and 29 x 561 343 / 326: short sequences have too few tiles per workgroup to amortise the 8-wave barrier.
  // NVL_PREFILL_WAVES=4|8 forces one shape; 8 needs an even group size Hq / Hkv.
  // Round 3, with the current kernel (profiles/r03_prefill_ab_xcd_*.json, 4-wave shape, two A/B pairs): the XCD-aware
  // numbering is +3...11 % on bench-like / ragged batches of 100-1024-token prompts, +-1 % on the long shapes, -2...4 %
  // on launches of > 64 very short sequences => ON by default for 4-wave launches of <= 64 sequences; NVL_PREFILL_XCD=0|1
  // forces it off / on for every launch.
  static int xcd_env = -2, waves = -1, w64 = 1;
  static float thr = 8.f;
  if (xcd_env == -2) {
    const char* we = getenv("NVL_PREFILL_W64");     // the 64-rows-per-wave shape: 0 never, 1 long packed prompts (default), 2 every packed launch
    if (we) w64 = we[0] - '0';
    const char* te = getenv("NVL_PREFILL_RESCALE_THR");   // log2 units; 0 = rescale whenever a row's maximum moves
    if (te) thr = (float)atof(te);
    const char* e = getenv("NVL_PREFILL_XCD");
    xcd_env = (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1;
    const char* w = getenv("NVL_PREFILL_WAVES");
    waves = (w && w[0] == '8') ? 8 : ((w && w[0] == '4') ? 4 : 0);
  }
  // the one-wave-per-SIMD shape with the generated asm main loop (attn_prefill64.hip): packed K / V, long prompts (its
  // 256-row q tiles waste rows on short sequences: mean length >= 1024 as well). Measured +3 % (8 x 2048, G = 8) ... +10 %
  // (1 x 16,384) over the 8-wave loop, profiles/r06_prefill_w64_asm_*.txt
  // (packed: its tile offsets are 32-bit buffer offsets — the K / V of the launch must span < 2 GiB; paged: 64-bit)
  const bool w64_fits = paged || total_q * (k_tok_stride > v_tok_stride ? k_tok_stride : v_tok_stride) * 2 < (1ll << 31);
  if (w64 && ((max_seqlen_q >= 2048 && total_q >= (int64_t)num_seqs * 1024) || w64 == 2) && num_seqs <= 64 &&
      kv_dtype == NVL_KV_BF16 && !waves && w64_fits) {
    const int rc = nvl_prefill_w64_launch(q, k, v, k_tok_stride, v_tok_stride, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride,
                                          block_size, out, total_q, num_seqs, num_q_heads, num_kv_heads, sl2, lse, thr, s);
    if (rc != 1) return rc;
  }
  const int want = waves ? waves : (max_seqlen_q >= 2048 ? 8 : 4);
  const bool eight_ok = (num_q_heads / num_kv_heads) % 2 == 0;
  const int xcd_map = xcd_env >= 0 ? xcd_env : ((want == 4 || !eight_ok) && num_seqs <= 64 ? 1 : 0);
  const bool eight = want == 8 && !xcd_map && eight_ok;
  dim3 grid((unsigned)(eight ? num_q_heads / 2 : num_q_heads), (unsigned)tiles);
  if (xcd_map) {
    const int64_t groups = tiles * num_kv_heads;
    int64_t blocks = ((groups + 7) / 8) * 8 * (num_q_heads / num_kv_heads);
    NVL_REQUIRE(blocks < (1ll << 31), "nvl_attn_prefill_varlen: too many workgroups (%lld)", (long long)blocks);
    grid = dim3((unsigned)blocks, 1);
  } else {
    NVL_REQUIRE(tiles <= 65535, "nvl_attn_prefill_varlen: too many query tiles (%lld)", (long long)tiles);
  }
  static size_t lds_caps[NVL_MAX_DEVICES] = {};   // dynamic LDS above 64 KiB must be opted into per kernel (and device)
  size_t& lds_cap = lds_caps[nvl_device_slot()];
  if (lds > lds_cap) {
    const size_t want = lds < 160 * 1024 ? lds + 16 * 1024 : lds;   // headroom: num_seqs moves it by a few KiB
    const size_t cap = want > 160 * 1024 ? 160 * 1024 : want;
    NVL_REQUIRE(lds <= 160 * 1024, "nvl_attn_prefill_varlen: %d sequences need %zu B of LDS (> 160 KiB)", num_seqs, lds);
#define NVL_PF_ATTR(P, K8, NWV)                                                                            \
    (hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_attn_kernel<P, K8, NWV>),                  \
                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess)
    if (NVL_PF_ATTR(true, false, 4) || NVL_PF_ATTR(true, true, 4) || NVL_PF_ATTR(false, false, 4) ||
        NVL_PF_ATTR(true, false, 8) || NVL_PF_ATTR(true, true, 8) || NVL_PF_ATTR(false, false, 8)) {
      nvl_set_error("nvl_attn_prefill_varlen: cannot reserve %zu B of LDS", cap);
      return NVL_ELAUNCH;
    }
#undef NVL_PF_ATTR
    lds_cap = cap;
  }
#define NVL_PF_LAUNCH(P, K8, NWV)                                                                                      \
  hipLaunchKernelGGL((prefill_attn_kernel<P, K8, NWV>), grid, dim3(NWV * 64), lds, s, (const bf16_t*)q,                 \
                     (const bf16_t*)k, (const bf16_t*)v, k_tok_stride, v_tok_stride, cu_seqlens_q, cu_seqlens_k,        \
                     block_tables, bt_stride, (bf16_t*)out, num_seqs, num_q_heads, num_kv_heads, block_size, sl2, xcd_map, lse, thr)
  if (paged && kv_dtype == NVL_KV_FP8) {
    if (eight) NVL_PF_LAUNCH(true, true, 8); else NVL_PF_LAUNCH(true, true, 4);
  } else if (paged) {
    if (eight) NVL_PF_LAUNCH(true, false, 8); else NVL_PF_LAUNCH(true, false, 4);
  } else {
    if (eight) NVL_PF_LAUNCH(false, false, 8); else NVL_PF_LAUNCH(false, false, 4);
  }
#undef NVL_PF_LAUNCH
  return nvl_check_launch("nvl_attn_prefill_varlen");
}
