// Paged single-query (decode) attention for gfx950 — the HBM-roofline kernel of the path.
// Replaces flash_attn_with_kvcache as used by nano-vllm layers/attention.py:72-74.
//
// Work decomposition (persistent, hipGraph-safe, no host work list): every workgroup prefix-sums
// ceil(len_b / 32) over the batch into LDS, which flattens the step's work into one sequence of
// (sequence b, kv-head h, 32-token tile) units. Every WAVE then takes an equal contiguous share
// of that sequence ("stream-K" over tokens): no workgroup barriers or LDS merge in the loop, and
// load balance to within one 16 KiB tile however ragged the context lengths are. A wave's share
// may cover the tail of one (b, h), several whole short ones and the head of another; it emits
// one split partial (m, l, O) per (b, h) segment into slot k = wave - first_wave(b, h), and a
// second tiny kernel merges the slots. The grid is a launch-time constant, so the same captured
// launch serves any mix of lengths; padded rows (context_len 0) contribute no tiles.
//
// Three kernels share this bookkeeping, the split-partial format and the merge kernel:
//   decode_stream_kernel      packed-dot (VALU) scores straight from registers — group size 1 (and 2 / 4 / 8 with
//                             NVL_DECODE_MFMA=0 / NVL_DECODE_G8_VALU=1); described in the next paragraphs
//   decode_mfma8_kernel       scores and P.V on the matrix cores through a per-wave LDS tile, K/V loads one tile
//                             ahead of the matrix work — group sizes 2, 4, 8, bf16 or fp8 cache (the default there)
//   decode_stream_fp8_kernel  fp8 cache on the packed-dot path — group size 1 (and 2 / 4 with NVL_DECODE_MFMA=0)
//
// Data path (packed-dot kernel): K and V tiles go HBM -> VGPR directly with non-temporal loads (each byte is used
// exactly once; an LDS round trip would be pure overhead, and `nt` measured +8 % bandwidth). One
// wave instruction fetches 4 token rows x 256 B = 1 KiB contiguous (head-major cache layout),
// 16 instructions (16 KiB) are in flight per wave before the first use. Lane (rq = lane>>4,
// sub = lane&15) holds elements sub*8..sub*8+7 of rows i*4+rq. q.K partial dot products use
// packed v_dot2c_f32_bf16 and are summed over the 16 lanes of a DPP row (row_ror); softmax
// max/sum use wave shuffles; P is rounded to bf16 before P.V (flash-attn convention), fp32
// accumulation. All G = Hq/Hkv query heads of a group are served from one K/V read.
//
// FUSED variant (nvl_paged_attn_decode_fused): the decode step's q/k-RMSNorm -> RoPE -> KV-cache
// store (models/qwen3.py:82-85 + layers/attention.py:63, a separate ~5 us launch per layer) is
// folded in. Every wave norms+rotates the q heads it needs straight from the qkv GEMM output
// (position = context_len - 1); the ONE wave that owns the last tile of a (sequence, kv-head)
// norms+rotates the new token's k, writes k and v into their cache slot (derived from the block
// table) and substitutes them for that row of its register tile — the row is never read back
// from HBM inside this launch, so there is no intra-launch dependency to order.
#include "common.h"
#include <stdlib.h>
#include <mutex>
#include "rope_common.h"

namespace {

constexpr int kWaves = 4;
constexpr int kTile = 32;                 // tokens per wave per step
constexpr int kLoads = kTile / 4;         // 16-byte loads per lane per tile (K or V)
constexpr float kNegBig = -1.0e30f;

__device__ __forceinline__ float dot8(const u32x4_t& a, const u32x4_t& b) {
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // NB: bit_cast the scalar copies, not the vector-element expressions a[i]/b[i]: hipcc
    // (ROCm 7.2) folds `__builtin_bit_cast(T, vec[i])` to element 0 for every i.
    const unsigned int ai = a[i], bi = b[i];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ai), __builtin_bit_cast(bf16x2_t, bi), acc,
                                          false);
  }
  return acc;
}

// Inclusive prefix of ceil(len/chunk) into pre[1..B], pre[0] = 0. All 256 threads participate.
// `skip`: leading tiles that are NOT part of the share of a sequence with member[i] != 0 (the shared-prefix pass takes them).
__device__ __forceinline__ void chunk_prefix(const int32_t* __restrict__ ctx, int batch, int chunk, int* pre,
                                             int* wsum, int skip = 0, const int32_t* __restrict__ member = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;
  if (threadIdx.x == 0) pre[0] = 0;
  for (int base = 0; base < batch; base += 256) {
    const int i = base + threadIdx.x;
    int v = 0;
    if (i < batch) {
      const int len = ctx[i];
      const int mine = (skip > 0 && member[i] != 0) ? skip : 0;
      v = len > 0 ? max((len + chunk - 1) / chunk - mine, 0) : 0;
    }
    int s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(s, o, 64);
      if (lane >= o) s += n;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const int t = wsum[w];
      if (w < wave) woff += t;
      tot += t;
    }
    if (i < batch) pre[i + 1] = carry + woff + s;
    carry += tot;
    __syncthreads();
  }
}

// A wave never takes fewer than this many tiles, which bounds the number of split partials per
// (b, h) by max_context / (32 * 4) + 2 — the workspace size is therefore static.
constexpr int kMinTilesPerWave = 4;

// ---------------------------------------------------------------------------------------------------
// Per-STEP plan (nvl_decode_plan): the flattening above depends only on (context_lens, Hkv, grid), which are the same
// for every layer of a decode step, yet every workgroup of every layer's launch recomputed it — a coalesced read of
// context_lens, a 256-thread prefix scan with two barriers, and a binary search per wave, ~2 us between launch and the
// first K/V byte requested, 28 times per step on Qwen3-0.6B. One tiny kernel per step now writes, for every wave of
// the attention grid, where its share starts; the attention kernel reads ONE 16-byte record through the scalar cache
// and goes. Later segments of a wave's share need no search at all: they always start at tile 0 of the next
// (sequence, kv-head) pair, and the tile count of a sequence is ceil(context_len / 32).
struct PlanHeader {          // 48 bytes, followed by nwaves PlanEntry records
  int64_t total;             // tiles of the whole step = hkv * sum_b (ceil(len_b / 32) - [b is a member] sh_tiles)
  int64_t per;               // tiles per wave
  int32_t nwaves, batch, hkv;
  int32_t sh_tiles;          // leading tiles of every MEMBER sequence that the shared-prefix pass computes (0: none)
  const int32_t* member;     // [batch] != 0: the sequence starts with the shared blocks (the caller's array; sh_tiles > 0 only)
  int32_t px_groups;         // group slots of the shared-prefix pass (0: a plan without one)
  int32_t pad;
};
static_assert(sizeof(PlanHeader) % 16 == 0, "the wave records behind the header are 16-byte aligned");
struct __attribute__((aligned(16))) PlanEntry { int32_t b, h, t0, nb; };   // first segment of a wave's share (b < 0: none)

__global__ __launch_bounds__(256) void decode_plan_kernel(const int32_t* __restrict__ ctx, int batch, int hkv, int nwaves,
                                                          PlanHeader* __restrict__ hdr,
                                                          const int32_t* __restrict__ shared_blocks, int tiles_per_block,
                                                          int px_groups) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* wsum = reinterpret_cast<int*>(smem_raw);
  int* pre = wsum + kWaves;
  // Shared prefix: shared_blocks[0] = number of leading KV blocks the MEMBER sequences (shared_blocks[1 + b] != 0) have in
  // common. Whatever it says, the tile that takes a member's NEW token stays in that sequence's own share (the stream-K
  // kernel stores the token and starts its softmax there): sh <= min over live members of floor((len - 1) / 32).
  int sh = 0;
  const int32_t* member = shared_blocks != nullptr ? shared_blocks + 1 : nullptr;
  if (shared_blocks != nullptr) {
    int* smin = pre + batch + 1;                        // (one more int of the dynamic region: the launcher sizes it)
    if (threadIdx.x == 0) *smin = 0x7fffffff;
    __syncthreads();
    int lmin = 0x7fffffff;
    for (int i = threadIdx.x; i < batch; i += 256) {
      const int len = ctx[i];
      if (len > 0 && member[i] != 0) lmin = min(lmin, (len - 1) / kTile);
    }
    atomicMin(smin, lmin);
    __syncthreads();
    const int want = shared_blocks[0];
    const int lowest = *smin;
    sh = want > 0 ? min(want * tiles_per_block, lowest) : 0;
    if (lowest == 0x7fffffff) sh = 0;                   // no live member
    __syncthreads();
  }
  chunk_prefix(ctx, batch, kTile, pre, wsum, sh, member);
  __syncthreads();
  const int64_t total = (int64_t)pre[batch] * hkv;
  int64_t per = (total + nwaves - 1) / nwaves;
  if (per < kMinTilesPerWave) per = kMinTilesPerWave;
  if (threadIdx.x == 0) {
    hdr->total = total;
    hdr->per = per;
    hdr->nwaves = nwaves;
    hdr->batch = batch;
    hdr->hkv = hkv;
    hdr->sh_tiles = sh;
    hdr->member = member;
    hdr->px_groups = shared_blocks != nullptr ? px_groups : 0;
    hdr->pad = 0;
  }
  PlanEntry* ent = reinterpret_cast<PlanEntry*>(hdr + 1);
  for (int w = threadIdx.x; w < nwaves; w += 256) {
    const int64_t g = (int64_t)w * per;
    PlanEntry e = {-1, 0, 0, 0};
    if (g < total) {
      int lo = 0, hi = batch;  // largest b with hkv * pre[b] <= g
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)pre[mid] * hkv <= g) lo = mid; else hi = mid;
      }
      const int nb = pre[lo + 1] - pre[lo];
      const int r = (int)(g - (int64_t)pre[lo] * hkv);
      e.b = lo;
      e.nb = nb;
      e.h = r / nb;
      e.t0 = r - e.h * nb;
    }
    ent[w] = e;
  }
}

// streamed-once data: non-temporal load (does not displace q / block tables / partials in L2)
__device__ __forceinline__ u32x4_t load16_nt(const bf16_t* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}

struct FusedArgs {              // only read by the FUSED instantiation
  int64_t qkv_tok_stride;
  const bf16_t* q_norm_w;
  const bf16_t* k_norm_w;
  const float* cos_sin;
  int64_t max_pos;
  float eps;
  // qkv_splits > 0 (matrix-core kernel only): `qkv` is not the bf16 GEMM output but the fp32 split-K slabs
  // [splits][batch][qkv_tok_stride] of nvl_linear_wide mode 2 (slab s starts qkv_split_stride elements after slab s - 1):
  // a row piece is summed over the slabs in slab order and rounded to bf16 once — exactly what slab_reduce_kernel would
  // have written — before the norm / rotation. The separate reduce launch of the qkv projection disappears.
  int qkv_splits;
  int qkv_split_stride;         // elements (< 2^31: checked by the entry point)
};

// 8 consecutive elements of a qkv row: bf16 as stored, or the rounded sum of the fp32 split-K slabs. The slab pieces are
// requested FOUR SLABS AT A TIME before the first add (clamped index, predicated add): a runtime loop of load -> add would
// pay one dependent L2 round trip per slab on the critical path of every segment prologue (measured: +9 us per launch on
// the one-kv-head shape with 5 slabs). Sum order is slab 0, 1, 2, ... — slab_reduce_kernel's.
template <bool SLABS>
__device__ __forceinline__ u32x4_t load_qkv8(const bf16_t* base, int64_t elt, const FusedArgs& fa) {
  if constexpr (!SLABS) {
    return *reinterpret_cast<const u32x4_t*>(base + elt);
  } else {
    const float* p = reinterpret_cast<const float*>(base) + elt;
    const int S = fa.qkv_splits;
    // (stride and slab index live in VECTOR registers: the kernel sits at its scalar-register limit, and wave-uniform
    //  64-bit slab addresses would spill SGPRs)
    int stride_v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(stride_v) : "s"(fa.qkv_split_stride));
    f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 4) {
      f32x4_t ta[4], tb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int off = (s0 + j < S ? s0 + j : S - 1) * stride_v;
        ta[j] = *reinterpret_cast<const f32x4_t*>(p + off);
        tb[j] = *reinterpret_cast<const f32x4_t*>(p + off + 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (s0 + j < S) {
          a = s0 + j == 0 ? ta[j] : a + ta[j];                    // (slab 0 starts the sum: 0 + x would turn -0.0 into +0.0)
          b = s0 + j == 0 ? tb[j] : b + tb[j];
        }
    }
    return u32x4_t{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  }
}

template <int G, bool FUSED>
__global__ __launch_bounds__(256, G == 4 ? 2 : 1) void decode_stream_kernel(   // G = 4: stay within 256 VGPRs
    const bf16_t* __restrict__ q, bf16_t* kc, bf16_t* vc,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, const int32_t* __restrict__ ctx,
    float* __restrict__ part_o, float* __restrict__ part_ml, int* __restrict__ meta, int batch, int hkv,
    int block_size, int slots, float scale_log2e, FusedArgs fa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* wsum = reinterpret_cast<int*>(smem_raw);
  int* pre = wsum + kWaves;  // tile prefix [batch + 1]

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane & 15, rq = lane >> 4;
  const int hq = hkv * G;

  chunk_prefix(ctx, batch, kTile, pre, wsum);
  __syncthreads();
  const int64_t total = (int64_t)pre[batch] * hkv;
  const int64_t nwaves = (int64_t)gridDim.x * kWaves;
  int64_t per = (total + nwaves - 1) / nwaves;
  if (per < kMinTilesPerWave) per = kMinTilesPerWave;
  const int64_t wid = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t g1 = min(total, (wid + 1) * per);

  for (int64_t g = wid * per; g < g1;) {
    int lo = 0, hi = batch;  // largest b with hkv * pre[b] <= g
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)pre[mid] * hkv <= g) lo = mid; else hi = mid;
    }
    // wave-uniform by construction (every lane ran the same search on the same LDS words): say so, so the
    // segment bookkeeping, block-table loads and tile base addresses live in scalar registers
    const int b = __builtin_amdgcn_readfirstlane(lo);
    const int nb = __builtin_amdgcn_readfirstlane(pre[b + 1] - pre[b]);   // tiles per kv-head of this sequence (> 0 here)
    const int64_t base_b = (int64_t)__builtin_amdgcn_readfirstlane(pre[b]) * hkv;
    const int r = (int)(g - base_b);
    const int h = r / nb;
    const int t0 = r - h * nb;
    const int run = (int)min((int64_t)(nb - t0), g1 - g);
    const int len = ctx[b];

    // Segment prologue: q rows (FUSED: plus norm weights, the cos/sin row, the new token's raw k / v).
    u32x4_t qf[G], rawq[G];
    u32x4_t knew = {0u, 0u, 0u, 0u}, vnew = {0u, 0u, 0u, 0u}, wq = {0u, 0u, 0u, 0u}, wk = {0u, 0u, 0u, 0u};
    RopeRegs rr = {};
    const bool owns_last = FUSED && (t0 + run == nb);             // this wave processes the tile of token len-1
    if constexpr (FUSED) {
      // q is the raw qkv GEMM row [q heads | k heads | v heads]; the new token sits at position len-1
      int64_t pos = len - 1;
      pos = pos >= fa.max_pos ? fa.max_pos - 1 : pos;
      const bf16_t* row = q + (int64_t)b * fa.qkv_tok_stride;
#pragma unroll
      for (int gg = 0; gg < G; ++gg) rawq[gg] = *reinterpret_cast<const u32x4_t*>(row + (h * G + gg) * 128 + sub * 8);
      rr = load_rope_regs(fa.cos_sin + pos * 128, sub);
      if (fa.q_norm_w != nullptr) {
        wq = *reinterpret_cast<const u32x4_t*>(fa.q_norm_w + sub * 8);
        wk = *reinterpret_cast<const u32x4_t*>(fa.k_norm_w + sub * 8);
      }
      if (owns_last) {
        knew = *reinterpret_cast<const u32x4_t*>(row + (hq + h) * 128 + sub * 8);
        vnew = *reinterpret_cast<const u32x4_t*>(row + (hq + hkv + h) * 128 + sub * 8);
      }
    } else {
#pragma unroll
      for (int gg = 0; gg < G; ++gg)
        rawq[gg] = *reinterpret_cast<const u32x4_t*>(q + ((int64_t)b * hq + h * G + gg) * 128 + sub * 8);
    }
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      m[gg] = kNegBig;
      l[gg] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[gg][j] = 0.f;
    }

    u32x4_t kd[kLoads], vd[kLoads];
    auto issue_tile = [&](int ti) {
      const int t = ti * kTile;
      const int blk = block_tables[(int64_t)b * bt_stride + t / block_size];
      const int64_t base = (((int64_t)blk * hkv + h) * block_size + (t % block_size)) * 128 + lane * 8;
      const bf16_t* kp = kc + base;
      const bf16_t* vp = vc + base;
#pragma unroll
      for (int i = 0; i < kLoads; ++i) kd[i] = load16_nt(kp + i * 4 * 128);
#pragma unroll
      for (int i = 0; i < kLoads; ++i) vd[i] = load16_nt(vp + i * 4 * 128);
    };
    if constexpr (FUSED) {
#pragma unroll
      for (int gg = 0; gg < G; ++gg)
        qf[gg] = norm_rope_head_regs(rawq[gg], fa.q_norm_w != nullptr, wq, fa.eps, rr, sub);
      if (owns_last) {
        knew = norm_rope_head_regs(knew, fa.k_norm_w != nullptr, wk, fa.eps, rr, sub);
        const int tl = len - 1;
        if (rq == 0) {                                             // one 256-byte row each, for later steps
          const int blk = block_tables[(int64_t)b * bt_stride + tl / block_size];
          const int64_t dst = (((int64_t)blk * hkv + h) * block_size + (tl % block_size)) * 128 + sub * 8;
          *reinterpret_cast<u32x4_t*>(kc + dst) = knew;
          *reinterpret_cast<u32x4_t*>(vc + dst) = vnew;
        }
        // The new token enters the online softmax HERE, as the first key of this wave's segment
        // (m = its score, l = 1, O = v; P = bf16(2^0) = 1): its cache row is written above but never
        // read back inside this launch — the tile loop masks row len-1.
        float vf[8];
        unpack8(vnew, vf);
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
          m[gg] = row16_allreduce_sum(dot8(knew, qf[gg])) * scale_log2e;
          l[gg] = rq == 0 ? 1.f : 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[gg][j] = rq == 0 ? vf[j] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int gg = 0; gg < G; ++gg) qf[gg] = rawq[gg];
    }
    const int len_cached = FUSED ? len - 1 : len;    // rows of the cache that hold valid K/V for this launch

    for (int ti = t0; ti < t0 + run; ++ti) {
      const int t = ti * kTile;
      issue_tile(ti);
      __builtin_amdgcn_sched_barrier(0);  // all 16 loads in flight before the first use
      float s[G][kLoads];
#pragma unroll
      for (int i = 0; i < kLoads; ++i) {
        const bool valid = (t + i * 4 + rq) < len_cached;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
          const float d = row16_allreduce_sum(dot8(kd[i], qf[gg]));
          s[gg][i] = valid ? d * scale_log2e : kNegBig;
        }
      }
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        float mx = s[gg][0];
#pragma unroll
        for (int i = 1; i < kLoads; ++i) mx = fmaxf(mx, s[gg][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m[gg], mx);
        const float alpha = exp2f(m[gg] - mn);
        m[gg] = mn;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
          const float p = exp2f(s[gg][i] - mn);
          psum += p;
          s[gg][i] = round_bf16(p);
        }
        l[gg] = l[gg] * alpha + psum;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[gg][j] *= alpha;
      }
#pragma unroll
      for (int i = 0; i < kLoads; ++i) {
        float vf[8];
        unpack8(vd[i], vf);
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
          for (int j = 0; j < 8; ++j) o[gg][j] = fmaf(s[gg][i], vf[j], o[gg][j]);
      }
    }

    // fold the wave's 4 row-groups and emit the partial of this (b, h) segment
    const int64_t seg0 = base_b + (int64_t)h * nb;                 // first global tile of (b, h)
    const int first = (int)(seg0 / per);
    const int k = (int)(wid - first);
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      l[gg] += __shfl_xor(l[gg], 16, 64);
      l[gg] += __shfl_xor(l[gg], 32, 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[gg][j] += __shfl_xor(o[gg][j], 16, 64);
        o[gg][j] += __shfl_xor(o[gg][j], 32, 64);
      }
    }
    if (rq == 0) {
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const int64_t pidx = ((int64_t)b * hq + h * G + gg) * slots + k;
        float* dst = part_o + pidx * 128 + sub * 8;
        *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{o[gg][0], o[gg][1], o[gg][2], o[gg][3]};
        *reinterpret_cast<f32x4_t*>(dst + 4) = f32x4_t{o[gg][4], o[gg][5], o[gg][6], o[gg][7]};
        if (sub == 0) {
          part_ml[pidx * 2] = m[gg];
          part_ml[pidx * 2 + 1] = l[gg];
        }
      }
      if (sub == 0) meta[b * hkv + h] = (int)((seg0 + nb - 1) / per) - first + 1;   // #partials of (b, h)
    }
    g += run;
  }
}

// ---------------------------------------------------------------------------------------------------
// FP8 (OCP e4m3) KV-cache variant of decode_stream_kernel (opt-in, SURVEY.md §8f-4: decode is 92 % K/V bytes on the
// headline workload, so halving the bytes per token is the only lever that moves the roofline IDEAL; it changes
// numerics, so it stays outside the parity runs). Same stream-K bookkeeping, split-partial format and combine
// kernel. What differs: a cache row is 128 BYTES, so one 16-byte load holds 16 elements and 8 lanes (half a DPP
// row) cover a token: lane (r8 = lane >> 3, s8 = lane & 7) owns dims 16 s8 .. 16 s8 + 15 of rows 8 i + r8, and one
// wave instruction still fetches 1 KiB contiguous (8 token rows). fp8 -> fp32 is one v_cvt_pk_f32_fp8 per two
// elements (exact), the dot products are fp32 FMAs against the pre-unpacked q, reduced over the 8 lanes with three
// DPP adds. FUSED: q/k are normed + rotated in the bf16 kernel's 16-lane layout and re-distributed once per
// segment; the new token's k / v are QUANTISED before they enter this step's softmax, so a step sees exactly the
// values every later step will read back from the cache.
constexpr int kLoads8 = kTile / 8;       // 16-byte loads per lane per tile (K or V), 8 token rows each

template <int G, bool FUSED>
__global__ __launch_bounds__(256, G == 4 ? 2 : 1) void decode_stream_fp8_kernel(
    const bf16_t* __restrict__ q, unsigned char* kc, unsigned char* vc, const int32_t* __restrict__ block_tables,
    int64_t bt_stride, const int32_t* __restrict__ ctx, float* __restrict__ part_o, float* __restrict__ part_ml,
    int* __restrict__ meta, int batch, int hkv, int block_size, int slots, float scale_log2e, FusedArgs fa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* wsum = reinterpret_cast<int*>(smem_raw);
  int* pre = wsum + kWaves;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane & 15, rq = lane >> 4;      // prologue view (16 lanes x 8 dims = one head row)
  const int s8 = lane & 7, r8 = lane >> 3;        // tile view (8 lanes x 16 dims = one cache row)
  const int hq = hkv * G;
  // source lanes of the 16-lane layout holding this lane's two 8-element chunks (same 16-lane row group)
  const int src_lo = (lane & 48) | (s8 * 2), src_hi = src_lo | 1;

  chunk_prefix(ctx, batch, kTile, pre, wsum);
  __syncthreads();
  const int64_t total = (int64_t)pre[batch] * hkv;
  const int64_t nwaves = (int64_t)gridDim.x * kWaves;
  int64_t per = (total + nwaves - 1) / nwaves;
  if (per < kMinTilesPerWave) per = kMinTilesPerWave;
  const int64_t wid = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t g1 = min(total, (wid + 1) * per);

  for (int64_t g = wid * per; g < g1;) {
    int lo = 0, hi = batch;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)pre[mid] * hkv <= g) lo = mid; else hi = mid;
    }
    const int b = __builtin_amdgcn_readfirstlane(lo);
    const int nb = __builtin_amdgcn_readfirstlane(pre[b + 1] - pre[b]);
    const int64_t base_b = (int64_t)__builtin_amdgcn_readfirstlane(pre[b]) * hkv;
    const int r = (int)(g - base_b);
    const int h = r / nb;
    const int t0 = r - h * nb;
    const int run = (int)min((int64_t)(nb - t0), g1 - g);
    const int len = ctx[b];
    const bool owns_last = FUSED && (t0 + run == nb);

    // ---- segment prologue in the 16-lane layout -------------------------------------------------------------
    u32x4_t qf[G];
    u32x4_t knew8 = {0u, 0u, 0u, 0u}, vnew8 = {0u, 0u, 0u, 0u};       // new token's row, 8-lane layout, fp8
    if constexpr (FUSED) {
      int64_t pos = len - 1;
      pos = pos >= fa.max_pos ? fa.max_pos - 1 : pos;
      const bf16_t* row = q + (int64_t)b * fa.qkv_tok_stride;
      const RopeRegs rr = load_rope_regs(fa.cos_sin + pos * 128, sub);
      u32x4_t wq = {0u, 0u, 0u, 0u}, wk = {0u, 0u, 0u, 0u};
      if (fa.q_norm_w != nullptr) {
        wq = *reinterpret_cast<const u32x4_t*>(fa.q_norm_w + sub * 8);
        wk = *reinterpret_cast<const u32x4_t*>(fa.k_norm_w + sub * 8);
      }
#pragma unroll
      for (int gg = 0; gg < G; ++gg)
        qf[gg] = norm_rope_head_regs(*reinterpret_cast<const u32x4_t*>(row + (h * G + gg) * 128 + sub * 8),
                                     fa.q_norm_w != nullptr, wq, fa.eps, rr, sub);
      if (owns_last) {
        u32x4_t knew = *reinterpret_cast<const u32x4_t*>(row + (hq + h) * 128 + sub * 8);
        const u32x4_t vnew = *reinterpret_cast<const u32x4_t*>(row + (hq + hkv + h) * 128 + sub * 8);
        knew = norm_rope_head_regs(knew, fa.k_norm_w != nullptr, wk, fa.eps, rr, sub);
        const u32x2_t kq = bf16x8_to_fp8x8(knew), vq = bf16x8_to_fp8x8(vnew);
        const int tl = len - 1;
        if (rq == 0) {                                             // one 128-byte row each, for later steps
          const int blk = block_tables[(int64_t)b * bt_stride + tl / block_size];
          const int64_t dst = (((int64_t)blk * hkv + h) * block_size + (tl % block_size)) * 128 + sub * 8;
          *reinterpret_cast<u32x2_t*>(kc + dst) = kq;
          *reinterpret_cast<u32x2_t*>(vc + dst) = vq;
        }
        // quantised row -> 8-lane layout (16 bytes per lane)
        knew8 = u32x4_t{(unsigned)__shfl((int)kq[0], src_lo, 64), (unsigned)__shfl((int)kq[1], src_lo, 64),
                        (unsigned)__shfl((int)kq[0], src_hi, 64), (unsigned)__shfl((int)kq[1], src_hi, 64)};
        vnew8 = u32x4_t{(unsigned)__shfl((int)vq[0], src_lo, 64), (unsigned)__shfl((int)vq[1], src_lo, 64),
                        (unsigned)__shfl((int)vq[0], src_hi, 64), (unsigned)__shfl((int)vq[1], src_hi, 64)};
      }
    } else {
#pragma unroll
      for (int gg = 0; gg < G; ++gg)
        qf[gg] = *reinterpret_cast<const u32x4_t*>(q + ((int64_t)b * hq + h * G + gg) * 128 + sub * 8);
    }
    // q -> 8-lane layout, unpacked to fp32: qv[gg][j] = q[16 s8 + j]
    float qv[G][16];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      u32x4_t a, c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = (unsigned)__shfl((int)qf[gg][i], src_lo, 64);
        c[i] = (unsigned)__shfl((int)qf[gg][i], src_hi, 64);
      }
      unpack8(a, qv[gg]);
      unpack8(c, qv[gg] + 8);
    }

    float m[G], l[G], o[G][16];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      m[gg] = kNegBig;
      l[gg] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) o[gg][j] = 0.f;
    }

    u32x4_t kd[kLoads8], vd[kLoads8];
    // one tile step on the rows held in kd / vd: row i * 8 + r8 of the tile is valid iff valid_row(i)
    auto consume = [&](auto valid_row) {
      float s[G][kLoads8];
#pragma unroll
      for (int i = 0; i < kLoads8; ++i) {
        float kf[16];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) unpack_fp8x4(kd[i][w4], kf + 4 * w4);
        const bool valid = valid_row(i);
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
          float d = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) d = fmaf(kf[j], qv[gg][j], d);
          d = half8_allreduce_sum(d);
          s[gg][i] = valid ? d * scale_log2e : kNegBig;
        }
      }
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        float mx = s[gg][0];
#pragma unroll
        for (int i = 1; i < kLoads8; ++i) mx = fmaxf(mx, s[gg][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m[gg], mx);
        const float alpha = exp2f(m[gg] - mn);
        m[gg] = mn;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < kLoads8; ++i) {
          const float p = exp2f(s[gg][i] - mn);
          psum += p;
          s[gg][i] = round_bf16(p);
        }
        l[gg] = l[gg] * alpha + psum;
#pragma unroll
        for (int j = 0; j < 16; ++j) o[gg][j] *= alpha;
      }
#pragma unroll
      for (int i = 0; i < kLoads8; ++i) {
        float vf[16];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) unpack_fp8x4(vd[i][w4], vf + 4 * w4);
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
          for (int j = 0; j < 16; ++j) o[gg][j] = fmaf(s[gg][i], vf[j], o[gg][j]);
      }
    };

    if constexpr (FUSED) {
      if (owns_last) {
        // the new token is the first key of this wave's segment: a one-row tile (row 0 of row-group 0); its cache
        // row is written above but never read back inside this launch — the tile loop masks row len - 1
#pragma unroll
        for (int i = 0; i < kLoads8; ++i) {
          kd[i] = u32x4_t{0u, 0u, 0u, 0u};
          vd[i] = u32x4_t{0u, 0u, 0u, 0u};
        }
        kd[0] = knew8;
        vd[0] = vnew8;
        consume([&](int i) { return i == 0 && r8 == 0; });
      }
    }
    const int len_cached = FUSED ? len - 1 : len;

    // (Loading TWO tiles — 16 KiB per wave, what the bf16 kernel keeps in flight — before consuming them was
    // measured and dropped: 59.3 vs 57.1 us per launch on the bench schedule, profiles/r02_bench_fp8kv_pair_tiles_rejected.json;
    // at 8 bytes of conversion + FMA work per loaded byte this kernel is no longer waiting on the loads.)
    auto tile_base = [&](int ti) {
      const int t = ti * kTile;
      const int blk = block_tables[(int64_t)b * bt_stride + t / block_size];
      return (((int64_t)blk * hkv + h) * block_size + (t % block_size)) * 128 + lane * 16;
    };
    int ti = t0;
    for (; ti < t0 + run; ++ti) {
      const int t = ti * kTile;
      const int64_t base = tile_base(ti);
#pragma unroll
      for (int i = 0; i < kLoads8; ++i)
        kd[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kc + base + i * 8 * 128));
#pragma unroll
      for (int i = 0; i < kLoads8; ++i)
        vd[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vc + base + i * 8 * 128));
      __builtin_amdgcn_sched_barrier(0);  // all 8 loads in flight before the first use
      consume([&](int i) { return (t + i * 8 + r8) < len_cached; });
    }

    // fold the wave's 8 row-groups and emit the partial of this (b, h) segment
    const int64_t seg0 = base_b + (int64_t)h * nb;
    const int first = (int)(seg0 / per);
    const int k = (int)(wid - first);
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      l[gg] += __shfl_xor(l[gg], 8, 64);
      l[gg] += __shfl_xor(l[gg], 16, 64);
      l[gg] += __shfl_xor(l[gg], 32, 64);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        o[gg][j] += __shfl_xor(o[gg][j], 8, 64);
        o[gg][j] += __shfl_xor(o[gg][j], 16, 64);
        o[gg][j] += __shfl_xor(o[gg][j], 32, 64);
      }
    }
    if (r8 == 0) {
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const int64_t pidx = ((int64_t)b * hq + h * G + gg) * slots + k;
        float* dst = part_o + pidx * 128 + s8 * 16;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
          *reinterpret_cast<f32x4_t*>(dst + 4 * j4) =
              f32x4_t{o[gg][4 * j4], o[gg][4 * j4 + 1], o[gg][4 * j4 + 2], o[gg][4 * j4 + 3]};
        if (s8 == 0) {
          part_ml[pidx * 2] = m[gg];
          part_ml[pidx * 2 + 1] = l[gg];
        }
      }
      if (s8 == 0) meta[b * hkv + h] = (int)((seg0 + nb - 1) / per) - first + 1;
    }
    g += run;
  }
}

// ---------------------------------------------------------------------------------------------------
// G = 8 (Qwen3-32B per-rank shapes at TP 4 / 8: one K/V row serves 8 query heads, 8 FLOP/B): the packed
// v_dot2 path above is VALU-bound there (2.4 TB/s), so the scores and the P.V product go to the matrix
// cores. Same stream-K bookkeeping, same split-partial format, same HBM->VGPR non-temporal tile loads;
// a wave then lays its 16 KiB tile out in a PRIVATE LDS region (no workgroup barriers) and reads it back
// as v_mfma_f32_16x16x32_bf16 operands:
//   S^T[16 tokens x 16 heads] = K[16 x 128] . Q^T[128 x 16]      A = K rows (ds_read_b128, XOR-swizzled
//                                                                 16-byte slots), B = Q^T (registers,
//                                                                 heads 8..15 are zero padding)
//   O^T[16 dims  x 16 heads] += V^T[16 x 32] . P^T[32 x 16]      A = V^T via ds_read_b64_tr_b16 (hardware
//                                                                 transpose of the row-major V tile),
//                                                                 B = P^T straight from the S^T
//                                                                 accumulators: the MFMA k-slot <-> token
//                                                                 map is chosen so P never changes lane
// Lane (head = lane & 15, quad = lane >> 4) owns one head column: the online softmax is lane-local plus
// two shuffles, the O rescale is a per-lane scalar. LDS traffic is 2 x 16 KiB per tile per wave, far
// below what the CU sustains at HBM-bound rates (~10 B/clk/CU).
constexpr int kMKRow = 256;             // K tile row bytes in LDS (32 rows, swizzled slots)
constexpr int kMVRow = 288;             // V tile row bytes (256 + 32: conflict-free transpose reads)
constexpr int kMWaveLds = kTile * (kMKRow + kMVRow);   // 17,408 B per wave

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

// KV8: the cache holds OCP fp8 e4m3 rows of 128 bytes (opt-in, see decode_stream_fp8_kernel): a tile is loaded as
// 8 rows x 16 elements per wave instruction and converted (exactly) to bf16 on its way into the wave's LDS tile; the
// new token's k / v are quantised before they enter this step's softmax. Everything after the LDS write is unchanged.
// (the body is a device function of (workgroup index, workgroups of the stream-K grid): the plain kernel runs it on every
//  workgroup, decode_mfma8_shared_kernel on the first `main_blocks` of a launch whose remaining workgroups serve the
//  shared-prefix packs)
template <bool FUSED, bool KV8, int G, bool SLABS>
__device__ __forceinline__ void mfma8_body(
    const bf16_t* __restrict__ q, bf16_t* kc, bf16_t* vc, const int32_t* __restrict__ block_tables,
    int64_t bt_stride, const int32_t* __restrict__ ctx, float* __restrict__ part_o, float* __restrict__ part_ml,
    int* __restrict__ meta, bf16_t* __restrict__ out, int batch, int hkv, int block_size, int slots,
    float scale_log2e, const FusedArgs& fa, const PlanHeader* __restrict__ plan, int g_rt, const int block, const int nblocks) {
  // G = 0: the group size is the runtime argument g_rt (1 ... 16; Qwen3-14B is 40 / 8 = 5) — one instantiation serves every
  // group size without a tuned one; the heads still fill ONE 16-column MFMA tile, padded with zero columns
  static_assert(G >= 0 && G <= 16, "one 16-column MFMA tile holds the heads of a kv group (padded with zero columns)");
  int Gv = G;
  if constexpr (G == 0) {
    // (kept in a VECTOR register: the kernel sits at its scalar-register limit, and every use of the group size — head
    //  offsets, column masks — is per-lane arithmetic anyway)
    asm volatile("v_mov_b32 %0, %1" : "=v"(Gv) : "s"(g_rt));
  }
  constexpr int NIT = (G > 0 && G <= 8) ? 2 : 4;          // prologue passes: 4 head rows per pass
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* wsum = reinterpret_cast<int*>(smem_raw + kWaves * kMWaveLds);
  int* pre = wsum + kWaves;  // tile prefix [batch + 1]

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane & 15, rq = lane >> 4;      // load / prologue view: 16 lanes x 8 dims = one row
  const int head = lane & 15, quad = lane >> 4;   // MFMA view: one head column per lane
  const int hq = hkv * Gv;
  unsigned char* k_lds = smem_raw + wave * kMWaveLds;
  unsigned char* v_lds = k_lds + kTile * kMKRow;

  const int64_t wid = (int64_t)block * kWaves + wave;
  int64_t total, per;
  int sh = 0;                // leading tiles of every MEMBER sequence that belong to the shared-prefix pass (plan only)
  const int32_t* member = nullptr;
  PlanEntry first_seg = {-1, 0, 0, 0};
  if (plan != nullptr) {
    // per-step plan: header + this wave's record, two scalar loads (the addresses are wave-uniform); no LDS prefix,
    // no barrier, no search
    // A plan is only valid for the (batch, Hkv, grid) it was built for. The host side refuses a mismatch
    // (plan_shadow_check); should one reach the device anyway — a replayed graph whose plan buffer was overwritten by a
    // differently shaped nvl_decode_plan — the launch does NO work instead of indexing past the records.
    const bool plan_ok = plan->nwaves == (int)(nblocks * kWaves) && plan->batch == batch && plan->hkv == hkv;
    total = plan_ok ? plan->total : 0;
    per = plan->per;
    sh = plan->sh_tiles;
    member = plan->member;
    if (plan_ok) first_seg = reinterpret_cast<const PlanEntry*>(plan + 1)[wid];
  } else {
    chunk_prefix(ctx, batch, kTile, pre, wsum);
    __syncthreads();
    total = (int64_t)pre[batch] * hkv;
    const int64_t nwaves = (int64_t)nblocks * kWaves;
    per = (total + nwaves - 1) / nwaves;
    if (per < kMinTilesPerWave) per = kMinTilesPerWave;
  }
  const int64_t g1 = min(total, (wid + 1) * per);

  // loop-invariant LDS byte offsets
  int kfrag[4];   // K A-fragment of dim chunk c: row `head` (= token within the 16-token half), swizzled slot
#pragma unroll
  for (int c = 0; c < 4; ++c) kfrag[c] = head * kMKRow + (((4 * c + quad) ^ head) << 4);
  const int vfrag = (4 * quad + (head >> 2)) * kMVRow + (head & 3) * 8;   // + 32 * db, + 16 rows for the 2nd half

  // global tile index -> (sequence b, its tile count, kv head h, tile t0 inside (b, h)); only the FIRST segment of a
  // wave's share needs this search (from the plan when there is one): every later segment starts at tile 0 of the
  // next (sequence, kv-head) pair — see `advance`
  auto locate = [&](int64_t gg, int& b_, int& nb_, int& h_, int& t0_) {
    int lo = 0, hi = batch;  // largest b with hkv * pre[b] <= gg
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)pre[mid] * hkv <= gg) lo = mid; else hi = mid;
    }
    b_ = __builtin_amdgcn_readfirstlane(lo);
    nb_ = __builtin_amdgcn_readfirstlane(pre[b_ + 1] - pre[b_]);
    const int r = (int)(gg - (int64_t)__builtin_amdgcn_readfirstlane(pre[b_]) * hkv);
    h_ = r / nb_;
    t0_ = r - h_ * nb_;
  };
  // leading tiles of sequence bb that are not its own (0 without a shared prefix: no load)
  auto sh_of = [&](int bb) { return (sh > 0 && __builtin_amdgcn_readfirstlane(member[bb]) != 0) ? sh : 0; };
  // the pair after (b_, h_): next kv-head of the same sequence, or head 0 of the next sequence that has tiles
  // (context_len 0 = graph padding). Only called when tiles remain (g + run < g1 <= total), so the scan terminates.
  auto advance = [&](int b_, int nb_, int h_, int& b2_, int& nb2_, int& h2_) {
    if (h_ + 1 < hkv) {
      b2_ = b_; nb2_ = nb_; h2_ = h_ + 1;
      return;
    }
    int bb = b_ + 1, n = 0;
    while (bb < batch) {
      const int len2 = __builtin_amdgcn_readfirstlane(ctx[bb]);
      n = len2 > 0 ? (len2 + kTile - 1) / kTile - sh_of(bb) : 0;
      if (n > 0) break;
      ++bb;
    }
    b2_ = bb; nb2_ = n; h2_ = 0;
  };
  // K/V tile loads run ONE TILE AHEAD of the matrix work, across segment boundaries too: a tile is parked in this
  // wave's LDS region before it is used, so its registers are free again and take the next tile's loads while the
  // MFMAs / softmax of the current one run; the first tile of the NEXT (b, h) segment is requested under the last
  // tile of the current one (a wave sees ~15 tiles in 1-3 segments per launch: a cold start per segment would cost
  // a full HBM round trip each). A wave that loads, waits, computes, loads ... leaves the memory queue empty during
  // its compute phases; tools/probes/hbm_pattern_probe.hip reads the same paged 8 KiB tiles at 6.8 TB/s when
  // nothing else happens between the loads.
  constexpr int kTL = KV8 ? kLoads8 : kLoads;
  u32x4_t kd[kTL], vd[kTL];
  // (tile indices are relative to the sequence's OWN share: tile ti of the share is tile ti + shb of the sequence,
  //  shb = sh_of(sequence))
  auto tile_block = [&](int bb, int ti, int shb) {
    return block_tables[(int64_t)bb * bt_stride + ((ti + shb) * kTile) / block_size];
  };
  auto tile_load = [&](int blk, int hh, int ti, int shb) {
    const int t = (ti + shb) * kTile;
    if constexpr (KV8) {
      const int64_t base = (((int64_t)blk * hkv + hh) * block_size + (t % block_size)) * 128 + lane * 16;   // bytes
      const unsigned char* kp = reinterpret_cast<const unsigned char*>(kc) + base;
      const unsigned char* vp = reinterpret_cast<const unsigned char*>(vc) + base;
#pragma unroll
      for (int i = 0; i < kLoads8; ++i)
        kd[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kp + i * 8 * 128));
#pragma unroll
      for (int i = 0; i < kLoads8; ++i)
        vd[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vp + i * 8 * 128));
    } else {
      const int64_t base = (((int64_t)blk * hkv + hh) * block_size + (t % block_size)) * 128 + lane * 8;
      const bf16_t* kp = kc + base;
      const bf16_t* vp = vc + base;
#pragma unroll
      for (int i = 0; i < kLoads; ++i) kd[i] = load16_nt(kp + i * 4 * 128);
#pragma unroll
      for (int i = 0; i < kLoads; ++i) vd[i] = load16_nt(vp + i * 4 * 128);
    }
  };

  bool prefetched = false;   // the first tile of the segment is already in kd / vd (or on its way)
  int b = 0, nb = 1, h = 0, t0 = 0;
  if (wid * per < g1) {
    if (plan != nullptr) {
      b = __builtin_amdgcn_readfirstlane(first_seg.b);
      nb = __builtin_amdgcn_readfirstlane(first_seg.nb);
      h = __builtin_amdgcn_readfirstlane(first_seg.h);
      t0 = __builtin_amdgcn_readfirstlane(first_seg.t0);
    } else {
      locate(wid * per, b, nb, h, t0);
    }
  }
  for (int64_t g = wid * per; g < g1;) {
    const int run = (int)min((int64_t)(nb - t0), g1 - g);
    const int shb = sh_of(b);
    if (!prefetched) tile_load(tile_block(b, t0, shb), h, t0, shb);
    __builtin_amdgcn_sched_barrier(0);
    // the segment after this one (its first tile is requested under this segment's last tile); it starts at tile 0
    const bool has_next = g + run < g1;
    int b2 = 0, nb2 = 1, h2 = 0;
    constexpr int t02 = 0;
    if (has_next) advance(b, nb, h, b2, nb2, h2);
    const int shb2 = has_next ? sh_of(b2) : 0;
    const int len = ctx[b];
    const bool owns_last = FUSED && (t0 + run == nb);

    // ---- segment prologue: q (FUSED: norm + rope; new token's k, v) in the 16-lane-row layout, two heads
    //      groups per pass (rq + 4 it), staged through this wave's K region into the MFMA B layout -------
    float m_run = kNegBig, l_run = 0.f;
    f32x4_t oacc[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) oacc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
      RopeRegs rr = {};
      u32x4_t wq = {0u, 0u, 0u, 0u}, wk = {0u, 0u, 0u, 0u};
      const bf16_t* row = q + (int64_t)b * hq * 128;                 // (unfused: q [batch, hq, 128])
      const int64_t row_e = FUSED ? (int64_t)b * fa.qkv_tok_stride : 0;    // fused: element offset of the token's qkv row
      if constexpr (FUSED) {
        int64_t pos = len - 1;
        pos = pos >= fa.max_pos ? fa.max_pos - 1 : pos;
        rr = load_rope_regs(fa.cos_sin + pos * 128, sub);
        if (fa.q_norm_w != nullptr) {
          wq = *reinterpret_cast<const u32x4_t*>(fa.q_norm_w + sub * 8);
          wk = *reinterpret_cast<const u32x4_t*>(fa.k_norm_w + sub * 8);
        }
      }
      // (runtime group size: the passes run as a rolled loop and the new token's scores re-read q from the staging tile —
      //  four unrolled passes push the kernel past its scalar-register limit)
      constexpr int kUnroll = G == 0 ? 1 : NIT;
      u32x4_t qh[kUnroll];
#pragma unroll kUnroll
      for (int it = 0; it < NIT; ++it) {
        u32x4_t& qq = qh[G == 0 ? 0 : it];
        qq = u32x4_t{0u, 0u, 0u, 0u};
        if (rq + 4 * it < Gv) {
          if constexpr (FUSED) qq = load_qkv8<SLABS>(q, row_e + (h * Gv + rq + 4 * it) * 128 + sub * 8, fa);
          else qq = *reinterpret_cast<const u32x4_t*>(row + (h * Gv + rq + 4 * it) * 128 + sub * 8);
        }
        if constexpr (FUSED) qq = norm_rope_head_regs(qq, fa.q_norm_w != nullptr, wq, fa.eps, rr, sub);
        *reinterpret_cast<u32x4_t*>(k_lds + (rq + 4 * it) * 256 + sub * 16) = qq;      // q tile [8 or 16 heads][128]
      }
      if constexpr (FUSED) {
        if (owns_last) {
          u32x4_t knew = load_qkv8<SLABS>(q, row_e + (hq + h) * 128 + sub * 8, fa);
          const u32x4_t vnew = load_qkv8<SLABS>(q, row_e + (hq + hkv + h) * 128 + sub * 8, fa);
          knew = norm_rope_head_regs(knew, fa.k_norm_w != nullptr, wk, fa.eps, rr, sub);
          const int tl = len - 1;
          if constexpr (KV8) {
            // quantise, store the 8 fp8 bytes, and continue with the DEQUANTISED values (what later steps will read)
            const u32x2_t kq = bf16x8_to_fp8x8(knew), vq = bf16x8_to_fp8x8(vnew);
            float kf[8], vf[8];
            unpack_fp8x4(kq[0], kf);
            unpack_fp8x4(kq[1], kf + 4);
            unpack_fp8x4(vq[0], vf);
            unpack_fp8x4(vq[1], vf + 4);
            knew = pack8(kf);
            if (rq == 0) {                                         // one 128-byte row each, for later steps
              const int blk = block_tables[(int64_t)b * bt_stride + tl / block_size];
              const int64_t dst = (((int64_t)blk * hkv + h) * block_size + (tl % block_size)) * 128 + sub * 8;
              *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned char*>(kc) + dst) = kq;
              *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned char*>(vc) + dst) = vq;
              *reinterpret_cast<u32x4_t*>(v_lds + sub * 16) = pack8(vf);               // v row for the O init
            }
          } else if (rq == 0) {                                    // one 256-byte row each, for later steps
            const int blk = block_tables[(int64_t)b * bt_stride + tl / block_size];
            const int64_t dst = (((int64_t)blk * hkv + h) * block_size + (tl % block_size)) * 128 + sub * 8;
            *reinterpret_cast<u32x4_t*>(kc + dst) = knew;
            *reinterpret_cast<u32x4_t*>(vc + dst) = vnew;
            *reinterpret_cast<u32x4_t*>(v_lds + sub * 16) = vnew;                      // v row for the O init
          }
          // scores of the new token against the 8 heads, via the same staging region
#pragma unroll kUnroll
          for (int it = 0; it < NIT; ++it) {
            const u32x4_t qq = G == 0 ? *reinterpret_cast<const u32x4_t*>(k_lds + (rq + 4 * it) * 256 + sub * 16) : qh[G == 0 ? 0 : it];
            const float sn = row16_allreduce_sum(dot8(knew, qq));
            if (sub == 0) *reinterpret_cast<float*>(v_lds + 256 + (rq + 4 * it) * 4) = sn;
          }
        }
      }
    }
    // (same wave wrote and reads: LDS operations of one wave execute in order)
    bf16x8_t qb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      u32x4_t w = {0u, 0u, 0u, 0u};
      if (head < Gv) w = *reinterpret_cast<const u32x4_t*>(k_lds + head * 256 + (4 * c + quad) * 16);
      qb[c] = __builtin_bit_cast(bf16x8_t, w);
    }
    if constexpr (FUSED) {
      if (owns_last) {
        // The new token enters the online softmax as the first key of this wave's segment (m = its score,
        // l = 1 counted once per head, O = v): its cache row is never read back inside this launch.
        m_run = *reinterpret_cast<const float*>(v_lds + 256 + (head & (4 * NIT - 1)) * 4) * scale_log2e;
        l_run = quad == 0 ? 1.f : 0.f;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
          const u32x2_t w = *reinterpret_cast<const u32x2_t*>(v_lds + (16 * db + 4 * quad) * 2);
          oacc[db] = f32x4_t{bf16lo_to_f32(w[0]), bf16hi_to_f32(w[0]), bf16lo_to_f32(w[1]), bf16hi_to_f32(w[1])};
        }
      }
    }
    const int len_cached = FUSED ? len - 1 : len;

    for (int ti = t0; ti < t0 + run; ++ti) {
      const int t = (ti + shb) * kTile;
      // which tile comes next (this segment's, or the first one of the next segment), and its block id: the table
      // lookup is issued HERE so that its round trip hides under the wait for the current tile
      const bool in_seg = ti + 1 < t0 + run;
      const bool has_pf = in_seg || has_next;
      const int pf_b = in_seg ? b : b2, pf_h = in_seg ? h : h2, pf_t = in_seg ? ti + 1 : t02;
      const int pf_sh = in_seg ? shb : shb2;
      const int pf_blk = has_pf ? tile_block(pf_b, pf_t, pf_sh) : 0;
      // registers -> this wave's LDS tile (K in swizzled 16-byte slots, V row-major padded), then the next tile's loads
      if constexpr (KV8) {
        // lane holds elements 16 (lane & 7) .. of row 8 i + (lane >> 3): bf16 16-byte chunks c0, c0 + 1 of that row
        const int r8 = lane >> 3, c0 = (lane & 7) * 2;
#pragma unroll
        for (int i = 0; i < kLoads8; ++i) {
          const int rowi = i * 8 + r8;
          u32x4_t lo16, hi16;
          fp8x16_to_bf16(kd[i], &lo16, &hi16);
          *reinterpret_cast<u32x4_t*>(k_lds + rowi * kMKRow + ((c0 ^ (rowi & 15)) << 4)) = lo16;
          *reinterpret_cast<u32x4_t*>(k_lds + rowi * kMKRow + (((c0 + 1) ^ (rowi & 15)) << 4)) = hi16;
        }
#pragma unroll
        for (int i = 0; i < kLoads8; ++i) {
          const int rowi = i * 8 + r8;
          u32x4_t lo16, hi16;
          fp8x16_to_bf16(vd[i], &lo16, &hi16);
          *reinterpret_cast<u32x4_t*>(v_lds + rowi * kMVRow + c0 * 16) = lo16;
          *reinterpret_cast<u32x4_t*>(v_lds + rowi * kMVRow + (c0 + 1) * 16) = hi16;
        }
      } else {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
          const int rowi = i * 4 + rq;
          *reinterpret_cast<u32x4_t*>(k_lds + rowi * kMKRow + ((sub ^ (rowi & 15)) << 4)) = kd[i];
        }
#pragma unroll
        for (int i = 0; i < kLoads; ++i)
          *reinterpret_cast<u32x4_t*>(v_lds + (i * 4 + rq) * kMVRow + sub * 16) = vd[i];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_pf) tile_load(pf_blk, pf_h, pf_t, pf_sh);
      __builtin_amdgcn_sched_barrier(0);

      // ---- S^T: two 16-token halves x four 32-dim chunks ------------------------------------------------
      f32x4_t sacc[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        sacc[hf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const u32x4_t a = *reinterpret_cast<const u32x4_t*>(k_lds + hf * 16 * kMKRow + kfrag[c]);
          sacc[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), qb[c], sacc[hf], 0, 0, 0);
        }
      }
      // ---- online softmax for this lane's head; token of (hf, r) = t + 16 hf + 4 quad + r ------------------
      float mx = kNegBig;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const bool valid = (t + 16 * hf + 4 * quad + rr) < len_cached;
          const float sv = valid ? sacc[hf][rr] * scale_log2e : kNegBig;
          sacc[hf][rr] = sv;
          mx = fmaxf(mx, sv);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - mn);
      m_run = mn;
      float psum = 0.f;
      bf16x8_t pb;     // P^T B operand: k-slot j <-> token 4 quad + j (j < 4), 16 + 4 quad + j - 4 (j >= 4)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float pv = exp2f(sacc[hf][rr] - mn);
          psum += pv;
          pb[hf * 4 + rr] = (bf16_t)pv;
        }
      l_run = l_run * alpha + psum;
#pragma unroll
      for (int db = 0; db < 8; ++db) oacc[db] *= alpha;
      // ---- O^T += V^T . P^T: 8 blocks of 16 dims; A = two transpose reads of 4 tokens x 16 dims --------------
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        const unsigned char* p0 = v_lds + vfrag + db * 32;
        const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
        const s16x4_t a1 =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * kMVRow));
        const s16x8_t a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), pb, oacc[db], 0, 0, 0);
      }
    }

    // ---- emit: lane (head, quad) holds O[head][16 db + 4 quad + r]; l is a per-quad partial -------------------
    const int64_t seg0 = g - t0;                                     // first global tile of (b, h)
    const int first = (int)(seg0 / per);
    const int k = (int)(wid - first);
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (head < Gv) {
      const int64_t pidx = ((int64_t)b * hq + h * Gv + head) * slots + k;
      float* dst = part_o + pidx * 128 + 4 * quad;
#pragma unroll
      for (int db = 0; db < 8; ++db) *reinterpret_cast<f32x4_t*>(dst + 16 * db) = oacc[db];
      if (quad == 0) {
        part_ml[pidx * 2] = m_run;
        part_ml[pidx * 2 + 1] = l_run;
      }
    }
    if (lane == 0) meta[b * hkv + h] = (int)((seg0 + nb - 1) / per) - first + 1;   // #partials of (b, h)
    prefetched = has_next;
    g += run;
    b = b2; nb = nb2; h = h2; t0 = 0;
  }
}

template <bool FUSED, bool KV8, int G = 8, bool SLABS = false>
__global__ __launch_bounds__(256, 2) void decode_mfma8_kernel(
    const bf16_t* __restrict__ q, bf16_t* kc, bf16_t* vc, const int32_t* __restrict__ block_tables,
    int64_t bt_stride, const int32_t* __restrict__ ctx, float* __restrict__ part_o, float* __restrict__ part_ml,
    int* __restrict__ meta, bf16_t* __restrict__ out, int batch, int hkv, int block_size, int slots,
    float scale_log2e, FusedArgs fa, const PlanHeader* __restrict__ plan, int g_rt) {
  mfma8_body<FUSED, KV8, G, SLABS>(q, kc, vc, block_tables, bt_stride, ctx, part_o, part_ml, meta, out, batch, hkv, block_size,
                                   slots, scale_log2e, fa, plan, g_rt, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------
// Shared-prefix pass (plans built with `shared_prefix`, nvl_decode_plan). When the MEMBER sequences of a step start
// with the same `sh` tiles of KV (prefix-cache hits on one system prompt — BASELINE config 3: 256 sequences x a
// 512-token prompt prefix, of which the ones prefilled after the first batch share one copy), those tiles are read once
// per PACK of 16 / G consecutive sequences instead of once per sequence (non-members of a pack ride along as zero columns): the 16
// MFMA columns that carry one sequence's G heads (plus zero padding) in decode_mfma8_kernel carry the heads of 16 / G
// sequences here — same K / V fragments, same online softmax, every column useful. One workgroup per (pack, kv head);
// its four waves take a quarter of the prefix tiles each (tile loads one tile ahead, through L2 on purpose: the other
// packs read the same tiles) and merge lane by lane through LDS. The result is ONE more split partial per (sequence,
// head) — (m, l, O) in the log2 domain like every other — in the slot the stream-K kernel never writes (slots - 1);
// decode_stream_combine_kernel folds it in. The stream-K kernel itself starts every member at tile `sh` (PlanHeader).
// The new token's q is needed here as well: FUSED recomputes norm + rotation from the raw qkv row (bit-identical to the
// stream-K kernel's, same helpers); K / V of the new token are the stream-K kernel's business alone.
//
// Round 6, last form: the pass is NOT a launch of its own any more. The packs are served by the LAST `nblocks` workgroups of
// the stream-K launch itself (decode_mfma8_shared_kernel; the plan's stream-K grid is that much smaller), each of them walking
// over its share of the (group slot, pack, kv head) items: the ~19 us latency chain of the separate launch (launch boundary,
// header -> lengths -> block table -> first tile, LDS merge) now runs UNDER the HBM-bound stream-K shares instead of in front of
// them, on the same CUs.
// A wave-uniform pointer / integer kept in VECTOR registers: the pack workgroups' walk holds more uniform state (two items, the
// header of a third, every argument of the stream-K body) than the 102 scalar registers take, and everything moved here is only
// ever used in per-lane address arithmetic.
template <typename T>
__device__ __forceinline__ T* in_vgprs(T* p) {
  const unsigned lo = (unsigned)reinterpret_cast<uintptr_t>(p), hi = (unsigned)(reinterpret_cast<uintptr_t>(p) >> 32);
  unsigned vlo, vhi;
  asm volatile("v_mov_b32 %0, %1" : "=v"(vlo) : "s"(lo));
  asm volatile("v_mov_b32 %0, %1" : "=v"(vhi) : "s"(hi));
  return reinterpret_cast<T*>(((uintptr_t)vhi << 32) | (uintptr_t)vlo);
}
__device__ __forceinline__ int in_vgprs(int x) {
  int v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
  return v;
}

template <bool FUSED, bool KV8, int G, bool SLABS>
__device__ __forceinline__ void prefix_body(
    const bf16_t* __restrict__ q_s, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, const int32_t* __restrict__ ctx_s,
    float* __restrict__ part_o_s, float* __restrict__ part_ml_s, int batch, int hkv, int block_size, int slots_s,
    float scale_log2e, const FusedArgs& fa_s, const PlanHeader* __restrict__ plan, int g_rt, const int block, const int nblocks,
    const int group_slots) {
  // a pack fills the 16 MFMA columns with P = floor(16 / G) sequences; columns P G .. 15 (group sizes that do not divide 16)
  // are zero padding. G = 0: runtime group size g_rt, as in decode_mfma8_kernel
  static_assert(G >= 0 && G <= 16, "a pack fills the 16 MFMA columns with floor(16 / G) sequences");
  const int Gv = G > 0 ? G : g_rt;
  const int P = 16 / Gv;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int sh = plan->sh_tiles;
  if (sh <= 0 || plan->batch != batch || plan->hkv != hkv) return;          // (workgroup-uniform)
  const int32_t* member = in_vgprs(plan->member);
  const bf16_t* q = in_vgprs(q_s);
  const int32_t* ctx = in_vgprs(ctx_s);
  float* part_o = in_vgprs(part_o_s);
  float* part_ml = in_vgprs(part_ml_s);
  const int slots = in_vgprs(slots_s);
  // (the fused arguments are parked in vector registers for the whole walk; an item's prologue takes the few it needs as
  //  scalars back for its own duration — `fa` below)
  FusedArgs fa_v = fa_s;
  if constexpr (FUSED) {
    fa_v.cos_sin = in_vgprs(fa_s.cos_sin);
    fa_v.q_norm_w = in_vgprs(fa_s.q_norm_w);
    fa_v.qkv_tok_stride = ((int64_t)in_vgprs((int)(fa_s.qkv_tok_stride >> 32)) << 32) | (unsigned)in_vgprs((int)fa_s.qkv_tok_stride);
    fa_v.max_pos = ((int64_t)in_vgprs((int)(fa_s.max_pos >> 32)) << 32) | (unsigned)in_vgprs((int)fa_s.max_pos);
    fa_v.eps = __int_as_float(in_vgprs(__float_as_int(fa_s.eps)));
    fa_v.qkv_splits = in_vgprs(fa_s.qkv_splits);
    fa_v.qkv_split_stride = in_vgprs(fa_s.qkv_split_stride);
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane & 15, rq = lane >> 4;      // prologue view: 16 lanes x 8 dims = one row
  const int col = lane & 15, quad = lane >> 4;    // MFMA view: one (sequence, head) column per lane
  const int hq = hkv * Gv;
  unsigned char* k_lds = smem_raw + wave * kMWaveLds;
  unsigned char* v_lds = k_lds + kTile * kMKRow;
  int kfrag[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) kfrag[c] = col * kMKRow + (((4 * c + quad) ^ col) << 4);
  const int vfrag = (4 * quad + (col >> 2)) * kMVRow + (col & 3) * 8;
  // items: (group slot, pack, kv head), slot-major — a workgroup takes items block, block + nblocks, ...
  // The walk is SOFTWARE-PIPELINED: beside the HBM-bound stream-K workgroups of the same CU a dependent round trip costs several
  // microseconds, and an item is a chain of them (lengths / group ids -> block table -> tiles -> merge). So the lengths and group
  // ids of the NEXT item are requested while the current one is computed, the next item's first tile is requested
  // before the current item's merge, and the prologue takes a row's length from the pack's header by a
  // lane shuffle instead of loading it again.
  const int per_slot = ((batch + P - 1) / P) * hkv;
  const int n_items = per_slot * group_slots;
  const int t_begin = (wave * sh) / kWaves, t_end = ((wave + 1) * sh) / kWaves;       // this wave's quarter of the prefix
  constexpr int kTL = KV8 ? kLoads8 : kLoads;
  u32x4_t kdA[kTL], vdA[kTL];
  struct Item { int b0, h, gid, tb; };          // (wave-uniform)
  // lane j < P of the header holds (context length, group id) of sequence b0 + j
  auto hdr_load = [&](int item, int& c, int& g) {
    const int pitem = item % per_slot;
    const int bj = (pitem / hkv) * P + lane;
    c = 0; g = 0;
    if (item < n_items && lane < P && bj < batch) {
      c = ctx[bj];
      g = member[bj];
    }
  };
  // Member flags are GROUP ids: rows with the same id > 0 start with the same `sh` tiles (two system prompts in one batch = two
  // groups). A pack is served once per group that has a live member in it — one item (slot) per group, that group's tiles with
  // the other rows as zero columns; almost every pack holds one group. gid == 0: nothing to do (graph padding, non-members, or
  // fewer groups in the pack than slots)
  auto resolve = [&](int item, int c, int g) {
    Item it = {0, 0, 0, 0};
    if (item >= n_items) return it;
    const int slot_y = item / per_slot;
    const int pitem = item - slot_y * per_slot;
    const int pack = pitem / hkv;
    it.h = pitem - pack * hkv;
    it.b0 = pack * P;
    const unsigned bit_j = (c > 0 && g > 0 && g < 32) ? 1u << g : 0u;
    unsigned present = 0;
    for (int j = 0; j < P; ++j) present |= (unsigned)__builtin_amdgcn_readlane((int)bit_j, j);
    unsigned rest = present;
    for (int sidx = 0; rest != 0; ++sidx) {
      const int low = __builtin_ctz(rest);
      if (sidx == slot_y) { it.gid = low; break; }
      rest &= rest - 1;
    }
    if (it.gid != 0) {
      // the block-table row the prefix tiles are looked up in: the pack's first live member of this group (all of them agree)
      const unsigned long long mine_m = __ballot(c > 0 && g == it.gid);
      it.tb = __builtin_amdgcn_readfirstlane(it.b0 + (int)__builtin_ctzll(mine_m));
    }
    return it;
  };
  auto tile_block = [&](const Item& it, int ti) { return block_tables[(int64_t)it.tb * bt_stride + (ti * kTile) / block_size]; };
  auto tile_load = [&](u32x4_t (&kd)[kTL], u32x4_t (&vd)[kTL], const Item& it, int blk, int ti) {
    const int t = ti * kTile;
    if constexpr (KV8) {
      const int64_t base = (((int64_t)blk * hkv + it.h) * block_size + (t % block_size)) * 128 + lane * 16;   // bytes
      const unsigned char* kp = reinterpret_cast<const unsigned char*>(kc) + base;
      const unsigned char* vp = reinterpret_cast<const unsigned char*>(vc) + base;
#pragma unroll
      for (int i = 0; i < kLoads8; ++i) kd[i] = *reinterpret_cast<const u32x4_t*>(kp + i * 8 * 128);
#pragma unroll
      for (int i = 0; i < kLoads8; ++i) vd[i] = *reinterpret_cast<const u32x4_t*>(vp + i * 8 * 128);
    } else {
      const int64_t base = (((int64_t)blk * hkv + it.h) * block_size + (t % block_size)) * 128 + lane * 8;
      const bf16_t* kp = kc + base;
      const bf16_t* vp = vc + base;
#pragma unroll
      for (int i = 0; i < kLoads; ++i) kd[i] = *reinterpret_cast<const u32x4_t*>(kp + i * 4 * 128);
#pragma unroll
      for (int i = 0; i < kLoads; ++i) vd[i] = *reinterpret_cast<const u32x4_t*>(vp + i * 4 * 128);
    }
  };
  // the first tile of this wave's quarter (through L2 on purpose: the other packs read the same tiles)
  auto first_tiles = [&](const Item& it) {
    if (it.gid == 0) return;
    if (t_begin < t_end) tile_load(kdA, vdA, it, tile_block(it, t_begin), t_begin);
  };

  int ctx_j, grp_j, ctx_n, grp_n;
  hdr_load(block, ctx_j, grp_j);
  hdr_load(block + nblocks, ctx_n, grp_n);
  Item cur = resolve(block, ctx_j, grp_j);
  first_tiles(cur);
  for (int item = block; item < n_items; item += nblocks) {
    __builtin_amdgcn_sched_barrier(0);
    const int gid = cur.gid, h = cur.h, b0 = cur.b0;
    float m_run = kNegBig, l_run = 0.f;
    f32x4_t oacc[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) oacc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (gid != 0) {
      // ---- q tile [16 columns][128]: column r = (sequence b0 + r / G, head h G + r % G); rows of dead sequences are zero ----
      {
        FusedArgs fa = fa_v;
        if constexpr (FUSED && SLABS) {           // (the slab loop's bound and stride are scalars again, for this prologue only)
          fa.qkv_splits = __builtin_amdgcn_readfirstlane(fa_v.qkv_splits);
          fa.qkv_split_stride = __builtin_amdgcn_readfirstlane(fa_v.qkv_split_stride);
        }
        u32x4_t wq = {0u, 0u, 0u, 0u};
        if constexpr (FUSED) {
          if (fa.q_norm_w != nullptr) wq = *reinterpret_cast<const u32x4_t*>(fa.q_norm_w + sub * 8);
        }
        // (slab sums / the runtime group size: a rolled loop, as in the stream-K body — four unrolled passes hold more scalars
        //  than there are)
        constexpr int kPxUnroll = (SLABS || G == 0) ? 1 : 4;
#pragma unroll kPxUnroll
        for (int it = 0; it < 4; ++it) {
          const int r = rq + 4 * it;
          const int sj = r / Gv, hd = r % Gv;
          const int seq = b0 + sj;
          const int seq_c = seq < batch ? seq : batch - 1;
          const int len = __shfl(ctx_j, sj < P ? sj : 0, 64);             // (lane sj of the header; lanes >= P hold 0)
          const int grp = __shfl(grp_j, sj < P ? sj : 0, 64);
          const bool live = r < P * Gv && seq < batch && len > 0 && grp == gid;
          u32x4_t qh;
          if constexpr (FUSED) {
            int64_t pos = len > 0 ? len - 1 : 0;
            pos = pos >= fa.max_pos ? fa.max_pos - 1 : pos;
            const RopeRegs rr = load_rope_regs(fa.cos_sin + pos * 128, sub);
            qh = load_qkv8<SLABS>(q, (int64_t)seq_c * fa.qkv_tok_stride + (h * Gv + hd) * 128 + sub * 8, fa);
            qh = norm_rope_head_regs(qh, fa.q_norm_w != nullptr, wq, fa.eps, rr, sub);
          } else {
            qh = *reinterpret_cast<const u32x4_t*>(q + ((int64_t)seq_c * hq + h * Gv + hd) * 128 + sub * 8);
          }
          if (!live) qh = u32x4_t{0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4_t*>(k_lds + r * 256 + sub * 16) = qh;
        }
      }
      bf16x8_t qb[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        qb[c] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(k_lds + col * 256 + (4 * c + quad) * 16));

      auto tile_step = [&](u32x4_t (&kd)[kTL], u32x4_t (&vd)[kTL], int ti) {
        const bool has_pf = ti + 1 < t_end;                 // tile loads one tile ahead
        const int pf_blk = has_pf ? tile_block(cur, ti + 1) : 0;
        if constexpr (KV8) {
          const int r8 = lane >> 3, c0 = (lane & 7) * 2;
#pragma unroll
          for (int i = 0; i < kLoads8; ++i) {
            const int rowi = i * 8 + r8;
            u32x4_t lo16, hi16;
            fp8x16_to_bf16(kd[i], &lo16, &hi16);
            *reinterpret_cast<u32x4_t*>(k_lds + rowi * kMKRow + ((c0 ^ (rowi & 15)) << 4)) = lo16;
            *reinterpret_cast<u32x4_t*>(k_lds + rowi * kMKRow + (((c0 + 1) ^ (rowi & 15)) << 4)) = hi16;
          }
#pragma unroll
          for (int i = 0; i < kLoads8; ++i) {
            const int rowi = i * 8 + r8;
            u32x4_t lo16, hi16;
            fp8x16_to_bf16(vd[i], &lo16, &hi16);
            *reinterpret_cast<u32x4_t*>(v_lds + rowi * kMVRow + c0 * 16) = lo16;
            *reinterpret_cast<u32x4_t*>(v_lds + rowi * kMVRow + (c0 + 1) * 16) = hi16;
          }
        } else {
#pragma unroll
          for (int i = 0; i < kLoads; ++i) {
            const int rowi = i * 4 + rq;
            *reinterpret_cast<u32x4_t*>(k_lds + rowi * kMKRow + ((sub ^ (rowi & 15)) << 4)) = kd[i];
          }
#pragma unroll
          for (int i = 0; i < kLoads; ++i)
            *reinterpret_cast<u32x4_t*>(v_lds + (i * 4 + rq) * kMVRow + sub * 16) = vd[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_pf) tile_load(kd, vd, cur, pf_blk, ti + 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- S^T, online softmax (every prefix token precedes every live sequence's new token: no masking), O^T ----
        f32x4_t sacc[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          sacc[hf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const u32x4_t a = *reinterpret_cast<const u32x4_t*>(k_lds + hf * 16 * kMKRow + kfrag[c]);
            sacc[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), qb[c], sacc[hf], 0, 0, 0);
          }
        }
        float mx = kNegBig;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            sacc[hf][rr] *= scale_log2e;
            mx = fmaxf(mx, sacc[hf][rr]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - mn);
        m_run = mn;
        float psum = 0.f;
        bf16x8_t pb;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float pv = exp2f(sacc[hf][rr] - mn);
            psum += pv;
            pb[hf * 4 + rr] = (bf16_t)pv;
          }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < 8; ++db) oacc[db] *= alpha;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
          const unsigned char* p0 = v_lds + vfrag + db * 32;
          const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
          const s16x4_t a1 =
              __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * kMVRow));
          const s16x8_t a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), pb, oacc[db], 0, 0, 0);
        }
      };
      for (int ti = t_begin; ti < t_end; ++ti) tile_step(kdA, vdA, ti);
      l_run += __shfl_xor(l_run, 16, 64);
      l_run += __shfl_xor(l_run, 32, 64);
    }
    // ---- the next item: its header arrived long ago; its first tiles are requested HERE, in front of this item's merge, and
    //      the header of the item after it goes out ----
    const int live_ctx = ctx_j, live_grp = grp_j;                  // (this item's header: the store mask below)
    const Item nxt = resolve(item + nblocks, ctx_n, grp_n);
    ctx_j = ctx_n; grp_j = grp_n;
    hdr_load(item + 2 * nblocks, ctx_n, grp_n);
    first_tiles(nxt);
    __builtin_amdgcn_sched_barrier(0);
    if (gid != 0) {
      // ---- the four quarters merge lane by lane (every wave holds the same (column, dims) per lane): waves 1..3 park
      //      (O, m, l) in their own LDS region as float[34][64], wave 0 folds them in and writes the partial ----
      float* mine = reinterpret_cast<float*>(k_lds);
      if (wave != 0) {
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) mine[(db * 4 + r) * 64 + lane] = oacc[db][r];
        mine[32 * 64 + lane] = m_run;
        mine[33 * 64 + lane] = l_run;
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w = 1; w < kWaves; ++w) {
          const float* other = reinterpret_cast<const float*>(smem_raw + w * kMWaveLds);
          const float mw = other[32 * 64 + lane], lw = other[33 * 64 + lane];
          const float mn = fmaxf(m_run, mw);
          const float fa_ = exp2f(m_run - mn), fb_ = exp2f(mw - mn);
          m_run = mn;
          l_run = l_run * fa_ + lw * fb_;
#pragma unroll
          for (int db = 0; db < 8; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[db][r] = oacc[db][r] * fa_ + other[(db * 4 + r) * 64 + lane] * fb_;
        }
      }
      __syncthreads();                            // the parked quarters are consumed: the regions are free for the next item
      const int sj = col / Gv, hd = col % Gv;
      const int seq = b0 + sj;
      const int len = __shfl(live_ctx, sj < P ? sj : 0, 64), grp = __shfl(live_grp, sj < P ? sj : 0, 64);
      if (wave == 0 && col < P * Gv && seq < batch && len > 0 && grp == gid) {
        const int64_t pidx = ((int64_t)seq * hq + h * Gv + hd) * slots + (slots - 1);
        float* dst = part_o + pidx * 128 + 4 * quad;
#pragma unroll
        for (int db = 0; db < 8; ++db) *reinterpret_cast<f32x4_t*>(dst + 16 * db) = oacc[db];
        if (quad == 0) {
          part_ml[pidx * 2] = m_run;
          part_ml[pidx * 2 + 1] = l_run;
        }
      }
    }
    cur = nxt;
  }   // items
}

// One launch for a step with a shared prefix: workgroups [0, main_blocks) are the stream-K grid the plan was built for
// (main_blocks = plan->nwaves / 4), workgroups [main_blocks, gridDim.x) serve the shared-prefix packs (px_split() on the host side
// decides the division).
template <bool FUSED, bool KV8, int G, bool SLABS>
__global__ __launch_bounds__(256, 2) void decode_mfma8_shared_kernel(
    const bf16_t* __restrict__ q, bf16_t* kc, bf16_t* vc, const int32_t* __restrict__ block_tables,
    int64_t bt_stride, const int32_t* __restrict__ ctx, float* __restrict__ part_o, float* __restrict__ part_ml,
    int* __restrict__ meta, bf16_t* __restrict__ out, int batch, int hkv, int block_size, int slots,
    float scale_log2e, FusedArgs fa, const PlanHeader* __restrict__ plan, int g_rt, int main_blocks, int group_slots) {
  if ((int)blockIdx.x < main_blocks)
    mfma8_body<FUSED, KV8, G, SLABS>(q, kc, vc, block_tables, bt_stride, ctx, part_o, part_ml, meta, out, batch, hkv, block_size,
                                     slots, scale_log2e, fa, plan, g_rt, (int)blockIdx.x, main_blocks);
  else
    prefix_body<FUSED, KV8, G, SLABS>(q, kc, vc, block_tables, bt_stride, ctx, part_o, part_ml, batch, hkv, block_size, slots,
                                      scale_log2e, fa, plan, g_rt, (int)blockIdx.x - main_blocks, (int)gridDim.x - main_blocks,
                                      group_slots);
}

// The pack workgroups alone — the instantiations whose stream-K body has no scalar registers to spare for a second body in the
// same kernel (split-K slab prologue, runtime group size) run the pass as a launch of its own in front of the plain kernel.
template <bool FUSED, bool KV8, int G, bool SLABS>
__global__ __launch_bounds__(256, 2) void decode_px_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, const int32_t* __restrict__ ctx,
    float* __restrict__ part_o, float* __restrict__ part_ml, int batch, int hkv, int block_size, int slots,
    float scale_log2e, FusedArgs fa, const PlanHeader* __restrict__ plan, int g_rt) {
  prefix_body<FUSED, KV8, G, SLABS>(q, kc, vc, block_tables, bt_stride, ctx, part_o, part_ml, batch, hkv, block_size, slots,
                                    scale_log2e, fa, plan, g_rt, (int)blockIdx.x, (int)gridDim.x, plan->px_groups);
}

__global__ __launch_bounds__(128) void decode_stream_combine_kernel(const float* __restrict__ part_o,
                                                                     const float* __restrict__ part_ml,
                                                                     const int* __restrict__ meta,
                                                                     const int32_t* __restrict__ ctx,
                                                                     bf16_t* __restrict__ out, int hq, int hkv,
                                                                     int slots, float* __restrict__ lse,
                                                                     const PlanHeader* __restrict__ plan,
                                                                     int prefix_slot) {
  // out[b, head, :] = sum_k 2^(m_k - M) O_k / sum_k 2^(m_k - M) l_k over the (b, h) segment's split partials;
  // zero rows when the sequence is padding. The kernel is pure latency (a few KB per block), so everything a
  // typical segment needs — the count, and the first kSpec (m, l, O) slots — is loaded SPECULATIVELY in one
  // round (the slots exist in the workspace whatever the count; unused ones are never used in arithmetic);
  // only segments split over more than kSpec waves take a second round.
  constexpr int kSpec = 6;
  const int b = blockIdx.x / hq;
  const int head = blockIdx.x - b * hq;
  const int64_t row = blockIdx.x;
  const int d = threadIdx.x;
  const float* ml = part_ml + row * slots * 2;
  const float* po = part_o + row * slots * 128;
  const int len = ctx[b];
  const int cnt_raw = meta[b * hkv + head / (hq / hkv)];
  // the shared-prefix pass's partial (prefix_slot >= 0: the launch belongs to a plan with a shared prefix; whether THIS
  // step has one is in the plan header) — requested with everything else, folded in last
  const int ps = prefix_slot >= 0 ? prefix_slot : 0;
  const float m_p = ml[ps * 2], l_p = ml[ps * 2 + 1], o_p = po[ps * 128 + d];
  const bool has_pre = prefix_slot >= 0 && plan->sh_tiles > 0 && plan->member[b] != 0;
  float m_s[kSpec], l_s[kSpec], o_s[kSpec];
#pragma unroll
  for (int c = 0; c < kSpec; ++c) {
    const int cc = c < slots ? c : 0;
    m_s[c] = ml[cc * 2];
    l_s[c] = ml[cc * 2 + 1];
    o_s[c] = po[cc * 128 + d];
  }
  const int cnt = len > 0 ? cnt_raw : 0;
  float M = kNegBig;
#pragma unroll
  for (int c = 0; c < kSpec; ++c)
    if (c < cnt) M = fmaxf(M, m_s[c]);
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int c = 0; c < kSpec; ++c)
    if (c < cnt) {
      const float f = exp2f(m_s[c] - M);
      num += f * o_s[c];
      den += f * l_s[c];
    }
  // Segments split over more than kSpec waves (long contexts on few kv heads: 16k tokens on ONE kv head — Qwen3-32B
  // per rank at TP = 8 — is 125 partials): online over rounds of kBatch slots whose 3 x kBatch loads are all issued
  // before the first use (the one-load-at-a-time loop this replaces paid a memory round trip per slot, ~60 us at 125
  // slots). Same value up to fp32 rounding of the running rescale; ordinary segments (<= kSpec slots) are unchanged.
  constexpr int kBatch = 8;
  for (int c0 = kSpec; c0 < cnt; c0 += kBatch) {
    float mb[kBatch], lb[kBatch], ob[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const int c = c0 + j < cnt ? c0 + j : cnt - 1;            // clamped: a valid slot, masked below
      mb[j] = ml[c * 2];
      lb[j] = ml[c * 2 + 1];
      ob[j] = po[c * 128 + d];
    }
    float Mb = M;
#pragma unroll
    for (int j = 0; j < kBatch; ++j)
      if (c0 + j < cnt) Mb = fmaxf(Mb, mb[j]);
    const float r = exp2f(M - Mb);                              // rescale what has been summed so far
    num *= r;
    den *= r;
    M = Mb;
#pragma unroll
    for (int j = 0; j < kBatch; ++j)
      if (c0 + j < cnt) {
        const float f = exp2f(mb[j] - M);
        num += f * ob[j];
        den += f * lb[j];
      }
  }
  if (has_pre && cnt > 0) {
    const float Mb = fmaxf(M, m_p);
    const float r = exp2f(M - Mb), f = exp2f(m_p - Mb);
    num = num * r + f * o_p;
    den = den * r + f * l_p;
    M = Mb;
  }
  out[row * 128 + d] = (bf16_t)(cnt > 0 ? num / den : 0.f);
  // optional log-sum-exp of the scaled scores (natural log; what flash-attn returns as softmax_lse): the scores are
  // kept in the log2 domain here, so LSE = ln 2 * (M + log2 den); -inf for padded rows
  if (lse != nullptr && d == 0) lse[row] = cnt > 0 ? 0.6931471805599453f * (M + log2f(den)) : -INFINITY;
}

// (A smaller minimum share per wave for small steps — 2 tiles instead of kMinTilesPerWave = 4, tried in round 4 on the
// one-kv-head shape of Qwen3-32B per rank at TP = 8 — moves 1.2 us from the main kernel into the split merge, which then
// has twice the partials: 19.1 + 4.5 -> 18.0 + 5.6 us per launch, nothing on the bench shape.
// profiles/r04_decode_min_tiles_ab.json)
// (+ 1: the last slot belongs to the shared-prefix pass — prefix_body — and is never written by a stream-K wave)
inline int stream_slots(int64_t max_context) { return (int)(max_context / (kTile * kMinTilesPerWave)) + 3; }

// LDS and grid of decode_mfma8_kernel — shared by its launcher and by nvl_decode_plan, whose per-wave records are only
// valid for the grid they were made for. `prefix_batch`: sequences whose tile prefix the kernel keeps in LDS (0 with a plan).
inline size_t mfma8_lds_bytes(int64_t prefix_batch) {
  return (size_t)kWaves * kMWaveLds + kWaves * sizeof(int) + (size_t)(prefix_batch + 1) * sizeof(int);
}
inline int64_t mfma8_grid(int64_t batch, int hkv, int64_t max_context, bool planned) {
  const size_t lds = mfma8_lds_bytes(planned ? 0 : batch);
  int64_t grid = (int64_t)nvl_device_cu_count() * (2 * lds <= 160 * 1024 ? 2 : 1);
  const int64_t max_tiles = batch * hkv * ((max_context + kTile - 1) / kTile);
  const int64_t max_wg = (max_tiles + kWaves * kMinTilesPerWave - 1) / (kWaves * kMinTilesPerWave);
  if (grid > max_wg) grid = max_wg;
  return grid < 1 ? 1 : grid;
}
// A step with a shared prefix (plan built with `groups` > 0 group slots): the launch's workgroups are divided between the
// stream-K grid (`main_blocks`: what the plan's wave records are made for) and the workgroups that serve the shared-prefix
// packs (`px_blocks`; decode_mfma8_shared_kernel). One prefix workgroup per px_items_per_wg() = 2 items (an item = one pack x kv head
// x group slot: a ~4-5 us latency chain, against ~7 us per tile of a stream-K wave's HBM-bound share), at most a third of the
// resident workgroups. nvl_decode_plan and the launcher both call this: the plan is only valid for `main_blocks`.
inline int px_items_per_wg() {
  static const int v = [] {
    const char* e = getenv("NVL_PX_ITEMS_PER_WG");
    const int x = e != nullptr ? atoi(e) : 0;
    return x >= 1 && x <= 64 ? x : 2;
  }();
  return v;
}
inline void px_split(int64_t batch, int hkv, int Gv, int64_t max_context, int groups, int64_t* main_blocks, int64_t* px_blocks) {
  const int64_t full = mfma8_grid(batch, hkv, max_context, true);
  const int64_t resident = (int64_t)nvl_device_cu_count() * 2;
  const int P = 16 / Gv;
  const int64_t items = ((batch + P - 1) / P) * hkv * groups;
  if (Gv != 2 && Gv != 4 && Gv != 8) {          // runtime-G instantiation: the pass is a launch of its own, one item per workgroup
    *main_blocks = full;
    *px_blocks = items;
    return;
  }
  int64_t px = (items + px_items_per_wg() - 1) / px_items_per_wg();
  if (px > resident / 3) px = resident / 3;
  if (px < 1) px = 1;
  int64_t mb = full < resident - px ? full : resident - px;
  if (mb < 1) mb = 1;
  *main_blocks = mb;
  *px_blocks = px;
}

int plan_shadow_prefix(const void* plan);       // group slots of the shared-prefix pass `plan` was built with (0: none; host-side shadow, below)

template <bool FUSED, bool KV8, int G, bool SLABS>
int launch_decode_mfma8_s(const void* q, void* kc, void* vc, const int32_t* bt, int64_t bt_stride, const int32_t* ctx,
                          void* out, int64_t batch, int hkv, int block_size, int64_t max_context, float scale,
                          void* workspace, hipStream_t s, const FusedArgs& fa, const void* plan, float* lse, int prefix,
                          int g_rt);

// qkv as fp32 split-K slabs (fa.qkv_splits > 0) is an instantiation of its own: the bf16 form keeps its registers
template <bool FUSED, bool KV8, int G = 8>
int launch_decode_mfma8(const void* q, void* kc, void* vc, const int32_t* bt, int64_t bt_stride, const int32_t* ctx,
                        void* out, int64_t batch, int hkv, int block_size, int64_t max_context, float scale,
                        void* workspace, hipStream_t s, const FusedArgs& fa, const void* plan, float* lse, int g_rt = 0) {
  const int prefix = plan != nullptr ? plan_shadow_prefix(plan) : 0;     // group slots of the plan's shared-prefix pass (0: none)
  if constexpr (FUSED) {
    if (fa.qkv_splits > 0)
      return launch_decode_mfma8_s<FUSED, KV8, G, true>(q, kc, vc, bt, bt_stride, ctx, out, batch, hkv, block_size, max_context,
                                                        scale, workspace, s, fa, plan, lse, prefix, g_rt);
  }
  return launch_decode_mfma8_s<FUSED, KV8, G, false>(q, kc, vc, bt, bt_stride, ctx, out, batch, hkv, block_size, max_context,
                                                     scale, workspace, s, fa, plan, lse, prefix, g_rt);
}

template <bool FUSED, bool KV8, int G, bool SLABS>
int launch_decode_mfma8_s(const void* q, void* kc, void* vc, const int32_t* bt, int64_t bt_stride, const int32_t* ctx,
                          void* out, int64_t batch, int hkv, int block_size, int64_t max_context, float scale,
                          void* workspace, hipStream_t s, const FusedArgs& fa, const void* plan, float* lse, int prefix,
                          int g_rt) {
  const int Gv = G > 0 ? G : g_rt;
  const int hq = hkv * Gv;
  const int slots = stream_slots(max_context);
  float* part_o = (float*)workspace;
  float* part_ml = part_o + (size_t)batch * hq * slots * 128;
  int* meta = (int*)(part_ml + (size_t)batch * hq * slots * 2);
  // with a per-step plan the kernel keeps no tile prefix in LDS
  const size_t lds = mfma8_lds_bytes(plan ? 0 : batch);
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& attr_set = attr_done[nvl_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_mfma8_kernel<FUSED, KV8, G, SLABS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      nvl_set_error("nvl_paged_attn_decode: cannot reserve LDS for the matrix-core kernel");
      return NVL_ELAUNCH;
    }
    // ... and for the shared-prefix form of the same instantiation (the stream-K grid + the pack workgroups in one launch):
    // made HERE, with the plain kernel's — the first launch of an instantiation is an eager warm-up, while the first launch WITH the pass may
    // sit inside a stream capture (the engine captures a bucket's prefix graph when a step first wants it)
    const void* px_kernel;
    if constexpr (G == 0 || SLABS) px_kernel = reinterpret_cast<const void*>(&decode_px_kernel<FUSED, KV8, G, SLABS>);
    else px_kernel = reinterpret_cast<const void*>(&decode_mfma8_shared_kernel<FUSED, KV8, G, SLABS>);
    if (hipFuncSetAttribute(px_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      nvl_set_error("nvl_paged_attn_decode: cannot reserve LDS for the shared-prefix kernel");
      return NVL_ELAUNCH;
    }
    attr_set = true;
  }
  NVL_REQUIRE(lds <= 160 * 1024, "nvl_paged_attn_decode: batch=%lld needs %zu B of LDS (> 160 KiB)", (long long)batch, lds);
  if (prefix) {
    // shared prefix: ONE launch — the stream-K grid the plan was built for, plus the workgroups that serve the packs (they
    // read q / K / V only and write the partial slot no stream-K wave writes, so nothing orders the two kinds of workgroup;
    // the merge launch behind sees both)
    int64_t main_blocks, px_blocks;
    px_split(batch, hkv, Gv, max_context, prefix, &main_blocks, &px_blocks);
    if constexpr (G == 0 || SLABS) {
      // (two launches: the packs, then the plain kernel on the grid the plan was built for — the full one for a runtime group
      //  size; a slab-sum launch, which the plan could not know about, keeps the divided grid)
      hipLaunchKernelGGL((decode_px_kernel<FUSED, KV8, G, SLABS>), dim3((unsigned)px_blocks), dim3(256), kWaves * kMWaveLds, s,
                         (const bf16_t*)q, (const bf16_t*)kc, (const bf16_t*)vc, bt, bt_stride, ctx, part_o, part_ml, (int)batch, hkv,
                         block_size, slots, scale * 1.4426950408889634f, fa, (const PlanHeader*)plan, g_rt);
      hipLaunchKernelGGL((decode_mfma8_kernel<FUSED, KV8, G, SLABS>), dim3((unsigned)main_blocks), dim3(256), lds, s, (const bf16_t*)q,
                         (bf16_t*)kc, (bf16_t*)vc, bt, bt_stride, ctx, part_o, part_ml, meta, (bf16_t*)out, (int)batch, hkv,
                         block_size, slots, scale * 1.4426950408889634f, fa, (const PlanHeader*)plan, g_rt);
    } else {
      hipLaunchKernelGGL((decode_mfma8_shared_kernel<FUSED, KV8, G, SLABS>), dim3((unsigned)(main_blocks + px_blocks)), dim3(256),
                         lds, s, (const bf16_t*)q, (bf16_t*)kc, (bf16_t*)vc, bt, bt_stride, ctx, part_o, part_ml, meta, (bf16_t*)out,
                         (int)batch, hkv, block_size, slots, scale * 1.4426950408889634f, fa, (const PlanHeader*)plan, g_rt,
                         (int)main_blocks, prefix);
    }
  } else {
    const int64_t grid = mfma8_grid(batch, hkv, max_context, plan != nullptr);
    hipLaunchKernelGGL((decode_mfma8_kernel<FUSED, KV8, G, SLABS>), dim3((unsigned)grid), dim3(256), lds, s, (const bf16_t*)q,
                       (bf16_t*)kc, (bf16_t*)vc, bt, bt_stride, ctx, part_o, part_ml, meta, (bf16_t*)out, (int)batch, hkv,
                       block_size, slots, scale * 1.4426950408889634f, fa, (const PlanHeader*)plan, g_rt);
  }
  hipLaunchKernelGGL(decode_stream_combine_kernel, dim3((unsigned)(batch * hq)), dim3(128), 0, s, part_o, part_ml,
                     meta, ctx, (bf16_t*)out, hq, hkv, slots, lse, (const PlanHeader*)plan, prefix ? slots - 1 : -1);
  return nvl_check_launch("nvl_paged_attn_decode");
}

template <int G, bool FUSED>
int launch_decode_stream_fp8(const void* q, void* kc, void* vc, const int32_t* bt, int64_t bt_stride,
                             const int32_t* ctx, void* out, int64_t batch, int hkv, int block_size,
                             int64_t max_context, float scale, void* workspace, hipStream_t s, const FusedArgs& fa,
                             const void* /*plan: only the matrix-core kernel consumes one*/, float* lse) {
  const int hq = hkv * G;
  const int slots = stream_slots(max_context);
  float* part_o = (float*)workspace;
  float* part_ml = part_o + (size_t)batch * hq * slots * 128;
  int* meta = (int*)(part_ml + (size_t)batch * hq * slots * 2);
  const size_t lds = kWaves * sizeof(int) + (size_t)(batch + 1) * sizeof(int);
  int64_t grid = (int64_t)nvl_device_cu_count() * 2;
  const int64_t max_tiles = batch * hkv * ((max_context + kTile - 1) / kTile);
  const int64_t max_wg = (max_tiles + kWaves * kMinTilesPerWave - 1) / (kWaves * kMinTilesPerWave);
  if (grid > max_wg) grid = max_wg;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((decode_stream_fp8_kernel<G, FUSED>), dim3((unsigned)grid), dim3(256), lds, s, (const bf16_t*)q,
                     (unsigned char*)kc, (unsigned char*)vc, bt, bt_stride, ctx, part_o, part_ml, meta, (int)batch, hkv,
                     block_size, slots, scale * 1.4426950408889634f, fa);
  hipLaunchKernelGGL(decode_stream_combine_kernel, dim3((unsigned)(batch * hq)), dim3(128), 0, s, part_o, part_ml,
                     meta, ctx, (bf16_t*)out, hq, hkv, slots, lse, (const PlanHeader*)nullptr, -1);
  return nvl_check_launch("nvl_paged_attn_decode");
}

template <int G, bool FUSED>
int launch_decode_stream(const void* q, void* kc, void* vc, const int32_t* bt, int64_t bt_stride,
                         const int32_t* ctx, void* out, int64_t batch, int hkv, int block_size, int64_t max_context,
                         float scale, void* workspace, hipStream_t s, const FusedArgs& fa, const void* /*plan*/,
                         float* lse) {
  const int hq = hkv * G;
  const int slots = stream_slots(max_context);
  float* part_o = (float*)workspace;
  float* part_ml = part_o + (size_t)batch * hq * slots * 128;
  int* meta = (int*)(part_ml + (size_t)batch * hq * slots * 2);
  const size_t lds = kWaves * sizeof(int) + (size_t)(batch + 1) * sizeof(int);
  static int per_cu_dev[NVL_MAX_DEVICES] = {};
  int& per_cu = per_cu_dev[nvl_device_slot()];
  const int cus = nvl_device_cu_count();
  if (per_cu == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, decode_stream_kernel<G, FUSED>, 256, lds) != hipSuccess || n < 1) n = 2;
    // 2 workgroups (8 waves, 128 KiB of loads in flight) per CU saturate HBM; measured on the bench replay:
    // 1 / 2 / 3 per CU = 93.3 / 89.4 / 91.1 us per launch (profiles/README.md)
    per_cu = n > 2 ? 2 : n;
    if (const char* e = getenv("NVL_DECODE_WGS_PER_CU")) { const int v = atoi(e); if (v >= 1 && v < per_cu) per_cu = v; }
  }
  int64_t grid = (int64_t)cus * per_cu;
  const int64_t max_tiles = batch * hkv * ((max_context + kTile - 1) / kTile);
  const int64_t max_wg = (max_tiles + kWaves * kMinTilesPerWave - 1) / (kWaves * kMinTilesPerWave);
  if (grid > max_wg) grid = max_wg;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((decode_stream_kernel<G, FUSED>), dim3((unsigned)grid), dim3(256), lds, s, (const bf16_t*)q,
                     (bf16_t*)kc, (bf16_t*)vc, bt, bt_stride, ctx, part_o, part_ml, meta, (int)batch, hkv,
                     block_size, slots, scale * 1.4426950408889634f, fa);
  hipLaunchKernelGGL(decode_stream_combine_kernel, dim3((unsigned)(batch * hq)), dim3(128), 0, s, part_o, part_ml,
                     meta, ctx, (bf16_t*)out, hq, hkv, slots, lse, (const PlanHeader*)nullptr, -1);
  return nvl_check_launch("nvl_paged_attn_decode");
}

}  // namespace

extern "C" size_t nvl_paged_attn_decode_workspace_bytes(int64_t max_batch, int num_q_heads, int64_t max_context) {
  if (max_batch <= 0 || num_q_heads <= 0 || max_context <= 0) return 0;
  return (size_t)max_batch * num_q_heads * (stream_slots(max_context) * 130 * sizeof(float) + sizeof(int));
}

namespace {

// Group sizes 2 and 4 also run on the matrix-core kernel (heads padded to one 16-column tile: the matrix pipe is idle
// anyway, and the tile loads run one tile ahead of the matrix work, which the register-resident packed-dot kernel
// cannot do without a second tile of registers). NVL_DECODE_MFMA=0 keeps the packed-dot kernels (A/B).
bool use_mfma_small_g() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("NVL_DECODE_MFMA");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

bool use_valu_g8() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("NVL_DECODE_G8_VALU");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

// Host-side shadow of the plans nvl_decode_plan has enqueued (keyed by the plan buffer's address): the attention entry
// points compare the geometry a plan was built for with the launch they are about to make — a plan is a list of
// per-wave records for ONE (batch, Hkv, max_context, device) and the kernel indexes it by wave id.
struct PlanShadow { const void* plan; int64_t batch, max_context; int hkv, dev; int prefix; };
static PlanShadow g_plan_shadow[64];
static int g_plan_shadow_n = 0, g_plan_shadow_next = 0;
static std::mutex g_plan_shadow_mu;

static void plan_shadow_put(const void* plan, int64_t batch, int hkv, int64_t max_context, int prefix) {
  std::lock_guard<std::mutex> lock(g_plan_shadow_mu);
  const PlanShadow rec = {plan, batch, max_context, hkv, nvl_device_slot(), prefix};
  for (int i = 0; i < g_plan_shadow_n; ++i)
    if (g_plan_shadow[i].plan == plan && g_plan_shadow[i].dev == rec.dev) { g_plan_shadow[i] = rec; return; }
  if (g_plan_shadow_n < 64) { g_plan_shadow[g_plan_shadow_n++] = rec; return; }
  g_plan_shadow[g_plan_shadow_next] = rec;                 // more than 64 live plan buffers: forget the oldest
  g_plan_shadow_next = (g_plan_shadow_next + 1) % 64;
}

int plan_shadow_prefix(const void* plan) {
  std::lock_guard<std::mutex> lock(g_plan_shadow_mu);
  const int dev = nvl_device_slot();
  for (int i = 0; i < g_plan_shadow_n; ++i)
    if (g_plan_shadow[i].plan == plan && g_plan_shadow[i].dev == dev) return g_plan_shadow[i].prefix;
  return 0;
}

static int plan_shadow_check(const void* plan, int64_t batch, int hkv, int64_t max_context, const char* who) {
  std::lock_guard<std::mutex> lock(g_plan_shadow_mu);
  const int dev = nvl_device_slot();
  for (int i = 0; i < g_plan_shadow_n; ++i)
    if (g_plan_shadow[i].plan == plan && g_plan_shadow[i].dev == dev) {
      const PlanShadow& r = g_plan_shadow[i];
      NVL_REQUIRE(r.batch == batch && r.hkv == hkv && r.max_context == max_context,
                  "%s: the plan was built for batch=%lld, Hkv=%d, max_context=%lld; this launch has batch=%lld, Hkv=%d, "
                  "max_context=%lld (a plan serves exactly the step it was made for)", who, (long long)r.batch, r.hkv,
                  (long long)r.max_context, (long long)batch, hkv, (long long)max_context);
      return NVL_OK;
    }
  NVL_REQUIRE(false, "%s: plan %p was not produced by nvl_decode_plan on this device", who, plan);
  return NVL_OK;
}

int decode_common(const void* q, void* k_cache, void* v_cache, const int32_t* block_tables, int64_t bt_stride,
                  const int32_t* context_lens, void* out, int64_t batch, int num_q_heads, int num_kv_heads,
                  int block_size, int64_t num_blocks, int64_t max_context, float softmax_scale, void* workspace,
                  size_t workspace_bytes, void* stream, const FusedArgs* fa, const char* who, int kv_dtype,
                  const void* plan, float* lse) {
  NVL_REQUIRE(q && k_cache && v_cache && block_tables && context_lens && out && workspace, "%s: null pointer", who);
  NVL_REQUIRE(kv_dtype == NVL_KV_BF16 || kv_dtype == NVL_KV_FP8, "%s: kv_dtype=%d (0 bf16, 1 fp8 e4m3)", who, kv_dtype);
  NVL_REQUIRE(batch >= 0 && batch <= 32768, "%s: batch=%lld out of range [0, 32768]", who, (long long)batch);
  NVL_REQUIRE(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, "%s: Hq=%d not a multiple of Hkv=%d", who, num_q_heads, num_kv_heads);
  NVL_REQUIRE(block_size > 0 && block_size % kTile == 0, "%s: block_size=%d must be a multiple of %d", who, block_size, kTile);
  NVL_REQUIRE(num_blocks > 0 && max_context > 0, "%s: bad cache geometry", who);
  NVL_REQUIRE(bt_stride * (int64_t)block_size >= max_context, "%s: block table (stride %lld) narrower than max_context=%lld", who, (long long)bt_stride, (long long)max_context);
  NVL_REQUIRE(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out | (uintptr_t)workspace) % 16 == 0,
              "%s: pointers must be 16-byte aligned", who);
  NVL_REQUIRE(((uintptr_t)plan % 16 == 0) && ((uintptr_t)lse % 4 == 0), "%s: plan must be 16-byte, lse 4-byte aligned", who);
  if (batch == 0) return NVL_OK;
  if (plan != nullptr) {
    const int rc = plan_shadow_check(plan, batch, num_kv_heads, max_context, who);
    if (rc != NVL_OK) return rc;
  }
  const int G = num_q_heads / num_kv_heads;
  const size_t need = nvl_paged_attn_decode_workspace_bytes(batch, num_q_heads, max_context);
  NVL_REQUIRE(workspace_bytes >= need, "%s: workspace %zu B < required %zu B", who, workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  const FusedArgs none = {};
  NVL_REQUIRE(G <= 16, "%s: group size Hq/Hkv=%d exceeds 16 (the heads of a kv group fill one 16-column matrix tile)", who, G);
  // Which kernel: G = 1 packed-dot; G = 2 / 4 / 8 their own matrix-core instantiations; every other group size up to 16
  // (Qwen3-14B: 40 / 8 = 5, also per rank at TP = 2 / 4 / 8) the runtime-G instantiation of the same kernel.
  const bool generic = G > 1 && G != 2 && G != 4 && G != 8;
  const bool mfma = generic || (G == 8 && (kv_dtype == NVL_KV_FP8 || !use_valu_g8())) || ((G == 2 || G == 4) && use_mfma_small_g());
  if (fa && fa->qkv_splits > 0 && !mfma) {
    // fp32 split-K slabs as the qkv input: only the matrix-core kernel's prologue sums them
    nvl_set_error("%s: qkv_splits > 0 needs the matrix-core kernel (Hq/Hkv in 2 ... 16; got %d)", who, G);
    return NVL_EUNSUPPORTED;
  }
  if (plan != nullptr && plan_shadow_prefix(plan)) {
    // a plan with a shared prefix starts every sequence's stream-K share behind it: only the matrix-core kernel knows
    NVL_REQUIRE(mfma, "%s: a plan with a shared prefix needs the matrix-core kernel (Hq/Hkv in 2 ... 16; got %d)", who, G);
    NVL_REQUIRE(block_size % 128 == 0, "%s: a shared prefix needs block_size %% 128 == 0 (got %d)", who, block_size);
  }
  if (kv_dtype == NVL_KV_FP8) {
#define NVL_DECODE8_CASE(GG)                                                                                           \
  case GG:                                                                                                             \
    return fa ? launch_decode_stream_fp8<GG, true>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,   \
                                                   batch, num_kv_heads, block_size, max_context, softmax_scale,       \
                                                   workspace, s, *fa, plan, lse)                                                 \
              : launch_decode_stream_fp8<GG, false>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,  \
                                                    batch, num_kv_heads, block_size, max_context, softmax_scale,      \
                                                    workspace, s, none, plan, lse);
    if (generic)
      return fa ? launch_decode_mfma8<true, true, 0>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,
                                                     batch, num_kv_heads, block_size, max_context, softmax_scale,
                                                     workspace, s, *fa, plan, lse, G)
                : launch_decode_mfma8<false, true, 0>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,
                                                      batch, num_kv_heads, block_size, max_context, softmax_scale,
                                                      workspace, s, none, plan, lse, G);
    if (use_mfma_small_g() && (G == 2 || G == 4)) {
#define NVL_MFMA8_G(GG)                                                                                               \
  if (G == GG)                                                                                                        \
    return fa ? launch_decode_mfma8<true, true, GG>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,  \
                                                    batch, num_kv_heads, block_size, max_context, softmax_scale,      \
                                                    workspace, s, *fa, plan, lse)                                                \
              : launch_decode_mfma8<false, true, GG>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out, \
                                                     batch, num_kv_heads, block_size, max_context, softmax_scale,     \
                                                     workspace, s, none, plan, lse);
      NVL_MFMA8_G(2)
      NVL_MFMA8_G(4)
#undef NVL_MFMA8_G
    }
    switch (G) {
      NVL_DECODE8_CASE(1)
      NVL_DECODE8_CASE(2)
      NVL_DECODE8_CASE(4)
      case 8:   // matrix-core variant: fp8 tiles are converted to bf16 on their way into LDS
        return fa ? launch_decode_mfma8<true, true>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,
                                                    batch, num_kv_heads, block_size, max_context, softmax_scale,
                                                    workspace, s, *fa, plan, lse)
                  : launch_decode_mfma8<false, true>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,
                                                     batch, num_kv_heads, block_size, max_context, softmax_scale,
                                                     workspace, s, none, plan, lse);
      default:
        nvl_set_error("%s: unsupported group size Hq/Hkv=%d (supported 1 ... 16)", who, G);
        return NVL_EUNSUPPORTED;
    }
#undef NVL_DECODE8_CASE
  }
#define NVL_DECODE_CASE(GG)                                                                                        \
  case GG:                                                                                                         \
    return fa ? launch_decode_stream<GG, true>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,   \
                                               batch, num_kv_heads, block_size, max_context, softmax_scale,       \
                                               workspace, s, *fa, plan, lse)                                                 \
              : launch_decode_stream<GG, false>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,  \
                                                batch, num_kv_heads, block_size, max_context, softmax_scale,      \
                                                workspace, s, none, plan, lse);
  if (generic)
    return fa ? launch_decode_mfma8<true, false, 0>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,
                                                    batch, num_kv_heads, block_size, max_context, softmax_scale,
                                                    workspace, s, *fa, plan, lse, G)
              : launch_decode_mfma8<false, false, 0>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,
                                                     batch, num_kv_heads, block_size, max_context, softmax_scale,
                                                     workspace, s, none, plan, lse, G);
  if (use_mfma_small_g() && (G == 2 || G == 4)) {
#define NVL_MFMA_G(GG)                                                                                                \
  if (G == GG)                                                                                                        \
    return fa ? launch_decode_mfma8<true, false, GG>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out, \
                                                     batch, num_kv_heads, block_size, max_context, softmax_scale,     \
                                                     workspace, s, *fa, plan, lse)                                               \
              : launch_decode_mfma8<false, false, GG>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,\
                                                      batch, num_kv_heads, block_size, max_context, softmax_scale,    \
                                                      workspace, s, none, plan, lse);
    NVL_MFMA_G(2)
    NVL_MFMA_G(4)
#undef NVL_MFMA_G
  }
  switch (G) {
    NVL_DECODE_CASE(1)
    NVL_DECODE_CASE(2)
    NVL_DECODE_CASE(4)
    case 8:   // matrix-core variant (the packed-dot kernel is VALU-bound at 8 FLOP/B); NVL_DECODE_G8_VALU=1 keeps it
      if (!use_valu_g8())
        return fa ? launch_decode_mfma8<true, false>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,
                                                     batch, num_kv_heads, block_size, max_context, softmax_scale,
                                                     workspace, s, *fa, plan, lse)
                  : launch_decode_mfma8<false, false>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out,
                                                      batch, num_kv_heads, block_size, max_context, softmax_scale,
                                                      workspace, s, none, plan, lse);
      return fa ? launch_decode_stream<8, true>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out, batch,
                                                num_kv_heads, block_size, max_context, softmax_scale, workspace, s, *fa, plan, lse)
                : launch_decode_stream<8, false>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out, batch,
                                                 num_kv_heads, block_size, max_context, softmax_scale, workspace, s, none, plan, lse);
    default:
      nvl_set_error("%s: unsupported group size Hq/Hkv=%d (supported 1 ... 16)", who, G);
      return NVL_EUNSUPPORTED;
  }
#undef NVL_DECODE_CASE
}

}  // namespace

extern "C" int nvl_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache,
                                     const int32_t* block_tables, int64_t bt_stride, const int32_t* context_lens,
                                     void* out, int64_t batch, int num_q_heads, int num_kv_heads, int block_size,
                                     int64_t num_blocks, int64_t max_context, float softmax_scale, void* workspace,
                                     size_t workspace_bytes, int kv_dtype, const void* plan, float* lse,
                                     void* stream) {
  return decode_common(q, const_cast<void*>(k_cache), const_cast<void*>(v_cache), block_tables, bt_stride,
                       context_lens, out, batch, num_q_heads, num_kv_heads, block_size, num_blocks, max_context,
                       softmax_scale, workspace, workspace_bytes, stream, nullptr, "nvl_paged_attn_decode", kv_dtype,
                       plan, lse);
}

extern "C" int nvl_paged_attn_decode_fused(const void* qkv, int64_t qkv_tok_stride, const void* q_norm_w,
                                           const void* k_norm_w, float eps, const float* cos_sin, int64_t max_pos,
                                           void* k_cache, void* v_cache, const int32_t* block_tables,
                                           int64_t bt_stride, const int32_t* context_lens, void* out, int64_t batch,
                                           int num_q_heads, int num_kv_heads, int block_size, int64_t num_blocks,
                                           int64_t max_context, float softmax_scale, void* workspace,
                                           size_t workspace_bytes, int kv_dtype, const void* plan, float* lse,
                                           int qkv_splits, int64_t qkv_split_stride, void* stream) {
  const char* who = "nvl_paged_attn_decode_fused";
  NVL_REQUIRE(qkv_splits >= 0 && qkv_splits <= 8, "%s: qkv_splits=%d (0 = bf16 qkv, 1..8 = fp32 split-K slabs)", who, qkv_splits);
  NVL_REQUIRE(qkv_splits <= 1 || qkv_split_stride >= batch * qkv_tok_stride, "%s: qkv_split_stride=%lld < batch x qkv_tok_stride",
              who, (long long)qkv_split_stride);
  NVL_REQUIRE(qkv_splits == 0 || (int64_t)qkv_splits * qkv_split_stride < (1ll << 31), "%s: qkv slabs of %lld elements exceed 32-bit offsets",
              who, (long long)qkv_split_stride);
  NVL_REQUIRE(cos_sin && max_pos > 0, "%s: rope table required", who);
  NVL_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), "%s: q/k norm weights must both be set or both NULL", who);
  NVL_REQUIRE(qkv_tok_stride % 8 == 0 && qkv_tok_stride >= (int64_t)(num_q_heads + 2 * num_kv_heads) * 128, "%s: bad qkv stride", who);
  NVL_REQUIRE(((uintptr_t)q_norm_w | (uintptr_t)k_norm_w | (uintptr_t)cos_sin) % 16 == 0, "%s: pointers must be 16-byte aligned", who);
  FusedArgs fa;
  fa.qkv_tok_stride = qkv_tok_stride;
  fa.q_norm_w = (const bf16_t*)q_norm_w;
  fa.k_norm_w = (const bf16_t*)k_norm_w;
  fa.cos_sin = cos_sin;
  fa.max_pos = max_pos;
  fa.eps = eps;
  fa.qkv_splits = qkv_splits;
  fa.qkv_split_stride = (int)qkv_split_stride;
  return decode_common(qkv, k_cache, v_cache, block_tables, bt_stride, context_lens, out, batch, num_q_heads,
                       num_kv_heads, block_size, num_blocks, max_context, softmax_scale, workspace, workspace_bytes,
                       stream, &fa, who, kv_dtype, plan, lse);
}

// ---- per-step plan ------------------------------------------------------------------------------------------------
extern "C" size_t nvl_decode_plan_bytes(void) {
  // header + one record per wave of the largest grid the matrix-core kernel launches (2 workgroups per CU)
  return sizeof(PlanHeader) + (size_t)nvl_device_cu_count() * 2 * kWaves * sizeof(PlanEntry);
}

extern "C" int nvl_decode_plan(const int32_t* context_lens, int64_t batch, int num_q_heads, int num_kv_heads,
                               int64_t max_context, const int32_t* shared_prefix, int block_size, int shared_prefix_groups,
                               void* plan, size_t plan_bytes, void* stream) {
  const int32_t* shared_prefix_blocks = shared_prefix;       // [0] = blocks, [1 + b] = member flags (include/nvl.h)
  const char* who = "nvl_decode_plan";
  NVL_REQUIRE(context_lens && plan, "%s: null pointer", who);
  if (shared_prefix_blocks != nullptr) {
    const int G = num_kv_heads > 0 ? num_q_heads / num_kv_heads : 0;
    NVL_REQUIRE((uintptr_t)shared_prefix_blocks % 4 == 0, "%s: shared_prefix must be 4-byte aligned", who);
    NVL_REQUIRE(shared_prefix_groups >= 1 && shared_prefix_groups <= 8, "%s: shared_prefix_groups=%d out of range [1, 8]", who,
                shared_prefix_groups);
    NVL_REQUIRE(block_size > 0 && block_size % 128 == 0, "%s: a shared prefix needs block_size %% 128 == 0 (got %d)", who, block_size);
    NVL_REQUIRE((G == 8 && !use_valu_g8()) || ((G == 2 || G == 4) && use_mfma_small_g()) || (G > 1 && G <= 16 && G != 2 && G != 4 && G != 8),
                "%s: a shared prefix needs the matrix-core decode kernel (Hq/Hkv in 2 ... 16; got %d)", who, G);
  }
  NVL_REQUIRE((uintptr_t)plan % 16 == 0, "%s: plan must be 16-byte aligned", who);
  NVL_REQUIRE(batch >= 0 && batch <= 32768, "%s: batch=%lld out of range [0, 32768]", who, (long long)batch);
  NVL_REQUIRE(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, "%s: Hq=%d not a multiple of Hkv=%d", who, num_q_heads, num_kv_heads);
  NVL_REQUIRE(max_context > 0, "%s: bad max_context", who);
  NVL_REQUIRE(plan_bytes >= nvl_decode_plan_bytes(), "%s: plan buffer %zu B < required %zu B", who, plan_bytes, nvl_decode_plan_bytes());
  if (batch == 0) return NVL_OK;
  int64_t main_blocks = mfma8_grid(batch, num_kv_heads, max_context, true), px_blocks = 0;
  if (shared_prefix_blocks != nullptr)      // the stream-K grid of a shared-prefix launch leaves room for the pack workgroups
    px_split(batch, num_kv_heads, num_q_heads / num_kv_heads, max_context, shared_prefix_groups, &main_blocks, &px_blocks);
  const int nwaves = (int)main_blocks * kWaves;
  const size_t lds = kWaves * sizeof(int) + (size_t)(batch + 2) * sizeof(int);      // wave sums, tile prefix, shortest row
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& attr_set = attr_done[nvl_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      nvl_set_error("%s: cannot reserve LDS", who);
      return NVL_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(decode_plan_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, context_lens, (int)batch,
                     num_kv_heads, nwaves, (PlanHeader*)plan, shared_prefix_blocks,
                     shared_prefix_blocks ? block_size / kTile : 0, shared_prefix_blocks ? shared_prefix_groups : 0);
  const int rc = nvl_check_launch(who);
  if (rc == NVL_OK) plan_shadow_put(plan, batch, num_kv_heads, max_context, shared_prefix_blocks != nullptr ? shared_prefix_groups : 0);   // (a failed launch leaves no record)
  return rc;
}
