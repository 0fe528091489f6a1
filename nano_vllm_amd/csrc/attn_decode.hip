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
// Data path: K and V tiles go HBM -> VGPR directly with non-temporal loads (each byte is used
// exactly once; an LDS round trip would be pure overhead, and `nt` measured +8 % bandwidth). One
// wave instruction fetches 4 token rows x 256 B = 1 KiB contiguous (head-major cache layout),
// 16 instructions (16 KiB) are in flight per wave before the first use. Lane (rq = lane>>4,
// sub = lane&15) holds elements sub*8..sub*8+7 of rows i*4+rq. q.K partial dot products use
// packed v_dot2c_f32_bf16 and are summed over the 16 lanes of a DPP row (row_ror); softmax
// max/sum use wave shuffles; P is rounded to bf16 before P.V (flash-attn convention), fp32
// accumulation. All G = Hq/Hkv query heads of a group are served from one K/V read.
#include "common.h"

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
__device__ __forceinline__ void chunk_prefix(const int32_t* __restrict__ ctx, int batch, int chunk, int* pre,
                                             int* wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;
  if (threadIdx.x == 0) pre[0] = 0;
  for (int base = 0; base < batch; base += 256) {
    const int i = base + threadIdx.x;
    int v = 0;
    if (i < batch) {
      const int len = ctx[i];
      v = len > 0 ? (len + chunk - 1) / chunk : 0;
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

// streamed-once data: non-temporal load (does not displace q / block tables / partials in L2)
__device__ __forceinline__ u32x4_t load16_nt(const bf16_t* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}

template <int G>
__global__ __launch_bounds__(256) void decode_stream_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, const int32_t* __restrict__ ctx,
    float* __restrict__ part_o, float* __restrict__ part_ml, int* __restrict__ meta, int batch, int hkv,
    int block_size, int slots, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* wsum = reinterpret_cast<int*>(smem_raw);
  int* pre = wsum + kWaves;  // tile prefix [batch + 1]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
    const int b = lo;
    const int nb = pre[b + 1] - pre[b];                 // tiles per kv-head of this sequence (> 0 here)
    const int64_t base_b = (int64_t)pre[b] * hkv;
    const int r = (int)(g - base_b);
    const int h = r / nb;
    const int t0 = r - h * nb;
    const int run = (int)min((int64_t)(nb - t0), g1 - g);
    const int len = ctx[b];

    u32x4_t qf[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg)
      qf[gg] = *reinterpret_cast<const u32x4_t*>(q + ((int64_t)b * hq + h * G + gg) * 128 + sub * 8);
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      m[gg] = kNegBig;
      l[gg] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[gg][j] = 0.f;
    }

    for (int ti = t0; ti < t0 + run; ++ti) {
      const int t = ti * kTile;
      const int blk = block_tables[(int64_t)b * bt_stride + t / block_size];
      const int64_t base = (((int64_t)blk * hkv + h) * block_size + (t % block_size)) * 128 + lane * 8;
      const bf16_t* kp = kc + base;
      const bf16_t* vp = vc + base;
      u32x4_t kd[kLoads], vd[kLoads];
#pragma unroll
      for (int i = 0; i < kLoads; ++i) kd[i] = load16_nt(kp + i * 4 * 128);
#pragma unroll
      for (int i = 0; i < kLoads; ++i) vd[i] = load16_nt(vp + i * 4 * 128);
      __builtin_amdgcn_sched_barrier(0);  // all 16 loads in flight before the first use

      float s[G][kLoads];
#pragma unroll
      for (int i = 0; i < kLoads; ++i) {
        const bool valid = (t + i * 4 + rq) < len;
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

__global__ __launch_bounds__(128) void decode_stream_combine_kernel(const float* __restrict__ part_o,
                                                                     const float* __restrict__ part_ml,
                                                                     const int* __restrict__ meta,
                                                                     const int32_t* __restrict__ ctx,
                                                                     bf16_t* __restrict__ out, int hq, int hkv,
                                                                     int slots) {
  // out[b, head, :] = sum_k 2^(m_k - M) O_k / sum_k 2^(m_k - M) l_k over the (b, h) segment's
  // split partials; zero rows when the sequence is padding. Loads are issued independently
  // (one (m, l) pair per thread, then all O rows) so the kernel is ~2 memory latencies long.
  __shared__ float sm[128], sl[128];
  const int b = blockIdx.x / hq;
  const int head = blockIdx.x - b * hq;
  const int64_t row = blockIdx.x;
  const int cnt = ctx[b] > 0 ? meta[b * hkv + head / (hq / hkv)] : 0;
  const int d = threadIdx.x;
  const float* ml = part_ml + row * slots * 2;
  const float* po = part_o + row * slots * 128;
  for (int c = d; c < cnt; c += 128) {   // cnt <= slots (34 at max_context 4096)
    sm[c % 128] = ml[c * 2];
    sl[c % 128] = ml[c * 2 + 1];
  }
  __syncthreads();
  const int n = cnt < 128 ? cnt : 128;
  float M = kNegBig;
  for (int c = 0; c < n; ++c) M = fmaxf(M, sm[c]);
  float num = 0.f, den = 0.f;
  for (int c = 0; c < n; ++c) {
    const float f = exp2f(sm[c] - M);
    num += f * po[c * 128 + d];
    den += f * sl[c];
  }
  out[row * 128 + d] = (bf16_t)(n > 0 ? num / den : 0.f);
}

inline int stream_slots(int64_t max_context) { return (int)(max_context / (kTile * kMinTilesPerWave)) + 2; }

template <int G>
int launch_decode_stream(const void* q, const void* kc, const void* vc, const int32_t* bt, int64_t bt_stride,
                         const int32_t* ctx, void* out, int64_t batch, int hkv, int block_size, int64_t max_context,
                         float scale, void* workspace, hipStream_t s) {
  const int hq = hkv * G;
  const int slots = stream_slots(max_context);
  float* part_o = (float*)workspace;
  float* part_ml = part_o + (size_t)batch * hq * slots * 128;
  int* meta = (int*)(part_ml + (size_t)batch * hq * slots * 2);
  const size_t lds = kWaves * sizeof(int) + (size_t)(batch + 1) * sizeof(int);
  static int cus = 0, per_cu = 0;
  if (cus == 0) {
    cus = nvl_device_cu_count();
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, decode_stream_kernel<G>, 256, lds) != hipSuccess || n < 1) n = 2;
    per_cu = n > 4 ? 4 : n;
  }
  int64_t grid = (int64_t)cus * per_cu;
  const int64_t max_tiles = batch * hkv * ((max_context + kTile - 1) / kTile);
  const int64_t max_wg = (max_tiles + kWaves * kMinTilesPerWave - 1) / (kWaves * kMinTilesPerWave);
  if (grid > max_wg) grid = max_wg;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((decode_stream_kernel<G>), dim3((unsigned)grid), dim3(256), lds, s, (const bf16_t*)q,
                     (const bf16_t*)kc, (const bf16_t*)vc, bt, bt_stride, ctx, part_o, part_ml, meta, (int)batch, hkv,
                     block_size, slots, scale * 1.4426950408889634f);
  hipLaunchKernelGGL(decode_stream_combine_kernel, dim3((unsigned)(batch * hq)), dim3(128), 0, s, part_o, part_ml,
                     meta, ctx, (bf16_t*)out, hq, hkv, slots);
  return nvl_check_launch("nvl_paged_attn_decode");
}

}  // namespace

extern "C" size_t nvl_paged_attn_decode_workspace_bytes(int64_t max_batch, int num_q_heads, int64_t max_context) {
  if (max_batch <= 0 || num_q_heads <= 0 || max_context <= 0) return 0;
  return (size_t)max_batch * num_q_heads * (stream_slots(max_context) * 130 * sizeof(float) + sizeof(int));
}

extern "C" int nvl_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache,
                                     const int32_t* block_tables, int64_t bt_stride, const int32_t* context_lens,
                                     void* out, int64_t batch, int num_q_heads, int num_kv_heads, int block_size,
                                     int64_t num_blocks, int64_t max_context, float softmax_scale, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  NVL_REQUIRE(q && k_cache && v_cache && block_tables && context_lens && out && workspace,
              "nvl_paged_attn_decode: null pointer");
  NVL_REQUIRE(batch >= 0 && batch <= 32768, "nvl_paged_attn_decode: batch=%lld out of range [0, 32768]", (long long)batch);
  NVL_REQUIRE(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, "nvl_paged_attn_decode: Hq=%d not a multiple of Hkv=%d", num_q_heads, num_kv_heads);
  NVL_REQUIRE(block_size > 0 && block_size % kTile == 0, "nvl_paged_attn_decode: block_size=%d must be a multiple of %d", block_size, kTile);
  NVL_REQUIRE(num_blocks > 0 && max_context > 0, "nvl_paged_attn_decode: bad cache geometry");
  NVL_REQUIRE(bt_stride * (int64_t)block_size >= max_context, "nvl_paged_attn_decode: block table (stride %lld) narrower than max_context=%lld", (long long)bt_stride, (long long)max_context);
  NVL_REQUIRE(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out | (uintptr_t)workspace) % 16 == 0,
              "nvl_paged_attn_decode: pointers must be 16-byte aligned");
  if (batch == 0) return NVL_OK;
  const int G = num_q_heads / num_kv_heads;
  const size_t need = nvl_paged_attn_decode_workspace_bytes(batch, num_q_heads, max_context);
  NVL_REQUIRE(workspace_bytes >= need, "nvl_paged_attn_decode: workspace %zu B < required %zu B", workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
#define NVL_DECODE_CASE(GG)                                                                                   \
  case GG:                                                                                                    \
    return launch_decode_stream<GG>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out, batch,  \
                                    num_kv_heads, block_size, max_context, softmax_scale, workspace, s);
  switch (G) {
    NVL_DECODE_CASE(1)
    NVL_DECODE_CASE(2)
    NVL_DECODE_CASE(4)
    NVL_DECODE_CASE(8)
    default:
      nvl_set_error("nvl_paged_attn_decode: unsupported group size Hq/Hkv=%d (supported 1,2,4,8)", G);
      return NVL_EUNSUPPORTED;
  }
#undef NVL_DECODE_CASE
}
