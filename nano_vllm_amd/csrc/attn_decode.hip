// Paged single-query (decode) attention for gfx950 — the HBM-roofline kernel of the path.
// Replaces flash_attn_with_kvcache as used by nano-vllm layers/attention.py:72-74.
//
// Work decomposition (persistent, hipGraph-safe): the per-step work list is derived ON DEVICE
// from context_lens — every workgroup prefix-sums ceil(len_b / chunk) over the batch into LDS,
// then walks items  (sequence b, chunk c, kv-head h)  with a static stride. The grid is a
// launch-time constant, so the same captured launch serves any mix of lengths; padded rows
// (context_len 0) simply contribute no items.
//
// Data path: K and V tiles go HBM -> VGPR directly (each byte is used once; an LDS round trip
// would be pure overhead). One wave instruction fetches 4 token rows x 256 B = 1 KiB contiguous
// (head-major cache layout). Lane (rq = lane>>4, sub = lane&15) holds elements sub*8..sub*8+7
// of rows i*4+rq. q.K partial dot products use packed v_dot2_f32_bf16 and are summed over the
// 16 lanes of a DPP row; softmax max/sum use wave shuffles; P is rounded to bf16 before P.V
// (flash-attn convention), accumulation in fp32. All G = Hq/Hkv query heads of a group are
// served from one K/V read. Split-KV partials (m, l, O) go to a workspace and are merged by a
// second tiny kernel.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kWaves = 4;
constexpr int kTile = 32;                 // tokens per wave per step
constexpr int kStep = kWaves * kTile;     // tokens per workgroup step
constexpr int kLoads = kTile / 4;         // 16-byte loads per lane per tile (K or V)
constexpr float kNegBig = -1.0e30f;
constexpr int kMinChunk = 128;

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

template <int G>
__global__ __launch_bounds__(256) void decode_attn_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, const int32_t* __restrict__ ctx,
    float* __restrict__ part_o, float* __restrict__ part_ml, int batch, int hkv, int block_size, int chunk,
    int max_chunks, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // LDS carve: merge O [kWaves][G][128] f32 | merge ml [kWaves][G][2] f32 | wsum[4] | pre[batch+1]
  float* mo = reinterpret_cast<float*>(smem_raw);
  float* mml = mo + kWaves * G * 128;
  int* wsum = reinterpret_cast<int*>(mml + kWaves * G * 2);
  int* pre = wsum + kWaves;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 15, rq = lane >> 4;
  const int hq = hkv * G;

  chunk_prefix(ctx, batch, chunk, pre, wsum);
  __syncthreads();
  const int64_t total_items = (int64_t)pre[batch] * hkv;

  for (int64_t item = blockIdx.x; item < total_items; item += gridDim.x) {
    const int pair = (int)(item / hkv);
    const int h = (int)(item - (int64_t)pair * hkv);
    int lo = 0, hi = batch;  // largest b with pre[b] <= pair
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= pair) lo = mid; else hi = mid;
    }
    const int b = lo;
    const int c = pair - pre[b];
    const int len = ctx[b];
    const int tok0 = c * chunk;
    const int tok_end = min(len, tok0 + chunk);

    u32x4_t qf[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
      qf[g] = *reinterpret_cast<const u32x4_t*>(q + ((int64_t)b * hq + h * G + g) * 128 + sub * 8);

    float m[G], l[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      m[g] = kNegBig;
      l[g] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[g][j] = 0.f;
    }

    for (int t = tok0 + wave * kTile; t < tok_end; t += kStep) {
      const int blk = block_tables[(int64_t)b * bt_stride + t / block_size];
      const int64_t base = (((int64_t)blk * hkv + h) * block_size + (t % block_size)) * 128 + lane * 8;
      const bf16_t* kp = kc + base;
      const bf16_t* vp = vc + base;
      u32x4_t kd[kLoads], vd[kLoads];
#pragma unroll
      for (int i = 0; i < kLoads; ++i) kd[i] = *reinterpret_cast<const u32x4_t*>(kp + i * 4 * 128);
#pragma unroll
      for (int i = 0; i < kLoads; ++i) vd[i] = *reinterpret_cast<const u32x4_t*>(vp + i * 4 * 128);
      // keep all 16 loads (16 KiB per wave) in flight before the first use: without this fence
      // hipcc sinks the K loads between the dot products and serialises their latencies
      __builtin_amdgcn_sched_barrier(0);

      float s[G][kLoads];
#pragma unroll
      for (int i = 0; i < kLoads; ++i) {
        const bool valid = (t + i * 4 + rq) < len;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float d = row16_allreduce_sum(dot8(kd[i], qf[g]));
          s[g][i] = valid ? d * scale_log2e : kNegBig;
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float mx = s[g][0];
#pragma unroll
        for (int i = 1; i < kLoads; ++i) mx = fmaxf(mx, s[g][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m[g], mx);
        const float alpha = exp2f(m[g] - mn);
        m[g] = mn;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
          const float p = exp2f(s[g][i] - mn);
          psum += p;
          s[g][i] = round_bf16(p);
        }
        l[g] = l[g] * alpha + psum;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[g][j] *= alpha;
      }
#pragma unroll
      for (int i = 0; i < kLoads; ++i) {
        float vf[8];
        unpack8(vd[i], vf);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int j = 0; j < 8; ++j) o[g][j] = fmaf(s[g][i], vf[j], o[g][j]);
      }
    }

    // fold the 4 row-groups of the wave
#pragma unroll
    for (int g = 0; g < G; ++g) {
      l[g] += __shfl_xor(l[g], 16, 64);
      l[g] += __shfl_xor(l[g], 32, 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[g][j] += __shfl_xor(o[g][j], 16, 64);
        o[g][j] += __shfl_xor(o[g][j], 32, 64);
      }
    }
    if (rq == 0) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float* dst = mo + (wave * G + g) * 128 + sub * 8;
        *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{o[g][0], o[g][1], o[g][2], o[g][3]};
        *reinterpret_cast<f32x4_t*>(dst + 4) = f32x4_t{o[g][4], o[g][5], o[g][6], o[g][7]};
        if (sub == 0) {
          mml[(wave * G + g) * 2] = m[g];
          mml[(wave * G + g) * 2 + 1] = l[g];
        }
      }
    }
    __syncthreads();
    // merge the 4 waves and emit the split partial for (b, q-head, chunk c)
    for (int idx = threadIdx.x; idx < G * 128; idx += 256) {
      const int g = idx >> 7, d = idx & 127;
      float mw[kWaves], M = kNegBig;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        mw[w] = mml[(w * G + g) * 2];
        M = fmaxf(M, mw[w]);
      }
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const float f = exp2f(mw[w] - M);
        num += f * mo[(w * G + g) * 128 + d];
        den += f * mml[(w * G + g) * 2 + 1];
      }
      const int64_t pidx = ((int64_t)b * hq + h * G + g) * max_chunks + c;
      part_o[pidx * 128 + d] = num;
      if (d == 0) {
        part_ml[pidx * 2] = M;
        part_ml[pidx * 2 + 1] = den;
      }
    }
    __syncthreads();
  }
}

// out[b, hq, :] = sum_c 2^(m_c - M) O_c / sum_c 2^(m_c - M) l_c ; zero rows when no chunks.
__global__ __launch_bounds__(128) void decode_combine_kernel(const float* __restrict__ part_o,
                                                              const float* __restrict__ part_ml,
                                                              const int32_t* __restrict__ ctx,
                                                              bf16_t* __restrict__ out, int hq, int chunk,
                                                              int max_chunks) {
  const int b = blockIdx.x / hq;
  const int64_t row = blockIdx.x;  // b*hq + head
  const int len = ctx[b];
  const int nch = len > 0 ? (len + chunk - 1) / chunk : 0;
  const int d = threadIdx.x;
  const float* ml = part_ml + row * max_chunks * 2;
  const float* po = part_o + row * max_chunks * 128;
  float M = kNegBig;
  for (int c = 0; c < nch; ++c) M = fmaxf(M, ml[c * 2]);
  float num = 0.f, den = 0.f;
  for (int c = 0; c < nch; ++c) {
    const float f = exp2f(ml[c * 2] - M);
    num += f * po[c * 128 + d];
    den += f * ml[c * 2 + 1];
  }
  const float r = nch > 0 ? num / den : 0.f;
  out[row * 128 + d] = (bf16_t)r;
}

// ================================================================================================
// Stream variant: every WAVE is an independent worker over an equal share of the flattened
// (sequence, kv-head, 32-token tile) space — no workgroup barriers or LDS merge in the loop, and
// load balance to within one 16 KiB tile regardless of how ragged the context lengths are.
// A wave's share may cover the tail of one (b, h), several whole short ones and the head of
// another; it emits one split partial per (b, h) segment into slot k = wave - first_wave(b, h).
constexpr int kMinTilesPerWave = 4;

template <bool NT>
__device__ __forceinline__ u32x4_t load16(const bf16_t* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  else return *reinterpret_cast<const u32x4_t*>(p);
}

template <int G, bool NT, int OCC>
__global__ __launch_bounds__(256, OCC) void decode_stream_kernel(
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
      for (int i = 0; i < kLoads; ++i) kd[i] = load16<NT>(kp + i * 4 * 128);
#pragma unroll
      for (int i = 0; i < kLoads; ++i) vd[i] = load16<NT>(vp + i * 4 * 128);
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
  const int b = blockIdx.x / hq;
  const int head = blockIdx.x - b * hq;
  const int64_t row = blockIdx.x;
  const int cnt = ctx[b] > 0 ? meta[b * hkv + head / (hq / hkv)] : 0;
  const int d = threadIdx.x;
  const float* ml = part_ml + row * slots * 2;
  const float* po = part_o + row * slots * 128;
  float M = kNegBig;
  for (int c = 0; c < cnt; ++c) M = fmaxf(M, ml[c * 2]);
  float num = 0.f, den = 0.f;
  for (int c = 0; c < cnt; ++c) {
    const float f = exp2f(ml[c * 2] - M);
    num += f * po[c * 128 + d];
    den += f * ml[c * 2 + 1];
  }
  out[row * 128 + d] = (bf16_t)(cnt > 0 ? num / den : 0.f);
}

inline int stream_slots(int64_t max_context) { return (int)(max_context / (kTile * kMinTilesPerWave)) + 2; }

template <int G, bool NT, int OCC>
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
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, decode_stream_kernel<G, NT, OCC>, 256, lds) != hipSuccess || n < 1) n = 2;
    per_cu = n > 4 ? 4 : n;
  }
  int64_t grid = (int64_t)cus * per_cu;
  const int64_t max_tiles = batch * hkv * ((max_context + kTile - 1) / kTile);
  const int64_t max_wg = (max_tiles + kWaves * kMinTilesPerWave - 1) / (kWaves * kMinTilesPerWave);
  if (grid > max_wg) grid = max_wg;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((decode_stream_kernel<G, NT, OCC>), dim3((unsigned)grid), dim3(256), lds, s, (const bf16_t*)q,
                     (const bf16_t*)kc, (const bf16_t*)vc, bt, bt_stride, ctx, part_o, part_ml, meta, (int)batch, hkv,
                     block_size, slots, scale * 1.4426950408889634f);
  hipLaunchKernelGGL(decode_stream_combine_kernel, dim3((unsigned)(batch * hq)), dim3(128), 0, s, part_o, part_ml,
                     meta, ctx, (bf16_t*)out, hq, hkv, slots);
  return nvl_check_launch("nvl_paged_attn_decode");
}

inline int pick_chunk(int64_t batch, int hkv, int block_size) {
  (void)block_size;
  return (batch * hkv >= 1024) ? 256 : kMinChunk;
}

template <int G>
int launch_decode(const void* q, const void* kc, const void* vc, const int32_t* bt, int64_t bt_stride,
                  const int32_t* ctx, void* out, int64_t batch, int hkv, int block_size, int64_t max_context,
                  float scale, float* part_o, float* part_ml, int max_chunks, int chunk, hipStream_t s) {
  const int hq = hkv * G;
  const size_t lds = (size_t)kWaves * G * 130 * sizeof(float) + kWaves * sizeof(int) + (size_t)(batch + 1) * sizeof(int);
  // persistent grid = CUs x resident workgroups per CU (register-limited; queried once per G)
  static int cus = 0, per_cu = 0;
  if (cus == 0) {
    cus = nvl_device_cu_count();
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, decode_attn_kernel<G>, 256, lds) != hipSuccess || n < 1) n = 2;
    per_cu = n > 4 ? 4 : n;
  }
  int64_t grid = (int64_t)cus * per_cu;
  const int64_t max_items = batch * hkv * ((max_context + chunk - 1) / chunk);
  if (grid > max_items) grid = max_items;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(decode_attn_kernel<G>, dim3((unsigned)grid), dim3(256), lds, s, (const bf16_t*)q,
                     (const bf16_t*)kc, (const bf16_t*)vc, bt, bt_stride, ctx, part_o, part_ml, (int)batch, hkv,
                     block_size, chunk, max_chunks, scale * 1.4426950408889634f);
  hipLaunchKernelGGL(decode_combine_kernel, dim3((unsigned)(batch * hq)), dim3(128), 0, s, part_o, part_ml, ctx,
                     (bf16_t*)out, hq, chunk, max_chunks);
  return nvl_check_launch("nvl_paged_attn_decode");
}

}  // namespace

extern "C" size_t nvl_paged_attn_decode_workspace_bytes(int64_t max_batch, int num_q_heads, int64_t max_context) {
  if (max_batch <= 0 || num_q_heads <= 0 || max_context <= 0) return 0;
  const int64_t max_chunks = (max_context + kMinChunk - 1) / kMinChunk;
  const size_t a = (size_t)max_batch * num_q_heads * max_chunks * 130 * sizeof(float);
  const size_t b = (size_t)max_batch * num_q_heads * (stream_slots(max_context) * 130 * sizeof(float) + sizeof(int));
  return a > b ? a : b;
}

static int decode_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("NVL_DECODE_VARIANT");
    v = e ? atoi(e) : 0;
  }
  return v;
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
  NVL_REQUIRE(block_size > 0 && block_size % kStep == 0, "nvl_paged_attn_decode: block_size=%d must be a multiple of %d", block_size, kStep);
  NVL_REQUIRE(num_blocks > 0 && max_context > 0, "nvl_paged_attn_decode: bad cache geometry");
  NVL_REQUIRE(bt_stride * (int64_t)block_size >= max_context, "nvl_paged_attn_decode: block table (stride %lld) narrower than max_context=%lld", (long long)bt_stride, (long long)max_context);
  NVL_REQUIRE(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out | (uintptr_t)workspace) % 16 == 0,
              "nvl_paged_attn_decode: pointers must be 16-byte aligned");
  if (batch == 0) return NVL_OK;
  const int G = num_q_heads / num_kv_heads;
  const int chunk = pick_chunk(batch, num_kv_heads, block_size);
  const int64_t max_chunks = (max_context + chunk - 1) / chunk;
  const size_t need = (size_t)batch * num_q_heads * max_chunks * 130 * sizeof(float);
  NVL_REQUIRE(workspace_bytes >= need, "nvl_paged_attn_decode: workspace %zu B < required %zu B", workspace_bytes, need);
  float* part_o = (float*)workspace;
  float* part_ml = part_o + (size_t)batch * num_q_heads * max_chunks * 128;
  hipStream_t s = (hipStream_t)stream;
  const int variant = decode_variant();
  if (variant >= 1) {
    NVL_REQUIRE(workspace_bytes >= nvl_paged_attn_decode_workspace_bytes(batch, num_q_heads, max_context),
                "nvl_paged_attn_decode: workspace too small for the stream variant");
#define NVL_STREAM_ARGS q, k_cache, v_cache, block_tables, bt_stride, context_lens, out, batch, num_kv_heads, block_size, max_context, softmax_scale, workspace, s
#define NVL_STREAM_CASE(GG)                                                        \
  case GG:                                                                         \
    switch (variant) {                                                             \
      case 2: return launch_decode_stream<GG, true, 1>(NVL_STREAM_ARGS);           \
      case 3: return launch_decode_stream<GG, false, (GG <= 2 ? 4 : 1)>(NVL_STREAM_ARGS); \
      case 4: return launch_decode_stream<GG, true, (GG <= 2 ? 4 : 1)>(NVL_STREAM_ARGS);  \
      default: return launch_decode_stream<GG, false, 1>(NVL_STREAM_ARGS);         \
    }
    switch (G) {
      NVL_STREAM_CASE(1)
      NVL_STREAM_CASE(2)
      NVL_STREAM_CASE(4)
      NVL_STREAM_CASE(8)
      default:
        nvl_set_error("nvl_paged_attn_decode: unsupported group size Hq/Hkv=%d (supported 1,2,4,8)", G);
        return NVL_EUNSUPPORTED;
    }
#undef NVL_STREAM_CASE
  }
#define NVL_DECODE_CASE(GG)                                                                                     \
  case GG:                                                                                                      \
    return launch_decode<GG>(q, k_cache, v_cache, block_tables, bt_stride, context_lens, out, batch,           \
                             num_kv_heads, block_size, max_context, softmax_scale, part_o, part_ml,            \
                             (int)max_chunks, chunk, s);
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
