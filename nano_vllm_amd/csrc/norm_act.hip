// RMSNorm, fused residual-add RMSNorm and SiLU*mul for gfx950.
// HBM-bound elementwise/reduction kernels: 16-byte (8 x bf16) loads per lane, rows kept
// in registers (single pass), wave-shuffle + LDS reductions.
// Reference semantics: nano-vllm layers/layernorm.py:16-40, layers/activation.py:8-11
// (the @torch.compile'd graphs: fp32 math, ONE rounding to bf16 at the end).
#include "common.h"

namespace {

constexpr int kMaxChunks = 4;  // 8-element chunks held per thread

// Block-wide sum. T threads (T = 64: one wave, or 256: four waves).
template <int T>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_allreduce_sum(v);
  if constexpr (T > NVL_WAVE) {
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < T / NVL_WAVE; ++w) t += red[w];
    v = t;
  }
  return v;
}

// One workgroup of T threads per row. hidden % 8 == 0, hidden <= T*8*kMaxChunks.
// ADD: s = x + residual; residual <- bf16(s); normalise the un-rounded s.
template <int T, bool ADD>
__global__ __launch_bounds__(T) void rmsnorm_kernel(const bf16_t* __restrict__ x, int64_t x_outer_stride,
                                                     bf16_t* __restrict__ residual,
                                                     const bf16_t* __restrict__ weight,
                                                     bf16_t* __restrict__ y, int64_t y_outer_stride,
                                                     int n_inner, int hidden, float eps) {
  __shared__ float red[T / NVL_WAVE];
  const int64_t row = blockIdx.x;
  const int64_t outer = row / n_inner;
  const int inner = (int)(row - outer * n_inner);
  const bf16_t* xr = x + outer * x_outer_stride + (int64_t)inner * hidden;
  bf16_t* yr = y + outer * y_outer_stride + (int64_t)inner * hidden;
  bf16_t* rr = ADD ? residual + row * (int64_t)hidden : nullptr;
  const int nchunks = hidden >> 3;

  float v[kMaxChunks][8];
  u32x4_t wraw[kMaxChunks];   // weights fetched up front: keeps them off the post-reduction critical path
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int chunk = threadIdx.x + c * T;
    if (chunk < nchunks) wraw[c] = *reinterpret_cast<const u32x4_t*>(weight + chunk * 8);
  }
#pragma unroll
  for (int c = 0; c < kMaxChunks; ++c) {
    const int chunk = threadIdx.x + c * T;
    if (chunk < nchunks) {
      u32x4_t w = *reinterpret_cast<const u32x4_t*>(xr + chunk * 8);
      unpack8(w, v[c]);
      if constexpr (ADD) {
        u32x4_t rw = *reinterpret_cast<const u32x4_t*>(rr + chunk * 8);
        float r[8];
        unpack8(rw, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] += r[i];
        *reinterpret_cast<u32x4_t*>(rr + chunk * 8) = pack8(v[c]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += v[c][i] * v[c][i];
    }
  }
  ss = block_sum<T>(ss, red);
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

// head_dim-sized rows (hidden == 128): 16 lanes per row, 4 rows per wave, 16 per block.
template <bool DUMMY>
__global__ __launch_bounds__(256) void rmsnorm_d128_kernel(const bf16_t* __restrict__ x, int64_t x_outer_stride,
                                                            const bf16_t* __restrict__ weight,
                                                            bf16_t* __restrict__ y, int64_t y_outer_stride,
                                                            int64_t n_rows, int n_inner, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const bool live = row < n_rows;
  const int64_t rclamp = live ? row : 0;
  const int64_t outer = rclamp / n_inner;
  const int inner = (int)(rclamp - outer * n_inner);
  const bf16_t* xr = x + outer * x_outer_stride + (int64_t)inner * 128 + sub * 8;
  float v[8], wf[8], o[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(xr), v);
  unpack8(*reinterpret_cast<const u32x4_t*>(weight + sub * 8), wf);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
  ss = row16_allreduce_sum(ss);
  const float rstd = rsqrtf(ss * (1.f / 128.f) + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = v[i] * rstd * wf[i];
  if (live) *reinterpret_cast<u32x4_t*>(y + outer * y_outer_stride + (int64_t)inner * 128 + sub * 8) = pack8(o);
}

__device__ __forceinline__ float silu_f32(float g) { return g / (1.f + __expf(-g)); }

// y[r, i] = bf16(silu(x[r, i]) * x[r, inter + i]); 8 elements per thread, grid-stride.
__global__ __launch_bounds__(256) void silu_mul_kernel(const bf16_t* __restrict__ x, int64_t x_row_stride,
                                                        bf16_t* __restrict__ y, int64_t rows, int inter) {
  const int chunks_per_row = inter >> 3;
  const int64_t total = rows * chunks_per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / chunks_per_row;
    const int c = (int)(idx - r * chunks_per_row);
    const bf16_t* xr = x + r * x_row_stride + c * 8;
    float g[8], u[8], o[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(xr), g);
    unpack8(*reinterpret_cast<const u32x4_t*>(xr + inter), u);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = silu_f32(g[i]) * u[i];
    *reinterpret_cast<u32x4_t*>(y + r * (int64_t)inter + c * 8) = pack8(o);
  }
}

}  // namespace

extern "C" int nvl_rmsnorm(const void* x, int64_t x_outer_stride, const void* weight, void* y,
                           int64_t y_outer_stride, int64_t n_outer, int n_inner, int hidden, float eps,
                           void* stream) {
  NVL_REQUIRE(x && weight && y, "nvl_rmsnorm: null pointer");
  NVL_REQUIRE(n_outer >= 0 && n_inner > 0, "nvl_rmsnorm: bad row counts (%lld, %d)", (long long)n_outer, n_inner);
  NVL_REQUIRE(hidden > 0 && hidden % 8 == 0 && hidden <= 256 * 8 * kMaxChunks,
              "nvl_rmsnorm: hidden=%d must be a multiple of 8 and <= %d", hidden, 256 * 8 * kMaxChunks);
  NVL_REQUIRE(x_outer_stride % 8 == 0 && y_outer_stride % 8 == 0, "nvl_rmsnorm: strides must be multiples of 8");
  NVL_REQUIRE(((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight) % 16 == 0, "nvl_rmsnorm: pointers must be 16-byte aligned");
  const int64_t rows = n_outer * n_inner;
  if (rows == 0) return NVL_OK;
  NVL_REQUIRE(rows < (1ll << 31), "nvl_rmsnorm: too many rows");
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* xp = (const bf16_t*)x;
  const bf16_t* wp = (const bf16_t*)weight;
  bf16_t* yp = (bf16_t*)y;
  if (hidden == 128) {
    hipLaunchKernelGGL(rmsnorm_d128_kernel<true>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, xp,
                       x_outer_stride, wp, yp, y_outer_stride, rows, n_inner, eps);
  } else if (hidden <= 64 * 8 * kMaxChunks) {
    hipLaunchKernelGGL((rmsnorm_kernel<64, false>), dim3((unsigned)rows), dim3(64), 0, s, xp, x_outer_stride,
                       (bf16_t*)nullptr, wp, yp, y_outer_stride, n_inner, hidden, eps);
  } else {
    hipLaunchKernelGGL((rmsnorm_kernel<256, false>), dim3((unsigned)rows), dim3(256), 0, s, xp, x_outer_stride,
                       (bf16_t*)nullptr, wp, yp, y_outer_stride, n_inner, hidden, eps);
  }
  return nvl_check_launch("nvl_rmsnorm");
}

extern "C" int nvl_add_rmsnorm(const void* x, void* residual, const void* weight, void* y, int64_t rows,
                               int hidden, float eps, void* stream) {
  NVL_REQUIRE(x && residual && weight && y, "nvl_add_rmsnorm: null pointer");
  NVL_REQUIRE(rows >= 0 && rows < (1ll << 31), "nvl_add_rmsnorm: bad rows=%lld", (long long)rows);
  NVL_REQUIRE(hidden > 0 && hidden % 8 == 0 && hidden <= 256 * 8 * kMaxChunks,
              "nvl_add_rmsnorm: hidden=%d must be a multiple of 8 and <= %d", hidden, 256 * 8 * kMaxChunks);
  NVL_REQUIRE(((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)residual) % 16 == 0,
              "nvl_add_rmsnorm: pointers must be 16-byte aligned");
  if (rows == 0) return NVL_OK;
  hipStream_t s = (hipStream_t)stream;
  if (hidden <= 64 * 8 * kMaxChunks) {
    hipLaunchKernelGGL((rmsnorm_kernel<64, true>), dim3((unsigned)rows), dim3(64), 0, s, (const bf16_t*)x,
                       (int64_t)hidden, (bf16_t*)residual, (const bf16_t*)weight, (bf16_t*)y, (int64_t)hidden, 1,
                       hidden, eps);
  } else {
    hipLaunchKernelGGL((rmsnorm_kernel<256, true>), dim3((unsigned)rows), dim3(256), 0, s, (const bf16_t*)x,
                       (int64_t)hidden, (bf16_t*)residual, (const bf16_t*)weight, (bf16_t*)y, (int64_t)hidden, 1,
                       hidden, eps);
  }
  return nvl_check_launch("nvl_add_rmsnorm");
}

extern "C" int nvl_silu_mul(const void* x, int64_t x_row_stride, void* y, int64_t rows, int inter, void* stream) {
  NVL_REQUIRE(x && y, "nvl_silu_mul: null pointer");
  NVL_REQUIRE(rows >= 0 && inter > 0 && inter % 8 == 0, "nvl_silu_mul: inter=%d must be a positive multiple of 8", inter);
  NVL_REQUIRE(x_row_stride % 8 == 0 && x_row_stride >= 2ll * inter, "nvl_silu_mul: bad row stride");
  NVL_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "nvl_silu_mul: pointers must be 16-byte aligned");
  if (rows == 0) return NVL_OK;
  const int64_t total = rows * (inter >> 3);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     x_row_stride, (bf16_t*)y, rows, inter);
  return nvl_check_launch("nvl_silu_mul");
}
