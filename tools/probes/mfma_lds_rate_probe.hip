// Probe: what does a CU sustain on v_mfma_f32_16x16x32_bf16 with 8 waves (2 per SIMD) — alone, and with the decode GEMM's
// fragment traffic beside it (R ds_read_b128 per 16 MFMAs per wave)? Prints TFLOP/s and cycles per MFMA per SIMD.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_rate_probe.hip -o /tmp/mfmaprobe && /tmp/mfmaprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int READS, int WAVES, bool BIG>
__global__ __launch_bounds__(WAVES * 64) void probe(float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += WAVES * 64) reinterpret_cast<unsigned int*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  f32x4_t acc[16];
  f32x16_t accb[4];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) accb[i][j] = 0.f;
  u32x4_t f[10];
  for (int i = 0; i < 10; ++i) f[i] = *reinterpret_cast<const u32x4_t*>(smem + i * 1024 + lane * 16);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < READS; ++i) f[i] = *reinterpret_cast<const u32x4_t*>(smem + ((it + i) & 15) * 4096 + i * 1024 + lane * 16);
    if constexpr (!BIG) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f[i % 2]), __builtin_bit_cast(bf16x8_t, f[2 + i / 2]), acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)     // 8 x (32x32x16) = the flops of 16 x (16x16x32)
        accb[i % 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f[i % 2]), __builtin_bit_cast(bf16x8_t, f[2 + i]), accb[i % 4], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  for (int i = 0; i < 4; ++i) s += accb[i][0];
  if (s == 12345.f) sink[0] = s;
}

template <int READS, int WAVES, bool BIG>
void run(const char* name) {
  float* sink;
  hipMalloc(&sink, 4);
  const int iters = 20000, grid = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<READS, WAVES, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  probe<READS, WAVES, BIG><<<grid, WAVES * 64, 65536>>>(sink, 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<READS, WAVES, BIG><<<grid, WAVES * 64, 65536>>>(sink, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * WAVES * iters * 16 * 16384.0;
  const double per_simd = (double)WAVES / 4 * iters * 16;          // 16x16x32-equivalent MFMAs per SIMD
  printf("{\"case\": \"%s\", \"reads_per_16_mfma\": %d, \"waves\": %d, \"ms\": %.3f, \"TFLOPs\": %.0f, \"ns_per_mfma_per_simd\": %.2f}\n", name, READS,
         WAVES, ms, flops / ms / 1e9, ms * 1e6 / per_simd);
  hipFree(sink);
}

int main() {
  run<0, 8, false>("16x16x32 mfma only, 8 waves");
  run<0, 4, false>("16x16x32 mfma only, 4 waves");
  run<5, 8, false>("16x16x32 + 5 ds_read_b128 per 16 mfma");
  run<10, 8, false>("16x16x32 + 10 ds_read_b128 per 16 mfma");
  run<10, 4, false>("16x16x32 + 10 reads, 4 waves");
  run<0, 8, true>("32x32x16 mfma only, 8 waves");
  run<10, 8, true>("32x32x16 + 10 ds_read_b128 per 8 mfma");
  run<5, 8, true>("32x32x16 + 5 ds_read_b128 per 8 mfma");
  run<0, 4, true>("32x32x16 mfma only, 4 waves (one per SIMD)");
  run<5, 4, true>("32x32x16 + 5 reads per 8 mfma, 4 waves");
  run<10, 4, true>("32x32x16 + 10 reads per 8 mfma, 4 waves");
  run<0, 3, true>("32x32x16 mfma only, 3 waves");
  run<0, 3, false>("16x16x32 mfma only, 3 waves");
  return 0;
}
