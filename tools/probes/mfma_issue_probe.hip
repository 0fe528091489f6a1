// Probe (round 6): what ONE wave per SIMD sustains on the matrix pipe, measured with a hand-written (inline asm) loop so
// that hipcc's register allocation cannot serialise the accumulators (tools/probes/mfma_lds_rate_probe.hip did exactly that:
// its 16x16x32 loop compiled into a chain of MFMAs whose source / destination accumulator ranges overlap, i.e. it
// measured the DEPENDENT latency and round 5 read it as "one wave issues this shape at half rate").
// Variants: shape 16x16x32 / 32x32x16; with an s_waitcnt between MFMA pairs (the GEMM core's slot shape); with a
// ds_read_b128 per MFMA pair. Prints ns per 16x16x32-EQUIVALENT (16,384 FLOP) per SIMD, shader cycles per MFMA (s_memtime) and
// the effective clock.
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_issue_probe.hip -o /tmp/mfmaissue && /tmp/mfmaissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// VAR 0: 16 x (16x16x32) back to back on 16 accumulators; 1: the same with s_waitcnt lgkmcnt(7) after every 2nd;
// 2: 8 x (32x32x16) on 4 accumulators (2 chains apart); 3: VAR 0 + one ds_read_b128 per 2 MFMAs (ring of 8, waits as the core)
template <int VAR, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(float* sink, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 16384; i += WAVES * 64) reinterpret_cast<unsigned int*>(smem)[i] = 0x3c003c00u;
  __syncthreads();
  f32x4_t a[16];
  f32x16_t b[4];
  for (int i = 0; i < 16; ++i) a[i] = f32x4_t{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) b[i][j] = 0.f;
  u32x4_t f0 = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, f1 = f0;
  u32x4_t x[8];
  for (int i = 0; i < 8; ++i) x[i] = f0;
  const int addr = (threadIdx.x & 63) * 16;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (VAR == 0 || VAR == 1) {
#define M2(i, j) "v_mfma_f32_16x16x32_bf16 %" #i ", %16, %17, %" #i "\n v_mfma_f32_16x16x32_bf16 %" #j ", %16, %17, %" #j "\n"
#define W1 "s_waitcnt lgkmcnt(7)\n"
    for (int it = 0; it < iters; ++it) {
      if constexpr (VAR == 0)
        asm volatile(M2(0, 1) M2(2, 3) M2(4, 5) M2(6, 7) M2(8, 9) M2(10, 11) M2(12, 13) M2(14, 15)
                     : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7]), "+a"(a[8]),
                       "+a"(a[9]), "+a"(a[10]), "+a"(a[11]), "+a"(a[12]), "+a"(a[13]), "+a"(a[14]), "+a"(a[15])
                     : "v"(f0), "v"(f1));
      else
        asm volatile(M2(0, 1) W1 M2(2, 3) W1 M2(4, 5) W1 M2(6, 7) W1 M2(8, 9) W1 M2(10, 11) W1 M2(12, 13) W1 M2(14, 15) W1
                     : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7]), "+a"(a[8]),
                       "+a"(a[9]), "+a"(a[10]), "+a"(a[11]), "+a"(a[12]), "+a"(a[13]), "+a"(a[14]), "+a"(a[15])
                     : "v"(f0), "v"(f1));
    }
  } else if constexpr (VAR == 4) {
    // 64 MFMAs then an s_barrier (the GEMM core's k step); waves >= 4 of a 5- / 6-wave workgroup only take the barriers
    // (the loader waves of the GEMM: they arrive early and wait)
    if (threadIdx.x >= 256) {
      for (int it = 0; it < iters; it += 4) __builtin_amdgcn_s_barrier();
    } else {
      for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          asm volatile(M2(0, 1) M2(2, 3) M2(4, 5) M2(6, 7) M2(8, 9) M2(10, 11) M2(12, 13) M2(14, 15)
                       : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7]), "+a"(a[8]),
                         "+a"(a[9]), "+a"(a[10]), "+a"(a[11]), "+a"(a[12]), "+a"(a[13]), "+a"(a[14]), "+a"(a[15])
                       : "v"(f0), "v"(f1));
        __builtin_amdgcn_s_barrier();
      }
    }
  } else if constexpr (VAR == 2) {
#define B1(i) "v_mfma_f32_32x32x16_bf16 %" #i ", %4, %5, %" #i "\n"
    for (int it = 0; it < iters; ++it)
      asm volatile(B1(0) B1(1) B1(2) B1(3) B1(0) B1(1) B1(2) B1(3)
                   : "+a"(b[0]), "+a"(b[1]), "+a"(b[2]), "+a"(b[3]) : "v"(f0), "v"(f1));
  } else {
#define R1(k) "ds_read_b128 %" #k ", %25 offset:" #k "*1024-16*1024\n s_waitcnt lgkmcnt(7)\n"
#define MX(i, j, k) "v_mfma_f32_16x16x32_bf16 %" #i ", %24, %" #k ", %" #i "\n v_mfma_f32_16x16x32_bf16 %" #j ", %24, %" #k ", %" #j "\n"
    for (int it = 0; it < iters; ++it)
      asm volatile(R1(16) MX(0, 1, 17) R1(17) MX(2, 3, 18) R1(18) MX(4, 5, 19) R1(19) MX(6, 7, 20) R1(20) MX(8, 9, 21) R1(21)
                   MX(10, 11, 22) R1(22) MX(12, 13, 23) R1(23) MX(14, 15, 16)
                   : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]), "+a"(a[7]), "+a"(a[8]),
                     "+a"(a[9]), "+a"(a[10]), "+a"(a[11]), "+a"(a[12]), "+a"(a[13]), "+a"(a[14]), "+a"(a[15]), "+v"(x[0]), "+v"(x[1]),
                     "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                   : "v"(f0), "v"(addr) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i][0];
  for (int i = 0; i < 4; ++i) s += b[i][0];
  for (int i = 0; i < 8; ++i) s += __uint_as_float(x[i][0]);
  if (s == 12345.f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int VAR, int WAVES>
void run(const char* name) {
  float* sink; unsigned long long* cyc;
  hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
  const int iters = 20000, grid = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<VAR, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<VAR, WAVES><<<grid, WAVES * 64, 65536>>>(sink, cyc, 2000);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<VAR, WAVES><<<grid, WAVES * 64, 65536>>>(sink, cyc, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double eq = (double)iters * 16;                    // 16x16x32-equivalents per wave
  const double flops = (double)grid * (WAVES > 4 && VAR == 4 ? 4 : WAVES) * eq * 16384.0;
  printf("{\"case\": \"%s\", \"waves\": %d, \"ms\": %.3f, \"TFLOPs\": %.0f, \"ns_per_eq_mfma_per_wave\": %.2f, \"memtime_ticks_per_eq_mfma\": %.2f, \"ticks_per_us\": %.0f}\n",
         name, WAVES, ms, flops / ms / 1e9, ms * 1e6 / eq, (double)c / eq, (double)c / (ms * 1e3));
  hipFree(sink); hipFree(cyc);
}

int main() {
  run<0, 4>("16x16x32 back to back, 1 wave/SIMD");
  run<0, 3>("16x16x32 back to back, 3 waves/CU");
  run<1, 4>("16x16x32 + s_waitcnt per pair, 1 wave/SIMD");
  run<2, 4>("32x32x16 back to back, 1 wave/SIMD");
  run<3, 4>("16x16x32 + ds_read_b128 per pair (ring 8, lgkmcnt 7), 1 wave/SIMD");
  run<3, 3>("16x16x32 + ds_read_b128 per pair, 3 waves/CU");
  run<4, 4>("64 x 16x16x32 then s_barrier, 4 waves");
  run<4, 5>("64 x 16x16x32 then s_barrier, 4 matrix waves + 1 barrier-only wave");
  run<4, 6>("64 x 16x16x32 then s_barrier, 4 matrix waves + 2 barrier-only waves");
  run<0, 8>("16x16x32 back to back, 2 waves/SIMD");
  run<2, 8>("32x32x16 back to back, 2 waves/SIMD");
  return 0;
}
