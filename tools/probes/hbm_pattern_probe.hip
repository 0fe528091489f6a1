// Probe: HBM streaming rate of a weight matrix W[N][K] (bf16, row-major, K = 4096 => 8 KiB rows) as a function of the
// per-instruction access shape. Models the weight stream of the decode GEMMs (every byte read once, non-temporal).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_pattern_probe.hip -o /tmp/hbmprobe && /tmp/hbmprobe
// MODE 0  fragment: wave instruction = 16 rows x 64 B (MFMA 16x16x32 A-operand shape), 4 instr cover 256 B per row,
//         a wave owns 32 rows (2 tiles) and marches along K in 256-byte steps          [gemm_wide.hip / lmhead.hip]
// MODE 1  fragment, 1 KiB per row per step (16 instr per tile-step)
// MODE 2  packed: the same bytes per wave, but the wave's 256 KiB panel is one contiguous run, 1 KiB per instruction
// MODE 3  row-contiguous: instruction = 4 rows x 256 B (what an LDS-staged W tile load would issue)
// MODE 4  paged tiles: per iteration a wave reads two 8 KiB contiguous tiles (a K and a V tile of 32 tokens) at
//         pseudo-random 8-KiB-aligned places of the buffer                          [the decode-attention stream]
// MODE 5  like 4, but consecutive iterations walk through a 64 KiB block (256-token KV block) before jumping
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int MODE, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void probe(const unsigned char* __restrict__ w, int K_bytes, int rows_per_wave,
                                                      unsigned int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row0 = ((size_t)blockIdx.x * NWAVES + wave) * rows_per_wave;
  const unsigned char* base = w + row0 * K_bytes;
  u32x4_t acc = {0, 0, 0, 0};
  const int l15 = lane & 15, lq = lane >> 4;
  if (MODE == 0 || MODE == 1) {
    constexpr int STEP = MODE == 0 ? 256 : 1024;
    for (int k = 0; k < K_bytes; k += STEP) {
      for (int t = 0; t < rows_per_wave / 16; t += 2) {
        u32x4_t v[2 * STEP / 64];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int i = 0; i < STEP / 64; ++i)
            v[tt * (STEP / 64) + i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(
                base + (size_t)((t + tt) * 16 + l15) * K_bytes + k + i * 64 + lq * 16));
#pragma unroll
        for (int i = 0; i < 2 * STEP / 64; ++i) acc ^= v[i];
      }
    }
  } else if (MODE == 2) {
    const size_t total = (size_t)rows_per_wave * K_bytes;
    for (size_t off = 0; off < total; off += 8192) {
      u32x4_t v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(base + off + i * 1024 + lane * 16));
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= v[i];
    }
  } else if (MODE == 4 || MODE == 5) {
    const size_t total = (size_t)rows_per_wave * K_bytes;
    const size_t span = (size_t)gridDim.x * NWAVES * total;       // whole buffer
    const size_t half = span / 2;
    uint32_t hsh = (uint32_t)(blockIdx.x * NWAVES + wave) * 2654435761u + 12345u;
    size_t blk = 0;
    for (size_t off = 0; off < total; off += 16384) {
      const int it = (int)(off / 16384);
      if (MODE == 4 || (it & 7) == 0) {
        hsh = hsh * 1664525u + 1013904223u;
        blk = ((size_t)(hsh >> 8) % (half / 65536)) * 65536;
      }
      const size_t t8 = MODE == 4 ? (size_t)((hsh >> 4) & 7) * 8192 : (size_t)(it & 7) * 8192;
      u32x4_t v[16];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(w + blk + t8 + i * 1024 + lane * 16));
#pragma unroll
      for (int i = 0; i < 8; ++i)
        v[8 + i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(w + half + blk + t8 + i * 1024 + lane * 16));
#pragma unroll
      for (int i = 0; i < 16; ++i) acc ^= v[i];
    }
  } else {
    for (int k = 0; k < K_bytes; k += 256) {
      for (int t = 0; t < rows_per_wave; t += 32) {
        u32x4_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(
              base + (size_t)(t + i * 4 + lq) * K_bytes + k + l15 * 16));
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= v[i];
      }
    }
  }
  if (acc[0] == 0x12345 && acc[1] == 0x777) sink[0] = acc[2] ^ acc[3];
}

__global__ void fill_random(uint32_t* p, size_t n) {   // constant-filled buffers clock (and stream) higher
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t v = (uint32_t)i * 2654435761u;
    v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
    p[i] = v;
  }
}

template <int MODE, int NWAVES>
void run(const char* name, const unsigned char* w, size_t bytes, int K_bytes, int rows_per_wave, unsigned int* sink) {
  const size_t rows = bytes / K_bytes;
  const int wgs = (int)(rows / ((size_t)NWAVES * rows_per_wave));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, NWAVES>), dim3(wgs), dim3(NWAVES * 64), 0, 0, w, K_bytes, rows_per_wave, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("{\"pattern\": \"%s\", \"waves_per_wg\": %d, \"rows_per_wave\": %d, \"wgs\": %d, \"GBps\": %.0f}\n", name, NWAVES,
         rows_per_wave, wgs, (double)wgs * NWAVES * rows_per_wave * K_bytes / (best * 1e-3) / 1e9);
}

int main() {
  const size_t bytes = (size_t)2 << 30;      // 2 GiB: far beyond L2 + MALL
  const int K_bytes = 8192;
  unsigned char* w;
  unsigned int* sink;
  hipMalloc(&w, bytes);
  hipMalloc(&sink, 4);
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (uint32_t*)w, bytes / 4);
  hipDeviceSynchronize();
  // 256 workgroups x 8 waves, each wave owns 128 rows (1 MiB); and 1024 x 8 x 32 rows
  run<0, 8>("fragment_256B_step", w, bytes, K_bytes, 128, sink);
  run<1, 8>("fragment_1KiB_step", w, bytes, K_bytes, 128, sink);
  run<2, 8>("packed_contiguous", w, bytes, K_bytes, 128, sink);
  run<3, 8>("rows4x256B", w, bytes, K_bytes, 128, sink);
  run<0, 8>("fragment_256B_step", w, bytes, K_bytes, 32, sink);
  run<1, 8>("fragment_1KiB_step", w, bytes, K_bytes, 32, sink);
  run<2, 8>("packed_contiguous", w, bytes, K_bytes, 32, sink);
  run<3, 8>("rows4x256B", w, bytes, K_bytes, 32, sink);
  // the decode-attention launch of the bench: 512 workgroups x 4 waves, ~15 tiles of 16 KiB per wave, 512 MB in total
  run<4, 4>("paged_random_8KiB_tiles_512MB_launch", w, (size_t)512 << 20, K_bytes, 32, sink);
  run<5, 4>("paged_64KiB_blocks_512MB_launch", w, (size_t)512 << 20, K_bytes, 32, sink);
  run<2, 4>("packed_contiguous_512MB_launch", w, (size_t)512 << 20, K_bytes, 32, sink);
  run<4, 4>("paged_random_8KiB_tiles", w, bytes, K_bytes, 128, sink);
  run<5, 4>("paged_64KiB_blocks", w, bytes, K_bytes, 128, sink);
  run<4, 8>("paged_random_8KiB_tiles", w, bytes, K_bytes, 64, sink);
  run<5, 8>("paged_64KiB_blocks", w, bytes, K_bytes, 64, sink);
  run<2, 4>("packed_contiguous", w, bytes, K_bytes, 128, sink);
  run<0, 4>("fragment_256B_step", w, bytes, K_bytes, 32, sink);
  run<2, 4>("packed_contiguous", w, bytes, K_bytes, 32, sink);
  run<3, 4>("rows4x256B", w, bytes, K_bytes, 32, sink);
  return 0;
}
