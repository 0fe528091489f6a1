// Probe: how fast can every CU re-read the SAME small (L2-resident) buffer?  Models the x-operand
// traffic of a skinny GEMM (M x K bf16 re-read by every workgroup).  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l2_read_probe.hip -o /tmp/l2probe && /tmp/l2probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// MODE 0: fragment-shaped: lane (r = l&15, q = l>>4) reads 16 B at row r, byte (kb*64 + q*16)   [16 rows x 64 B / instr]
// MODE 1: full-line: lane l reads 16 B at byte l*16 of a 1 KiB contiguous run                      [1 KiB / instr]
// MODE 2: 2 rows x 512 B per instruction (what a K-slice-per-wave LDS staging pass would issue)
template <int MODE, bool ROT>
__global__ __launch_bounds__(256) void probe(const unsigned char* __restrict__ x, int rows, int row_bytes, int iters,
                                             unsigned int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4_t acc = {0, 0, 0, 0};
  const int slice = row_bytes / 4;           // bytes of a row owned by this wave
  const int tiles = rows / 16;
  const int rot = ROT ? (blockIdx.x % tiles) : 0;
  for (int it = 0; it < iters; ++it) {
    for (int t0 = 0; t0 < tiles; ++t0) {
      int t = t0 + rot; if (t >= tiles) t -= tiles;
      u32x4_t v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        size_t off;
        if (MODE == 0) off = (size_t)(t * 16 + (lane & 15)) * row_bytes + wave * slice + i * 64 + (lane >> 4) * 16;
        else if (MODE == 1) off = (size_t)(t * 16) * row_bytes + (size_t)(wave * 8 + i) * 1024 + lane * 16;   // contiguous 32 KiB tile
        else off = (size_t)(t * 16 + i * 2 + (lane >> 5)) * row_bytes + wave * slice + (lane & 31) * 16;
        v[i] = *reinterpret_cast<const u32x4_t*>(x + off);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= v[i];
    }
  }
  if (acc[0] == 0x12345 && acc[1] == 0x777) sink[0] = acc[2] ^ acc[3];
}

template <int MODE, bool ROT>
void run(const char* name, const unsigned char* x, int rows, int row_bytes, int grid, unsigned int* sink) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20;
  probe<MODE, ROT><<<grid, 256>>>(x, rows, row_bytes, 2, sink);
  hipEventRecord(a);
  probe<MODE, ROT><<<grid, 256>>>(x, rows, row_bytes, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)grid * iters * rows * row_bytes;
  double per_cu = bytes / 256.0 / (ms * 1e-3) / 2.4e9;
  printf("%-34s grid %4d rows %3d: %7.2f TB/s aggregate, %5.1f B/clk/CU (2.4 GHz)  %.1f us per pass\n", name, grid, rows,
         bytes / (ms * 1e-3) / 1e12, per_cu, ms * 1e3 / iters);
}

int main() {
  const int row_bytes = 2048;     // K = 1024 bf16
  unsigned char* x; unsigned int* sink;
  hipMalloc(&x, 512 * row_bytes); hipMemset(x, 1, 512 * row_bytes); hipMalloc(&sink, 64);
  for (int rows : {144, 256}) {
    for (int grid : {256, 512, 1024}) {
      run<0, false>("fragment 16rows x 64B", x, rows, row_bytes, grid, sink);
      run<0, true>("fragment 16rows x 64B, rotated", x, rows, row_bytes, grid, sink);
      run<1, false>("full-line 1KiB contiguous", x, rows, row_bytes, grid, sink);
      run<1, true>("full-line 1KiB contiguous, rotated", x, rows, row_bytes, grid, sink);
      run<2, false>("2 rows x 512B", x, rows, row_bytes, grid, sink);
      run<2, true>("2 rows x 512B, rotated", x, rows, row_bytes, grid, sink);
    }
  }
  return 0;
}
