// Shared device/host helpers for the gfx950 kernels of libnvl_hip.so.
// CDNA4 only: 64-wide wavefronts are hard-coded (no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/nvl.h"

#define NVL_WAVE 64
#define NVL_HEAD_DIM 128

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ---- host-side error plumbing -------------------------------------------------
void nvl_set_error(const char* fmt, ...);
int nvl_check_launch(const char* what);
// Launcher-side caches (CU count, per-kernel LDS opt-in) are kept PER DEVICE so the entry points are re-entrant
// per device: index of the calling thread's current HIP device, clamped to the table size.
#define NVL_MAX_DEVICES 16
int nvl_device_slot(void);

#define NVL_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      nvl_set_error(__VA_ARGS__);         \
      return NVL_EINVAL;                  \
    }                                     \
  } while (0)

// ---- bf16 <-> fp32 ------------------------------------------------------------
// A 32-bit word holds two bf16: element 0 in the low half.
__device__ __forceinline__ float bf16lo_to_f32(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// Round-to-nearest-even fp32 -> bf16 (the compiler emits v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  bf16x2_t v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float round_bf16(float x) { return (float)(bf16_t)x; }

__device__ __forceinline__ void unpack8(const u32x4_t& w, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf16lo_to_f32(w[i]);
    f[2 * i + 1] = bf16hi_to_f32(w[i]);
  }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  u32x4_t w;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
  return w;
}

// ---- OCP fp8 (e4m3fn: what gfx950's conversion instructions implement) <-> fp32, 4 values per dword ----------
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// v_cvt_pk_fp8_f32 does not saturate: a magnitude above the largest e4m3 value (448) would be stored as NaN and
// poison every later decode step of that sequence, so the operands are clamped first (what hip_fp8's software path
// does as well); in-range values are unaffected (bit-exact with torch's float8_e4m3fn cast, tests/test_kernels_gpu.py).
__device__ __forceinline__ float clamp_e4m3(float v) { return __builtin_amdgcn_fmed3f(v, 448.f, -448.f); }
__device__ __forceinline__ unsigned int pack_fp8x4(float a, float b, float c, float d) {
  a = clamp_e4m3(a); b = clamp_e4m3(b); c = clamp_e4m3(c); d = clamp_e4m3(d);
  unsigned int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return w;
}
__device__ __forceinline__ void unpack_fp8x4(unsigned int w, float* f) {
  const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
  const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
  f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}
// 8 bf16 (one 16-byte chunk) -> 8 fp8 (8 bytes), round to nearest even
__device__ __forceinline__ u32x2_t bf16x8_to_fp8x8(const u32x4_t& w) {
  float f[8];
  unpack8(w, f);
  u32x2_t o = {pack_fp8x4(f[0], f[1], f[2], f[3]), pack_fp8x4(f[4], f[5], f[6], f[7])};
  return o;
}
// 16 fp8 (16 bytes) -> two 16-byte chunks of bf16 (exact: every e4m3 value is a bf16 value)
__device__ __forceinline__ void fp8x16_to_bf16(const u32x4_t& w, u32x4_t* lo, u32x4_t* hi) {
  float f[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack_fp8x4(w[i], f + 4 * i);
  *lo = pack8(f);
  *hi = pack8(f + 8);
}

// ---- cross-lane -----------------------------------------------------------------
// Sum over the 16 lanes of a DPP row (lanes 16r..16r+15); every lane gets the total.
__device__ __forceinline__ float row16_allreduce_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}
// Sum over the 8 lanes of a half DPP row (lanes 8r..8r+7); every lane gets the total.
__device__ __forceinline__ float half8_allreduce_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  return v;
}
// Value held by the lane 8 positions away inside the same 16-lane row.
__device__ __forceinline__ float row16_ror8(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
}
// Sum over the wave, every lane gets the total. All on the VALU: four DPP row rotations inside the 16-lane rows, then
// v_permlane16_swap / v_permlane32_swap to add rows 0+1 | 2+3 and the two halves. (hipcc lowers a __shfl_xor butterfly
// to six dependent ds_bpermute_b32 — six LDS round trips, ~0.3 us in the middle of the latency-bound norm kernels.)
__device__ __forceinline__ float wave_allreduce_sum(float v) {
  v = row16_allreduce_sum(v);
  const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
  const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
}
__device__ __forceinline__ float wave_allreduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Natural log on the transcendental unit for arguments that are never denormal (the sampler's uniforms are >= 2^-25,
// its exponentials >= 1e-10): v_log_f32 (log2) x ln 2 — what hipcc's __logf computes, minus its denormal range fix-up
// (a compare, two selects, a scale and a subtraction per call, ~6 VALU), which never fires here. Bit-identical to
// __logf on normal inputs.
__device__ __forceinline__ float log_normal_f32(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// ---- Philox4x32-10 (host + device, identical) -------------------------------------
struct Philox4 {
  uint32_t v[4];
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}
