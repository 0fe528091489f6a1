// q/k head RMSNorm and neox rotary embedding on the "16 lanes x 8 bf16 = one 128-wide head" layout
// (lane `sub` = lane & 15 holds elements sub*8 .. sub*8+7). Shared by rope_kv.hip and the fused decode
// attention kernel so both produce bit-identical q / k. mul and add stay separately rounded (as the
// fp32 oracle computes them): no FMA contraction inside these functions.
// Reference semantics: layers/rotary_embedding.py:6-14,37-48; models/qwen3.py:82-85.
#pragma once
#include "common.h"

// Rotate the 8 values held by this lane. `v` are fp32 views of bf16 inputs; cs points at
// cos_sin[pos][0]. Returns fp32 results (caller rounds). The neox half-split pairs element i with
// i+64, i.e. lane `sub` with lane `sub ^ 8` of the same DPP row: the partner arrives by row_ror:8.
__device__ __forceinline__ void rope8(const float* v, const float* __restrict__ cs, int sub, float* o) {
#pragma clang fp contract(off)
  const int f0 = (sub & 7) * 8;  // frequency index of element 0
  const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(cs + f0);
  const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(cs + f0 + 4);
  const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(cs + 64 + f0);
  const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(cs + 64 + f0 + 4);
  const bool upper = sub >= 8;  // this lane holds x2 (elements 64..127)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float c = i < 4 ? c0[i] : c1[i - 4];
    const float s = i < 4 ? s0[i] : s1[i - 4];
    const float other = row16_ror8(v[i]);
    // lower: y1 = x1*c - x2*s ; upper: y2 = x2*c + x1*s   (rotary_embedding.py:12-13)
    const float a = v[i] * c;
    const float b = other * s;
    o[i] = upper ? a + b : a - b;
  }
}

// RMSNorm over one 128-wide head held by 16 lanes; result rounded to bf16 (as the reference
// materialises q/k between the norm graph and the rope graph).
__device__ __forceinline__ void headnorm8(float* v, const bf16_t* __restrict__ w, int sub, float eps) {
#pragma clang fp contract(off)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
  ss = row16_allreduce_sum(ss);
  const float rstd = rsqrtf(ss * (1.f / 128.f) + eps);
  float wf[8];
  unpack8(*reinterpret_cast<const u32x4_t*>(w + sub * 8), wf);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = round_bf16(v[i] * rstd * wf[i]);
}

// Register-resident operands of norm_rope_head, so a caller can ISSUE the loads early and compute late.
struct RopeRegs {
  f32x4_t c0, c1, s0, s1;       // cos / sin of this lane's 8 frequencies
};
__device__ __forceinline__ RopeRegs load_rope_regs(const float* __restrict__ cs, int sub) {
  const int f0 = (sub & 7) * 8;
  RopeRegs r;
  r.c0 = *reinterpret_cast<const f32x4_t*>(cs + f0);
  r.c1 = *reinterpret_cast<const f32x4_t*>(cs + f0 + 4);
  r.s0 = *reinterpret_cast<const f32x4_t*>(cs + 64 + f0);
  r.s1 = *reinterpret_cast<const f32x4_t*>(cs + 64 + f0 + 4);
  return r;
}
// Same arithmetic as headnorm8 + rope8 (bit-identical), operands already in registers.
__device__ __forceinline__ u32x4_t norm_rope_head_regs(const u32x4_t raw, bool has_w, const u32x4_t wraw, float eps,
                                                       const RopeRegs& rr, int sub) {
#pragma clang fp contract(off)
  float v[8], o[8];
  unpack8(raw, v);
  if (has_w) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
    ss = row16_allreduce_sum(ss);
    const float rstd = rsqrtf(ss * (1.f / 128.f) + eps);
    float wf[8];
    unpack8(wraw, wf);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = round_bf16(v[i] * rstd * wf[i]);
  }
  const bool upper = sub >= 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float c = i < 4 ? rr.c0[i] : rr.c1[i - 4];
    const float s = i < 4 ? rr.s0[i] : rr.s1[i - 4];
    const float other = row16_ror8(v[i]);
    const float a = v[i] * c;
    const float b = other * s;
    o[i] = upper ? a + b : a - b;
  }
  return pack8(o);
}

// raw qkv head (8 bf16 of this lane) -> [norm] -> rope at `pos` -> 8 bf16
__device__ __forceinline__ u32x4_t norm_rope_head(const u32x4_t raw, const bf16_t* __restrict__ w, float eps,
                                                  const float* __restrict__ cs, int sub) {
  float v[8], o[8];
  unpack8(raw, v);
  if (w != nullptr) headnorm8(v, w, sub, eps);
  rope8(v, cs, sub, o);
  return pack8(o);
}
