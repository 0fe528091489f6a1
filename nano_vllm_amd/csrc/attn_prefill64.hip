// The long-prompt shape of the varlen causal GQA prefill attention (flash_attn_varlen_func, nano-vllm
// layers/attention.py:64-70; both K/V sources like attn_prefill.hip): ONE wave per SIMD, 64 query rows per wave, the
// 512-register file split by hand — O (64 rows x 128) and Q live in the accumulator registers, S / P / fragments in the
// arch VGPRs — and every matrix instruction of the main loop placed in SOURCE ORDER between slices of the softmax.
//
// Why this shape (profiles/r06_prefill_pp_interval_stamps.txt, r06_prefill_pp_deletion_probes.txt): with two waves per SIMD
// the wave that loses the issue arbitration (priority, then age) runs at 0.3-0.4 of its own rate beside its partner —
// MFMA beside the OTHER wave's VALU does not overlap on this chip, whichever way the work is split between the two
// (lockstep 8-wave loop, ping-pong 8-wave loop). MFMA beside the SAME wave's VALU does: an in-order wave that issues
// an MFMA has ~7 issue slots until the matrix pipe takes the next one (MI355X_MICROARCH.md, cycle constants).
//
// Structure: workgroup = 4 waves = 256 query rows of one (sequence, q-head); wave w owns rows 64 w .. 64 w + 63 as two
// 32-row blocks A and B (lane & 31 = row of the block; the lane halves split keys / head-dim as in attn_prefill.hip).
// Per 64-key tile t a wave issues 64 MFMAs in four slots of 16, each slot carrying one slice of softmax work of the
// OTHER block / tile, so that no MFMA waits for a VALU result of its own slot:
//     slot 1   S_A(t)  = K(t) Q_A^T          beside   finish_B(t-1): P_B = exp2(x_B), row sums, bf16 pack
//     slot 2   S_B(t)  = K(t) Q_B^T          beside   start_A(t):    mask, row max, (deferred) rescale, x_A = S_A c - m_A
//     slot 3   O_B    += V(t-1)^T P_B(t-1)   beside   finish_A(t)
//     slot 4   O_A    += V(t)^T P_A(t)       beside   start_B(t)
// Block B runs half a tile behind block A; one score buffer per block. LDS: K double-buffered, V in a ring of three
// (V(t-1) and V(t) are read in iteration t while V(t+1) is written), one workgroup barrier per tile. The MFMAs are
// `asm volatile` statements (program order is the schedule; O is "+a", Q is "a"), everything else is compiler code fenced
// into its slot with sched_barrier: hipcc allocates registers, counts its own LDS / global loads and pads its own
// hazards; the placement rules below keep every asm MFMA result >= 2 MFMAs away from its first reader.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kQRows = 256;  // query rows per workgroup
constexpr int kKBlk = 64;    // keys per tile
constexpr int kKRowB = 256;  // K tile row bytes in LDS (16-byte XOR swizzle)
constexpr int kVRowB = 320;  // V tile row bytes in LDS (256 + 64 pad: conflict-free tr reads)
constexpr int kKBufB = kKBlk * kKRowB;   // 16 KiB
constexpr int kVBufB = kKBlk * kVRowB;   // 20 KiB
constexpr int kLdsBytes = 2 * kKBufB + 3 * kVBufB;
constexpr float kNegBig = -1.0e30f;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// The matrix instructions as asm statements (free functions: clang refuses asm operands that are by-reference captures of
// a generic lambda). S accumulates in arch VGPRs, O in AGPRs, Q fragments are read from AGPRs.
template <bool FIRST>
__device__ __forceinline__ void mfma_qk(f32x16_t& s, const u32x4_t& kfr, const u32x4_t& qfr) {
  if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(kfr), "a"(qfr));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(kfr), "a"(qfr));
}
// O tuple IDX = 4 X + db is PINNED to a[16 IDX : 16 IDX + 15] in every statement that touches it, so that the (rare)
// rescale can name its registers: arch VALU instructions cannot address AGPRs, and C++ arithmetic on an "+a" variable
// inside the loop makes hipcc keep a VGPR copy of 64 accumulator registers live around the whole loop.
template <int IDX>
__device__ __forceinline__ void mfma_pv(f32x16_t& o, const u32x4_t& vfr, const u32x4_t& pfr) {
  if constexpr (IDX == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[0:15]}"(o) : "v"(vfr), "v"(pfr));
  if constexpr (IDX == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[16:31]}"(o) : "v"(vfr), "v"(pfr));
  if constexpr (IDX == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[32:47]}"(o) : "v"(vfr), "v"(pfr));
  if constexpr (IDX == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[48:63]}"(o) : "v"(vfr), "v"(pfr));
  if constexpr (IDX == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[64:79]}"(o) : "v"(vfr), "v"(pfr));
  if constexpr (IDX == 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[80:95]}"(o) : "v"(vfr), "v"(pfr));
  if constexpr (IDX == 6) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[96:111]}"(o) : "v"(vfr), "v"(pfr));
  if constexpr (IDX == 7) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[112:127]}"(o) : "v"(vfr), "v"(pfr));
}
// O_X *= alpha (per lane): accvgpr_read / v_mul / accvgpr_write over block X's 64 accumulator registers. Callers keep it
// >= one 16-MFMA slot away from the block's P.V MFMAs on either side (the s_nops only cover the asm boundary).
template <int X>
__device__ __forceinline__ void rescale_o(f32x16_t (&o)[4], float alpha) {
  float t0, t1, t2, t3;
  if constexpr (X == 0)
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_accvgpr_read_b32 %[t0], a0\n\tv_accvgpr_read_b32 %[t1], a1\n\tv_accvgpr_read_b32 %[t2], a2\n\tv_accvgpr_read_b32 %[t3], a3\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a0, %[t0]\n\tv_accvgpr_write_b32 a1, %[t1]\n\tv_accvgpr_write_b32 a2, %[t2]\n\tv_accvgpr_write_b32 a3, %[t3]\n\tv_accvgpr_read_b32 %[t0], a4\n\tv_accvgpr_read_b32 %[t1], a5\n\tv_accvgpr_read_b32 %[t2], a6\n\tv_accvgpr_read_b32 %[t3], a7\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a4, %[t0]\n\tv_accvgpr_write_b32 a5, %[t1]\n\tv_accvgpr_write_b32 a6, %[t2]\n\tv_accvgpr_write_b32 a7, %[t3]\n\tv_accvgpr_read_b32 %[t0], a8\n\tv_accvgpr_read_b32 %[t1], a9\n\tv_accvgpr_read_b32 %[t2], a10\n\tv_accvgpr_read_b32 %[t3], a11\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a8, %[t0]\n\tv_accvgpr_write_b32 a9, %[t1]\n\tv_accvgpr_write_b32 a10, %[t2]\n\tv_accvgpr_write_b32 a11, %[t3]\n\tv_accvgpr_read_b32 %[t0], a12\n\tv_accvgpr_read_b32 %[t1], a13\n\tv_accvgpr_read_b32 %[t2], a14\n\tv_accvgpr_read_b32 %[t3], a15\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a12, %[t0]\n\tv_accvgpr_write_b32 a13, %[t1]\n\tv_accvgpr_write_b32 a14, %[t2]\n\tv_accvgpr_write_b32 a15, %[t3]\n\tv_accvgpr_read_b32 %[t0], a16\n\tv_accvgpr_read_b32 %[t1], a17\n\tv_accvgpr_read_b32 %[t2], a18\n\tv_accvgpr_read_b32 %[t3], a19\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a16, %[t0]\n\tv_accvgpr_write_b32 a17, %[t1]\n\tv_accvgpr_write_b32 a18, %[t2]\n\tv_accvgpr_write_b32 a19, %[t3]\n\tv_accvgpr_read_b32 %[t0], a20\n\tv_accvgpr_read_b32 %[t1], a21\n\tv_accvgpr_read_b32 %[t2], a22\n\tv_accvgpr_read_b32 %[t3], a23\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a20, %[t0]\n\tv_accvgpr_write_b32 a21, %[t1]\n\tv_accvgpr_write_b32 a22, %[t2]\n\tv_accvgpr_write_b32 a23, %[t3]\n\tv_accvgpr_read_b32 %[t0], a24\n\tv_accvgpr_read_b32 %[t1], a25\n\tv_accvgpr_read_b32 %[t2], a26\n\tv_accvgpr_read_b32 %[t3], a27\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a24, %[t0]\n\tv_accvgpr_write_b32 a25, %[t1]\n\tv_accvgpr_write_b32 a26, %[t2]\n\tv_accvgpr_write_b32 a27, %[t3]\n\tv_accvgpr_read_b32 %[t0], a28\n\tv_accvgpr_read_b32 %[t1], a29\n\tv_accvgpr_read_b32 %[t2], a30\n\tv_accvgpr_read_b32 %[t3], a31\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a28, %[t0]\n\tv_accvgpr_write_b32 a29, %[t1]\n\tv_accvgpr_write_b32 a30, %[t2]\n\tv_accvgpr_write_b32 a31, %[t3]\n\tv_accvgpr_read_b32 %[t0], a32\n\tv_accvgpr_read_b32 %[t1], a33\n\tv_accvgpr_read_b32 %[t2], a34\n\tv_accvgpr_read_b32 %[t3], a35\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a32, %[t0]\n\tv_accvgpr_write_b32 a33, %[t1]\n\tv_accvgpr_write_b32 a34, %[t2]\n\tv_accvgpr_write_b32 a35, %[t3]\n\tv_accvgpr_read_b32 %[t0], a36\n\tv_accvgpr_read_b32 %[t1], a37\n\tv_accvgpr_read_b32 %[t2], a38\n\tv_accvgpr_read_b32 %[t3], a39\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a36, %[t0]\n\tv_accvgpr_write_b32 a37, %[t1]\n\tv_accvgpr_write_b32 a38, %[t2]\n\tv_accvgpr_write_b32 a39, %[t3]\n\tv_accvgpr_read_b32 %[t0], a40\n\tv_accvgpr_read_b32 %[t1], a41\n\tv_accvgpr_read_b32 %[t2], a42\n\tv_accvgpr_read_b32 %[t3], a43\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a40, %[t0]\n\tv_accvgpr_write_b32 a41, %[t1]\n\tv_accvgpr_write_b32 a42, %[t2]\n\tv_accvgpr_write_b32 a43, %[t3]\n\tv_accvgpr_read_b32 %[t0], a44\n\tv_accvgpr_read_b32 %[t1], a45\n\tv_accvgpr_read_b32 %[t2], a46\n\tv_accvgpr_read_b32 %[t3], a47\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a44, %[t0]\n\tv_accvgpr_write_b32 a45, %[t1]\n\tv_accvgpr_write_b32 a46, %[t2]\n\tv_accvgpr_write_b32 a47, %[t3]\n\tv_accvgpr_read_b32 %[t0], a48\n\tv_accvgpr_read_b32 %[t1], a49\n\tv_accvgpr_read_b32 %[t2], a50\n\tv_accvgpr_read_b32 %[t3], a51\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a48, %[t0]\n\tv_accvgpr_write_b32 a49, %[t1]\n\tv_accvgpr_write_b32 a50, %[t2]\n\tv_accvgpr_write_b32 a51, %[t3]\n\tv_accvgpr_read_b32 %[t0], a52\n\tv_accvgpr_read_b32 %[t1], a53\n\tv_accvgpr_read_b32 %[t2], a54\n\tv_accvgpr_read_b32 %[t3], a55\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a52, %[t0]\n\tv_accvgpr_write_b32 a53, %[t1]\n\tv_accvgpr_write_b32 a54, %[t2]\n\tv_accvgpr_write_b32 a55, %[t3]\n\tv_accvgpr_read_b32 %[t0], a56\n\tv_accvgpr_read_b32 %[t1], a57\n\tv_accvgpr_read_b32 %[t2], a58\n\tv_accvgpr_read_b32 %[t3], a59\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a56, %[t0]\n\tv_accvgpr_write_b32 a57, %[t1]\n\tv_accvgpr_write_b32 a58, %[t2]\n\tv_accvgpr_write_b32 a59, %[t3]\n\tv_accvgpr_read_b32 %[t0], a60\n\tv_accvgpr_read_b32 %[t1], a61\n\tv_accvgpr_read_b32 %[t2], a62\n\tv_accvgpr_read_b32 %[t3], a63\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a60, %[t0]\n\tv_accvgpr_write_b32 a61, %[t1]\n\tv_accvgpr_write_b32 a62, %[t2]\n\tv_accvgpr_write_b32 a63, %[t3]\n\ts_nop 3"
                 : "+{a[0:15]}"(o[0]), "+{a[16:31]}"(o[1]), "+{a[32:47]}"(o[2]), "+{a[48:63]}"(o[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                 : [al] "v"(alpha));
  if constexpr (X == 1)
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_accvgpr_read_b32 %[t0], a64\n\tv_accvgpr_read_b32 %[t1], a65\n\tv_accvgpr_read_b32 %[t2], a66\n\tv_accvgpr_read_b32 %[t3], a67\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a64, %[t0]\n\tv_accvgpr_write_b32 a65, %[t1]\n\tv_accvgpr_write_b32 a66, %[t2]\n\tv_accvgpr_write_b32 a67, %[t3]\n\tv_accvgpr_read_b32 %[t0], a68\n\tv_accvgpr_read_b32 %[t1], a69\n\tv_accvgpr_read_b32 %[t2], a70\n\tv_accvgpr_read_b32 %[t3], a71\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a68, %[t0]\n\tv_accvgpr_write_b32 a69, %[t1]\n\tv_accvgpr_write_b32 a70, %[t2]\n\tv_accvgpr_write_b32 a71, %[t3]\n\tv_accvgpr_read_b32 %[t0], a72\n\tv_accvgpr_read_b32 %[t1], a73\n\tv_accvgpr_read_b32 %[t2], a74\n\tv_accvgpr_read_b32 %[t3], a75\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a72, %[t0]\n\tv_accvgpr_write_b32 a73, %[t1]\n\tv_accvgpr_write_b32 a74, %[t2]\n\tv_accvgpr_write_b32 a75, %[t3]\n\tv_accvgpr_read_b32 %[t0], a76\n\tv_accvgpr_read_b32 %[t1], a77\n\tv_accvgpr_read_b32 %[t2], a78\n\tv_accvgpr_read_b32 %[t3], a79\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a76, %[t0]\n\tv_accvgpr_write_b32 a77, %[t1]\n\tv_accvgpr_write_b32 a78, %[t2]\n\tv_accvgpr_write_b32 a79, %[t3]\n\tv_accvgpr_read_b32 %[t0], a80\n\tv_accvgpr_read_b32 %[t1], a81\n\tv_accvgpr_read_b32 %[t2], a82\n\tv_accvgpr_read_b32 %[t3], a83\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a80, %[t0]\n\tv_accvgpr_write_b32 a81, %[t1]\n\tv_accvgpr_write_b32 a82, %[t2]\n\tv_accvgpr_write_b32 a83, %[t3]\n\tv_accvgpr_read_b32 %[t0], a84\n\tv_accvgpr_read_b32 %[t1], a85\n\tv_accvgpr_read_b32 %[t2], a86\n\tv_accvgpr_read_b32 %[t3], a87\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a84, %[t0]\n\tv_accvgpr_write_b32 a85, %[t1]\n\tv_accvgpr_write_b32 a86, %[t2]\n\tv_accvgpr_write_b32 a87, %[t3]\n\tv_accvgpr_read_b32 %[t0], a88\n\tv_accvgpr_read_b32 %[t1], a89\n\tv_accvgpr_read_b32 %[t2], a90\n\tv_accvgpr_read_b32 %[t3], a91\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a88, %[t0]\n\tv_accvgpr_write_b32 a89, %[t1]\n\tv_accvgpr_write_b32 a90, %[t2]\n\tv_accvgpr_write_b32 a91, %[t3]\n\tv_accvgpr_read_b32 %[t0], a92\n\tv_accvgpr_read_b32 %[t1], a93\n\tv_accvgpr_read_b32 %[t2], a94\n\tv_accvgpr_read_b32 %[t3], a95\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a92, %[t0]\n\tv_accvgpr_write_b32 a93, %[t1]\n\tv_accvgpr_write_b32 a94, %[t2]\n\tv_accvgpr_write_b32 a95, %[t3]\n\tv_accvgpr_read_b32 %[t0], a96\n\tv_accvgpr_read_b32 %[t1], a97\n\tv_accvgpr_read_b32 %[t2], a98\n\tv_accvgpr_read_b32 %[t3], a99\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a96, %[t0]\n\tv_accvgpr_write_b32 a97, %[t1]\n\tv_accvgpr_write_b32 a98, %[t2]\n\tv_accvgpr_write_b32 a99, %[t3]\n\tv_accvgpr_read_b32 %[t0], a100\n\tv_accvgpr_read_b32 %[t1], a101\n\tv_accvgpr_read_b32 %[t2], a102\n\tv_accvgpr_read_b32 %[t3], a103\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a100, %[t0]\n\tv_accvgpr_write_b32 a101, %[t1]\n\tv_accvgpr_write_b32 a102, %[t2]\n\tv_accvgpr_write_b32 a103, %[t3]\n\tv_accvgpr_read_b32 %[t0], a104\n\tv_accvgpr_read_b32 %[t1], a105\n\tv_accvgpr_read_b32 %[t2], a106\n\tv_accvgpr_read_b32 %[t3], a107\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a104, %[t0]\n\tv_accvgpr_write_b32 a105, %[t1]\n\tv_accvgpr_write_b32 a106, %[t2]\n\tv_accvgpr_write_b32 a107, %[t3]\n\tv_accvgpr_read_b32 %[t0], a108\n\tv_accvgpr_read_b32 %[t1], a109\n\tv_accvgpr_read_b32 %[t2], a110\n\tv_accvgpr_read_b32 %[t3], a111\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a108, %[t0]\n\tv_accvgpr_write_b32 a109, %[t1]\n\tv_accvgpr_write_b32 a110, %[t2]\n\tv_accvgpr_write_b32 a111, %[t3]\n\tv_accvgpr_read_b32 %[t0], a112\n\tv_accvgpr_read_b32 %[t1], a113\n\tv_accvgpr_read_b32 %[t2], a114\n\tv_accvgpr_read_b32 %[t3], a115\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a112, %[t0]\n\tv_accvgpr_write_b32 a113, %[t1]\n\tv_accvgpr_write_b32 a114, %[t2]\n\tv_accvgpr_write_b32 a115, %[t3]\n\tv_accvgpr_read_b32 %[t0], a116\n\tv_accvgpr_read_b32 %[t1], a117\n\tv_accvgpr_read_b32 %[t2], a118\n\tv_accvgpr_read_b32 %[t3], a119\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a116, %[t0]\n\tv_accvgpr_write_b32 a117, %[t1]\n\tv_accvgpr_write_b32 a118, %[t2]\n\tv_accvgpr_write_b32 a119, %[t3]\n\tv_accvgpr_read_b32 %[t0], a120\n\tv_accvgpr_read_b32 %[t1], a121\n\tv_accvgpr_read_b32 %[t2], a122\n\tv_accvgpr_read_b32 %[t3], a123\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a120, %[t0]\n\tv_accvgpr_write_b32 a121, %[t1]\n\tv_accvgpr_write_b32 a122, %[t2]\n\tv_accvgpr_write_b32 a123, %[t3]\n\tv_accvgpr_read_b32 %[t0], a124\n\tv_accvgpr_read_b32 %[t1], a125\n\tv_accvgpr_read_b32 %[t2], a126\n\tv_accvgpr_read_b32 %[t3], a127\n\tv_mul_f32 %[t0], %[t0], %[al]\n\tv_mul_f32 %[t1], %[t1], %[al]\n\tv_mul_f32 %[t2], %[t2], %[al]\n\tv_mul_f32 %[t3], %[t3], %[al]\n\tv_accvgpr_write_b32 a124, %[t0]\n\tv_accvgpr_write_b32 a125, %[t1]\n\tv_accvgpr_write_b32 a126, %[t2]\n\tv_accvgpr_write_b32 a127, %[t3]\n\ts_nop 3"
                 : "+{a[64:79]}"(o[0]), "+{a[80:95]}"(o[1]), "+{a[96:111]}"(o[2]), "+{a[112:127]}"(o[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                 : [al] "v"(alpha));
}
// register pins: an empty asm makes a value opaque at this point of the asm-volatile order (pure VALU code is otherwise
// free to drift across the asm MFMAs before the machine scheduler ever sees the sched_barrier fences)
__device__ __forceinline__ void pin(f32x16_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(u32x4_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <bool PAGED>
__global__ __launch_bounds__(256, 1) void prefill_w64_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t k_tok_stride,
    int64_t v_tok_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, bf16_t* __restrict__ out, int num_seqs, int hq,
    int hkv, int block_size, float scale_log2e, float* __restrict__ lse, float rescale_thr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qcol = lane & 31, hi = lane >> 5;
  int head = blockIdx.x;
  const int tile_rank = blockIdx.y;

  // ---- which (sequence, 256-row q block)? lane i holds sequence i's bounds (launches of <= 64 sequences), longest
  // blocks first (attn_prefill.hip) --------------------------------------------------------------------------------
  int a0 = 0, a1 = 0, b0 = 0, b1 = 0;
  if (lane < num_seqs) {
    a0 = cu_q[lane]; a1 = cu_q[lane + 1];
    b0 = cu_k[lane]; b1 = cu_k[lane + 1];
  }
  const int val = (a1 - a0 + kQRows - 1) / kQRows;
  int sc = val;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(sc, o, 64);
    if (lane >= o) sc += n;
  }
  const int total = __builtin_amdgcn_readlane(sc, 63);
  if (tile_rank >= total) return;
  const int tile = total - 1 - tile_rank;
  int seq = __popcll(__ballot(sc <= tile));
  int qblk = tile - __builtin_amdgcn_readlane(sc - val, seq);
  int q0 = __builtin_amdgcn_readlane(a0, seq);
  int lq = __builtin_amdgcn_readlane(a1, seq) - q0;
  int k0 = __builtin_amdgcn_readlane(b0, seq);
  int lk = __builtin_amdgcn_readlane(b1, seq) - k0;
  seq = __builtin_amdgcn_readfirstlane(seq);
  qblk = __builtin_amdgcn_readfirstlane(qblk);
  q0 = __builtin_amdgcn_readfirstlane(q0);
  lq = __builtin_amdgcn_readfirstlane(lq);
  k0 = __builtin_amdgcn_readfirstlane(k0);
  lk = __builtin_amdgcn_readfirstlane(lk);
  head = __builtin_amdgcn_readfirstlane(head);

  const int kvh = head / (hq / hkv);
  const int off = lk - lq;                                         // query i sees keys j <= i + off
  const int kv_end = min(lk, qblk * kQRows + kQRows + off);        // keys visible to the block's last query
  const int nt = (kv_end + kKBlk - 1) / kKBlk;                     // key tiles of the workgroup's item (>= 1)
  const int row0 = qblk * kQRows + wave * 64;                      // the wave's first row
  // per 32-row block X (0 = A, 1 = B): this lane's row (clamped), the last key it sees
  int qi[2], qi_c[2], kmax_vis[2], kmin_blk[2];
#pragma unroll
  for (int X = 0; X < 2; ++X) {
    qi[X] = row0 + X * 32 + qcol;
    qi_c[X] = qi[X] < lq ? qi[X] : lq - 1;
    kmax_vis[X] = qi_c[X] + off;
    kmin_blk[X] = __builtin_amdgcn_readfirstlane(min(row0 + X * 32, lq - 1) + off);   // every row of the block sees keys <= this
  }
  // tiles this wave has work in (the causal frontier of its last row); past them it only helps staging
  const int ntw = row0 < lq ? min(nt, (min(row0 + 63, lq - 1) + off) / kKBlk + 1) : 0;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (row, hi) holds d = ds*16 + 8 hi .. + 8; they live in AGPRs ----
  u32x4_t qf[2][8];
#pragma unroll
  for (int X = 0; X < 2; ++X) {
    const bf16_t* qp = q + ((int64_t)(q0 + qi_c[X]) * hq + head) * 128 + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[X][ds] = *reinterpret_cast<const u32x4_t*>(qp + ds * 16);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- staging: each thread moves 4 16-byte chunks of K and of V per tile (chunk n: row (tid >> 4) + 16 n, column tid & 15)
  u32x4_t kreg[4], vreg[4];
  const int64_t kstride = PAGED ? 128 : k_tok_stride, vstride = PAGED ? 128 : v_tok_stride;
  const int srow = tid >> 4, sc16 = tid & 15;
  unsigned int koff[4], voff[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    koff[n] = ((unsigned int)((srow + n * 16) * kstride) + sc16 * 8) * 2u;
    voff[n] = ((unsigned int)((srow + n * 16) * vstride) + sc16 * 8) * 2u;
  }
  auto load_tile = [&](int tt) __attribute__((always_inline)) {      // global -> kreg / vreg, rows past the end read as zeros
    const int kt = tt * kKBlk;
    __amdgpu_buffer_rsrc_t krs, vrs;
    int ksoff, vsoff;
    if constexpr (PAGED) {
      const int blk = block_tables[(int64_t)seq * bt_stride + kt / block_size];
      const int64_t base = (((int64_t)blk * hkv + kvh) * block_size + (kt % block_size)) * 128;
      krs = __builtin_amdgcn_make_buffer_rsrc((void*)(k + base), 0, kKBlk * 256, 0x00020000);
      vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(v + base), 0, kKBlk * 256, 0x00020000);
      ksoff = vsoff = 0;
    } else {
      krs = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)k0 * k_tok_stride + kvh * 128), 0,
                                              (int)(lk * kstride * 2), 0x00020000);
      vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)k0 * v_tok_stride + kvh * 128), 0,
                                              (int)(lk * vstride * 2), 0x00020000);
      ksoff = (int)(kt * kstride * 2);
      vsoff = (int)(kt * vstride * 2);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      kreg[n] = __builtin_amdgcn_raw_buffer_load_b128(krs, koff[n], ksoff, 0);
      vreg[n] = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff[n], vsoff, 0);
    }
  };
  auto write_tile = [&](int kbuf, int vbuf) __attribute__((always_inline)) {   // kreg / vreg -> LDS K buffer kbuf, V buffer vbuf
    unsigned char* kl = smem + kbuf * kKBufB;
    unsigned char* vl = smem + 2 * kKBufB + vbuf * kVBufB;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int row = srow + n * 16;
      *reinterpret_cast<u32x4_t*>(kl + row * kKRowB + ((sc16 ^ (row & 15)) << 4)) = kreg[n];
      *reinterpret_cast<u32x4_t*>(vl + row * kVRowB + (sc16 << 4)) = vreg[n];
    }
  };

  // ---- accumulators and running statistics ---------------------------------------------------------------------------
  f32x16_t oacc[2][4];       // [block][32-wide head-dim block]: AGPRs (only "+a" asm and the rare rescale / epilogue touch them)
#pragma unroll
  for (int X = 0; X < 2; ++X)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[X][db][r] = 0.f;
  f32x16_t sacc[2][2];       // [block][32-key block]: scores, then x = S c - m in place
  u32x4_t pw[2][2][2];       // [block][kb][r0]: P^T fragments, bf16 pairs (k-slot (hi, e) <-> score register r0*8 + e)
  float m_run[2] = {kNegBig, kNegBig}, l_run[2] = {0.f, 0.f};

  // loop-invariant LDS byte offsets of this lane's K fragments (row qcol, swizzled slot) and of its V transpose reads
  int kslot[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) kslot[ds] = qcol * kKRowB + (((ds * 2 + hi) ^ (qcol & 15)) << 4);
  const int i16 = lane & 15;
  const int vlane = 2 * kKBufB + (4 * hi + (i16 >> 2)) * kVRowB + (16 * ((lane >> 4) & 1) + (i16 & 3) * 4) * 2;

  // ---- prologue: tile 0 into LDS -------------------------------------------------------------------------------------
  load_tile(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the Q fragments have landed as well
#pragma unroll
  for (int X = 0; X < 2; ++X)
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) asm volatile("" : "+a"(qf[X][ds]));
  write_tile(0, 0);
  __syncthreads();

  // fragment rings (three reads ahead of their MFMA)
  u32x4_t kf[4], vf[4];
  // K fragment h (0 .. 15: kb = h >> 3, ds = h & 7) of the K buffer at byte offset kb_off
  auto read_k = [&](int h, int kb_off) __attribute__((always_inline)) {
    kf[h & 3] = *reinterpret_cast<const u32x4_t*>(smem + kb_off + ((h >> 3) & 1) * 32 * kKRowB + kslot[h & 7]);
  };
  // V fragment j (0 .. 15: (kb, r0) = j >> 2, db = j & 3) of the V buffer whose lane base is vb (a VGPR)
  auto read_v = [&](int j, int vb) __attribute__((always_inline)) {
    const int kbr = j >> 2, db = j & 3;
    const unsigned char* p0 = smem + vb + kbr * 16 * kVRowB + db * 64;
    const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
    const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 8 * kVRowB));
    u32x2_t lo = __builtin_bit_cast(u32x2_t, x0), hi2 = __builtin_bit_cast(u32x2_t, x1);
    vf[j & 3] = u32x4_t{lo[0], lo[1], hi2[0], hi2[1]};
  };
#define NVL_FENCE() __builtin_amdgcn_sched_barrier(0)

  // ---- the softmax slices ----------------------------------------------------------------------------------------------
  // start_X, 16 chunks: [0,1] max over the first key block (chunk 0 masks it first when the tile straddles the block's causal
  // frontier), [2,3] the second key block, [4] cross-half max, decision, (rare) rescale of O_X / l_X, [5..15] x = S c - m
  float mxa = 0.f, mxb = 0.f;
  auto start_chunk = [&](auto Xc, auto cc, int kt) __attribute__((always_inline)) {
    constexpr int X = decltype(Xc)::value, c = decltype(cc)::value;
    pin(sacc[X][0]); pin(sacc[X][1]); pin(mxa); pin(mxb); pin(m_run[X]); pin(l_run[X]);
    if constexpr (c == 0 || c == 2) {
      constexpr int kb = c >> 1;
      if (kt + kKBlk - 1 > kmin_blk[X]) {          // key(kb, r) = kt + 4 hi + kb*32 + (r & 3) + 8 (r >> 2)
        const int lim = kmax_vis[X] - kt - 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[X][kb][r] = (kb * 32 + (r & 3) + 8 * (r >> 2)) <= lim ? sacc[X][kb][r] : kNegBig;
      }
    }
    if constexpr (c < 4) {
      constexpr int kb = c >> 1;
      const f32x16_t& s = sacc[X][kb];
      float m = (c & 1) ? (kb ? mxb : mxa) : 0.f;
      if constexpr ((c & 1) == 0) {
        m = max3f(s[0], s[1], s[2]);
        m = max3f(m, s[3], s[4]);
        m = max3f(m, s[5], s[6]);
        m = max3f(m, s[7], s[8]);
      } else {
        m = max3f(m, s[9], s[10]);
        m = max3f(m, s[11], s[12]);
        m = max3f(m, s[13], s[14]);
        m = max3f(m, s[15], s[15]);
      }
      if constexpr (kb) mxb = m; else mxa = m;
    } else if constexpr (c == 4) {
      float mx = max3f(mxa, mxb, mxb);
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = max3f(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
      const float m_new = max3f(m_run[X], mx * scale_log2e, m_run[X]);
      // deferred rescale (attn_prefill.hip; cdna_hip_programming.md T13): the block's previous P.V is complete (it ran
      // >= one 16-MFMA slot ago), this tile's P are exponentiated after the decision, O and l take the same factor
      if (__builtin_expect(__any(m_new > m_run[X] + rescale_thr), 0)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[X] - m_new);
        l_run[X] *= alpha;
        rescale_o<X>(oacc[X], alpha);
        m_run[X] = m_new;
      }
    } else {
      constexpr int e0 = (c - 5) * 3, n = c == 15 ? 2 : 3;
#pragma unroll
      for (int e = e0; e < e0 + n; ++e)
        sacc[X][e >> 4][e & 15] = fmaf(sacc[X][e >> 4][e & 15], scale_log2e, -m_run[X]);
    }
    pin(sacc[X][0]); pin(sacc[X][1]); pin(mxa); pin(mxb); pin(m_run[X]); pin(l_run[X]);
  };
  // finish_X, 16 chunks: two scores each -> exp2, row sums (two chains), one packed bf16 word of P
  float ps0 = 0.f, ps1 = 0.f;
  auto finish_chunk = [&](auto Xc, auto cc) __attribute__((always_inline)) {
    constexpr int X = decltype(Xc)::value, c = decltype(cc)::value;
    constexpr int kb = c >> 3, r = (c & 7) * 2;
    pin(sacc[X][kb]); pin(ps0); pin(ps1);
    const float p0 = __builtin_amdgcn_exp2f(sacc[X][kb][r]), p1 = __builtin_amdgcn_exp2f(sacc[X][kb][r + 1]);
    if constexpr (c == 0) { ps0 = p0; ps1 = p1; } else { ps0 += p0; ps1 += p1; }
    pw[X][kb][(c & 7) >> 2][c & 3] = pack_bf16x2(p0, p1);
    if constexpr (c == 15) l_run[X] += ps0 + ps1;
    pin(pw[X][kb][(c & 7) >> 2]); pin(ps0); pin(ps1); pin(l_run[X]);
  };

  // ---- the matrix instructions (asm: program order is the schedule) ---------------------------------------------------
  auto qk_mfma = [&](auto Xc, auto hc) __attribute__((always_inline)) {
    constexpr int X = decltype(Xc)::value, h = decltype(hc)::value;
    constexpr int kb = h >> 3, ds = h & 7;
    mfma_qk<ds == 0>(sacc[X][kb], kf[h & 3], qf[X][ds]);
  };
  auto pv_mfma = [&](auto Xc, auto jc) __attribute__((always_inline)) {
    constexpr int X = decltype(Xc)::value, j = decltype(jc)::value;
    constexpr int kbr = j >> 2, db = j & 3;
    mfma_pv<X * 4 + db>(oacc[X][db], vf[j & 3], pw[X][kbr >> 1][kbr & 1]);
  };

  // ---- one tile step: CUR = tile t has work for this wave, PREV = so had tile t - 1 ------------------------------------
  auto step = [&](auto cur_c, auto prev_c, const int t, const int kpar, const int vcur, const int vprev, const int vnext) __attribute__((always_inline)) {
    constexpr bool CUR = decltype(cur_c)::value, PREV = decltype(prev_c)::value;
    const int kt = t * kKBlk;
    const int kb_off = kpar * kKBufB;
    const int vb_cur = vlane + vcur * kVBufB, vb_prev = vlane + vprev * kVBufB;
    const bool more = t + 1 < nt;
    if (more) load_tile(t + 1);
    NVL_FENCE();
    // slot 1: QK_A(t) beside finish_B(t - 1)
    if constexpr (CUR) { read_k(0, kb_off); read_k(1, kb_off); read_k(2, kb_off); }
    NVL_FENCE();
    auto slot1 = [&](auto hc) __attribute__((always_inline)) {
      constexpr int h = decltype(hc)::value;
      if constexpr (CUR) read_k((h + 3) & 15, kb_off);          // h + 3 >= 16: slot 2 reads the same fragments again
      if constexpr (PREV) finish_chunk(std::integral_constant<int, 1>{}, hc);
      NVL_FENCE();
      if constexpr (CUR) qk_mfma(std::integral_constant<int, 0>{}, hc);
      NVL_FENCE();
    };
    auto slot2 = [&](auto hc) __attribute__((always_inline)) {
      constexpr int h = decltype(hc)::value;
      if constexpr (CUR) {
        if constexpr (h + 3 < 16) read_k(h + 3, kb_off);
        else if constexpr (PREV) read_v(h + 3 - 16, vb_prev);   // slot 3's first fragments
        else read_v(h + 3 - 16, vb_cur);                        // (no slot 3: slot 4's)
        start_chunk(std::integral_constant<int, 0>{}, hc, kt);
        NVL_FENCE();
        qk_mfma(std::integral_constant<int, 1>{}, hc);
        NVL_FENCE();
      }
    };
    auto slot3 = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (PREV) {
        if constexpr (j + 3 < 16) read_v(j + 3, vb_prev);
        else if constexpr (CUR) read_v(j + 3 - 16, vb_cur);     // slot 4's first fragments
      }
      if constexpr (CUR) finish_chunk(std::integral_constant<int, 0>{}, jc);
      NVL_FENCE();
      if constexpr (PREV) pv_mfma(std::integral_constant<int, 1>{}, jc);
      NVL_FENCE();
    };
    auto slot4 = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      if constexpr (CUR) {
        if constexpr (j + 3 < 16) read_v(j + 3, vb_cur);
        start_chunk(std::integral_constant<int, 1>{}, jc, kt);
        NVL_FENCE();
        pv_mfma(std::integral_constant<int, 0>{}, jc);
        NVL_FENCE();
      }
    };
#define NVL_16(F)                                                                                                       \
    F(std::integral_constant<int, 0>{}); F(std::integral_constant<int, 1>{}); F(std::integral_constant<int, 2>{});      \
    F(std::integral_constant<int, 3>{}); F(std::integral_constant<int, 4>{}); F(std::integral_constant<int, 5>{});      \
    F(std::integral_constant<int, 6>{}); F(std::integral_constant<int, 7>{}); F(std::integral_constant<int, 8>{});      \
    F(std::integral_constant<int, 9>{}); F(std::integral_constant<int, 10>{}); F(std::integral_constant<int, 11>{});    \
    F(std::integral_constant<int, 12>{}); F(std::integral_constant<int, 13>{}); F(std::integral_constant<int, 14>{});   \
    F(std::integral_constant<int, 15>{});
    NVL_16(slot1)
    NVL_16(slot2)
    if constexpr (!CUR && PREV) { read_v(0, vb_prev); read_v(1, vb_prev); read_v(2, vb_prev); NVL_FENCE(); }
    NVL_16(slot3)
    NVL_16(slot4)
    if (more) write_tile(kpar ^ 1, vnext);
    NVL_FENCE();
    __syncthreads();
    NVL_FENCE();
  };

  // ---- the tile loop: every wave passes nt + 1 barriers ---------------------------------------------------------------
  int t = 0, vcur = 0, vprev = 2, vnext = 1;          // V ring slots of tiles t, t - 1, t + 1
  auto advance = [&]() { ++t; vprev = vcur; vcur = vnext; vnext = vnext == 2 ? 0 : vnext + 1; };
  if (ntw > 0) {
    step(std::true_type{}, std::false_type{}, t, t & 1, vcur, vprev, vnext);
    advance();
    for (; t < ntw; advance()) step(std::true_type{}, std::true_type{}, t, t & 1, vcur, vprev, vnext);
    step(std::false_type{}, std::true_type{}, t, t & 1, vcur, vprev, vnext);       // finish_B / P.V_B of the wave's last tile
    advance();
  }
  for (; t <= nt; advance()) {                        // NOLINT tiles above this wave's rows: staging only
    const bool more = t + 1 < nt;
    if (more) load_tile(t + 1);
    if (more) write_tile((t & 1) ^ 1, vnext);
    __syncthreads();
  }

  // ---- epilogue: normalise and store O[query][d] (attn_prefill.hip: permlane swaps, dwordx4 stores) -------------------
  asm volatile("s_nop 15\n\ts_nop 15"
               : "+{a[0:15]}"(oacc[0][0]), "+{a[16:31]}"(oacc[0][1]), "+{a[32:47]}"(oacc[0][2]), "+{a[48:63]}"(oacc[0][3]), "+{a[64:79]}"(oacc[1][0]), "+{a[80:95]}"(oacc[1][1]), "+{a[96:111]}"(oacc[1][2]), "+{a[112:127]}"(oacc[1][3]));
#pragma unroll
  for (int X = 0; X < 2; ++X) {
    float l_tot;
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[X]), __float_as_uint(l_run[X]), false, false);
      l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.f / l_tot;
    const bool q_valid = qi[X] < lq;
    if (lse != nullptr && q_valid && hi == 0)
      lse[(int64_t)(q0 + qi[X]) * hq + head] = 0.6931471805599453f * (m_run[X] + log2f(l_tot));
    bf16_t* op = out + ((int64_t)(q0 + qi_c[X]) * hq + head) * 128 + 8 * hi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const int r0 = rp * 8;
        const unsigned int ax = pack_bf16x2(oacc[X][db][r0 + 0] * inv, oacc[X][db][r0 + 1] * inv);
        const unsigned int ay = pack_bf16x2(oacc[X][db][r0 + 2] * inv, oacc[X][db][r0 + 3] * inv);
        const unsigned int bx = pack_bf16x2(oacc[X][db][r0 + 4] * inv, oacc[X][db][r0 + 5] * inv);
        const unsigned int by = pack_bf16x2(oacc[X][db][r0 + 6] * inv, oacc[X][db][r0 + 7] * inv);
        const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        if (q_valid) *reinterpret_cast<u32x4_t*>(op + db * 32 + 16 * rp) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
      }
  }
#undef NVL_16
#undef NVL_FENCE
}

}  // namespace

// Launch of the 64-rows-per-wave shape; arguments as validated by nvl_attn_prefill_varlen (attn_prefill.hip), which calls
// this for launches of <= 64 sequences with long prompts. Returns 0 / NVL_E*.
int nvl_prefill_w64_launch(const void* q, const void* k, const void* v, int64_t k_tok_stride, int64_t v_tok_stride,
                           const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, const int32_t* block_tables,
                           int64_t bt_stride, void* out, int64_t total_q, int num_seqs, int num_q_heads, int num_kv_heads,
                           int block_size, float scale_log2e, float* lse, float rescale_thr, hipStream_t s) {
  const int64_t tiles = (total_q + kQRows - 1) / kQRows + num_seqs;  // upper bound on sum ceil(Lq / 256)
  NVL_REQUIRE(tiles <= 65535 && num_seqs <= 64, "nvl_attn_prefill_varlen: too many query tiles for the 64-row shape (%lld)", (long long)tiles);
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& done = attr_done[nvl_device_slot()];
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_w64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_w64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess) {
      nvl_set_error("nvl_attn_prefill_varlen: cannot reserve %d B of LDS", kLdsBytes);
      return NVL_ELAUNCH;
    }
    done = true;
  }
  dim3 grid((unsigned)num_q_heads, (unsigned)tiles);
  if (block_tables != nullptr)
    hipLaunchKernelGGL((prefill_w64_kernel<true>), grid, dim3(256), kLdsBytes, s, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, k_tok_stride, v_tok_stride, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride,
                       (bf16_t*)out, num_seqs, num_q_heads, num_kv_heads, block_size, scale_log2e, lse, rescale_thr);
  else
    hipLaunchKernelGGL((prefill_w64_kernel<false>), grid, dim3(256), kLdsBytes, s, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, k_tok_stride, v_tok_stride, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride,
                       (bf16_t*)out, num_seqs, num_q_heads, num_kv_heads, block_size, scale_log2e, lse, rescale_thr);
  return nvl_check_launch("nvl_attn_prefill_varlen");
}
