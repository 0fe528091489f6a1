// Varlen causal (bottom-right aligned) GQA prefill attention for gfx950 on the matrix cores.
// Replaces flash_attn_varlen_func as used by nano-vllm layers/attention.py:64-70, both K/V
// sources: packed [sum Lk, Hkv, 128] tensors, or the paged cache + block table (prefix cache /
// chunked-prefill continuation).
//
// Structure (flash-style, one workgroup = 4 waves = 128 query rows of one (sequence, q-head)):
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, operands swapped so that each LANE owns one
//     query column: the softmax row-reduction is 31 in-lane max/adds + one cross-half exchange.
//   * K tile (64 keys x 128) staged in LDS with a 16-byte XOR swizzle -> conflict-free
//     ds_read_b128 A-fragments; V tile staged row-major (320-byte row stride) and fed to the
//     P.V MFMA through ds_read_b64_tr_b16 (hardware transpose read).
//   * O^T = V^T . P^T accumulates in registers (lane = query column, so the online-softmax
//     rescale is lane-local). The MFMA k-slot <-> key mapping of the P.V product is permuted
//     to match the S^T accumulator layout, so P never moves between lanes.
//   * online softmax in fp32 (base-2), P rounded to bf16 before P.V, fp32 accumulation.
// The (sequence, q-block) of a workgroup is found on device from cu_seqlens_q (prefix sum in
// LDS + binary search), so no host-side tile list is needed.
#include "common.h"

namespace {

constexpr int kQBlk = 128;   // query rows per workgroup (4 waves x 32)
constexpr int kKBlk = 64;    // keys per tile
constexpr int kKRowB = 256;  // K tile row bytes in LDS
constexpr int kVRowB = 320;  // V tile row bytes in LDS (256 + 64 pad: conflict-free tr reads)
constexpr float kNegBig = -1.0e30f;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4_t& w) { return __builtin_bit_cast(bf16x8_t, w); }

struct SeqTile {
  int seq, qblk;
};

template <bool PAGED>
__global__ __launch_bounds__(256) void prefill_attn_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t k_tok_stride,
    int64_t v_tok_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, bf16_t* __restrict__ out, int num_seqs, int hq,
    int hkv, int block_size, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* k_lds = smem;                          // 64 x 256 B
  unsigned char* v_lds = smem + kKBlk * kKRowB;         // 64 x 320 B
  int* wsum = reinterpret_cast<int*>(v_lds + kKBlk * kVRowB);
  int* pre = wsum + 4;                                  // [num_seqs + 1]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qcol = lane & 31, hi = lane >> 5;

  // ---- which (sequence, q-block) is this workgroup? ------------------------------------
  {
    int carry = 0;
    if (tid == 0) pre[0] = 0;
    for (int base = 0; base < num_seqs; base += 256) {
      const int i = base + tid;
      int val = 0;
      if (i < num_seqs) val = (cu_q[i + 1] - cu_q[i] + kQBlk - 1) / kQBlk;
      int s = val;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(s, o, 64);
        if (lane >= o) s += n;
      }
      if (lane == 63) wsum[wave] = s;
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int t = wsum[w];
        if (w < wave) woff += t;
        tot += t;
      }
      if (i < num_seqs) pre[i + 1] = carry + woff + s;
      carry += tot;
      __syncthreads();
    }
  }
  __syncthreads();
  const int tile = blockIdx.x;
  if (tile >= pre[num_seqs]) return;  // grid is an upper bound
  int lo = 0, hi_s = num_seqs;
  while (hi_s - lo > 1) {
    const int mid = (lo + hi_s) >> 1;
    if (pre[mid] <= tile) lo = mid; else hi_s = mid;
  }
  const int seq = lo;
  const int qblk = tile - pre[seq];
  const int head = blockIdx.y;
  const int kvh = head / (hq / hkv);

  const int q0 = cu_q[seq], lq = cu_q[seq + 1] - q0;
  const int k0 = cu_k[seq], lk = cu_k[seq + 1] - k0;
  const int off = lk - lq;  // bottom-right alignment: query i sees keys j <= i + off
  const int qi = qblk * kQBlk + wave * 32 + qcol;
  const bool q_valid = qi < lq;
  const int qi_c = q_valid ? qi : lq - 1;
  const int kv_end = min(lk, qblk * kQBlk + kQBlk + off);  // keys visible to the block's last query

  // ---- Q fragments: B operand of S^T = K.Q^T : lane (query, hi) holds d = ds*16 + 8*hi .. +8
  bf16x8_t qf[8];
  {
    const bf16_t* qp = q + ((int64_t)(q0 + qi_c) * hq + head) * 128 + hi * 8;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = as_bf16x8(*reinterpret_cast<const u32x4_t*>(qp + ds * 16));
  }

  f32x16_t oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = kNegBig, l_run = 0.f;

  const int kmax_vis = qi_c + off;  // last key this query may see

  for (int kt = 0; kt < kv_end; kt += kKBlk) {
    __syncthreads();  // previous tile fully consumed
    // ---- stage K and V tiles (64 rows x 256 B each): 1024 16-byte chunks per tensor ----------
    {
      int64_t kbase, vbase, kstride, vstride;
      if constexpr (PAGED) {
        const int blk = block_tables[(int64_t)seq * bt_stride + kt / block_size];
        kbase = (((int64_t)blk * hkv + kvh) * block_size + (kt % block_size)) * 128;
        vbase = kbase;
        kstride = vstride = 128;
      } else {
        kbase = (int64_t)(k0 + kt) * k_tok_stride + kvh * 128;
        vbase = (int64_t)(k0 + kt) * v_tok_stride + kvh * 128;
        kstride = k_tok_stride;
        vstride = v_tok_stride;
      }
      const int rows_ok = lk - kt;  // rows >= rows_ok are clamped to the last valid row (masked later)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int chunk = tid + n * 256;
        const int row = chunk >> 4, c16 = chunk & 15;
        const int rsrc = row < rows_ok ? row : rows_ok - 1;
        const u32x4_t kw = *reinterpret_cast<const u32x4_t*>(k + kbase + (int64_t)rsrc * kstride + c16 * 8);
        const u32x4_t vw = *reinterpret_cast<const u32x4_t*>(v + vbase + (int64_t)rsrc * vstride + c16 * 8);
        *reinterpret_cast<u32x4_t*>(k_lds + row * kKRowB + ((c16 ^ (row & 15)) << 4)) = kw;
        *reinterpret_cast<u32x4_t*>(v_lds + row * kVRowB + (c16 << 4)) = vw;
      }
    }
    __syncthreads();

    // ---- S^T tile: 2 key blocks x 32 keys; lane = query column ---------------------------------
    f32x16_t sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
      const int row = kb * 32 + qcol;  // A operand: lane (key row, hi) holds d = ds*16 + 8*hi .. +8
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) {
        const int slot = (ds * 2 + hi) ^ (row & 15);
        const bf16x8_t a = as_bf16x8(*reinterpret_cast<const u32x4_t*>(k_lds + row * kKRowB + (slot << 4)));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ds], sacc[kb], 0, 0, 0);
      }
    }
    // ---- mask + online softmax (base 2) ----------------------------------------------------------
    float mx = kNegBig;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float s = key <= kmax_vis ? sacc[kb][r] * scale_log2e : kNegBig;
        sacc[kb][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    bf16x8_t pf[2][2];  // [kb][r0]: P^T fragment (B operand), k-slot (hi, e) <-> acc reg r0*8 + e
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r0 = 0; r0 < 2; ++r0)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = exp2f(sacc[kb][r0 * 8 + e] - m_new);
          psum += p;
          pf[kb][r0][e] = (bf16_t)p;
        }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;

    // ---- O^T += V^T . P^T : A operand lane (d = lane&31, hi) needs V[key(hi, e)][d] ----------
    // key(hi, e) = kb*32 + 16*r0 + 4*hi + (e & 3) + 8*(e >> 2): two transpose reads of 4 keys.
    const int i16 = lane & 15;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r0 = 0; r0 < 2; ++r0) {
        const int keybase = kb * 32 + 16 * r0 + 4 * hi;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int colb = (db * 32 + 16 * ((lane >> 4) & 1) + (i16 & 3) * 4) * 2;
          const unsigned char* p0 = v_lds + (keybase + (i16 >> 2)) * kVRowB + colb;
          const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(p0));
          const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(p0 + 8 * kVRowB));
          const s16x8_t a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), pf[kb][r0], oacc[db],
                                                             0, 0, 0);
        }
      }
  }

  // ---- epilogue: normalise and store O[query][d] ------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (q_valid) {
    bf16_t* op = out + ((int64_t)(q0 + qi) * hq + head) * 128;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        // regs rg*4 .. rg*4+3 are d = db*32 + 8*rg + 4*hi + (0..3): 8 contiguous bytes
        u32x2_t w;
        w[0] = pack_bf16x2(oacc[db][rg * 4 + 0] * inv, oacc[db][rg * 4 + 1] * inv);
        w[1] = pack_bf16x2(oacc[db][rg * 4 + 2] * inv, oacc[db][rg * 4 + 3] * inv);
        *reinterpret_cast<u32x2_t*>(op + db * 32 + 8 * rg + 4 * hi) = w;
      }
  }
}

}  // namespace

extern "C" int nvl_attn_prefill_varlen(const void* q, const void* k, const void* v, int64_t k_tok_stride,
                                       int64_t v_tok_stride, const int32_t* cu_seqlens_q,
                                       const int32_t* cu_seqlens_k, const int32_t* block_tables, int64_t bt_stride,
                                       void* out, int64_t total_q, int num_seqs, int max_seqlen_q, int num_q_heads,
                                       int num_kv_heads, int block_size, int64_t num_blocks, float softmax_scale,
                                       void* stream) {
  NVL_REQUIRE(q && k && v && cu_seqlens_q && cu_seqlens_k && out, "nvl_attn_prefill_varlen: null pointer");
  NVL_REQUIRE(total_q >= 0 && num_seqs >= 0 && num_seqs <= 32768, "nvl_attn_prefill_varlen: bad sizes (total_q=%lld, num_seqs=%d)", (long long)total_q, num_seqs);
  NVL_REQUIRE(num_kv_heads > 0 && num_q_heads > 0 && num_q_heads % num_kv_heads == 0 && num_q_heads <= 65535,
              "nvl_attn_prefill_varlen: Hq=%d must be a positive multiple of Hkv=%d", num_q_heads, num_kv_heads);
  NVL_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0,
              "nvl_attn_prefill_varlen: pointers must be 16-byte aligned");
  const bool paged = block_tables != nullptr;
  if (paged) {
    NVL_REQUIRE(block_size > 0 && block_size % kKBlk == 0 && num_blocks > 0 && bt_stride > 0,
                "nvl_attn_prefill_varlen: paged K/V needs block_size %% %d == 0 (got %d)", kKBlk, block_size);
  } else {
    NVL_REQUIRE(k_tok_stride % 8 == 0 && v_tok_stride % 8 == 0 && k_tok_stride >= (int64_t)num_kv_heads * 128 &&
                    v_tok_stride >= (int64_t)num_kv_heads * 128,
                "nvl_attn_prefill_varlen: bad K/V token strides");
  }
  (void)max_seqlen_q;
  if (total_q == 0 || num_seqs == 0) return NVL_OK;
  const int64_t tiles = (total_q + kQBlk - 1) / kQBlk + num_seqs;  // upper bound on sum ceil(Lq/128)
  NVL_REQUIRE(tiles < (1ll << 31), "nvl_attn_prefill_varlen: too many query tiles");
  const size_t lds = (size_t)kKBlk * (kKRowB + kVRowB) + 4 * sizeof(int) + (size_t)(num_seqs + 1) * sizeof(int);
  const float sl2 = softmax_scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)tiles, (unsigned)num_q_heads);
  if (paged) {
    hipLaunchKernelGGL(prefill_attn_kernel<true>, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, k_tok_stride, v_tok_stride, cu_seqlens_q, cu_seqlens_k, block_tables,
                       bt_stride, (bf16_t*)out, num_seqs, num_q_heads, num_kv_heads, block_size, sl2);
  } else {
    hipLaunchKernelGGL(prefill_attn_kernel<false>, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, k_tok_stride, v_tok_stride, cu_seqlens_q, cu_seqlens_k, block_tables,
                       bt_stride, (bf16_t*)out, num_seqs, num_q_heads, num_kv_heads, block_size, sl2);
  }
  return nvl_check_launch("nvl_attn_prefill_varlen");
}
