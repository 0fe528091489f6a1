// The long-prompt shape of the varlen causal GQA prefill attention (flash_attn_varlen_func, nano-vllm
// layers/attention.py:64-70; packed K / V or the paged cache, bf16): ONE wave per SIMD, 64 query rows per wave, and a main loop that is ONE
// generated asm statement (attn_prefill64_core.inc, tools/gen_prefill_asm.py): every register fixed, every lgkmcnt counted,
// 6.3 instructions per MFMA.
//
// Why (profiles/r06_prefill_pp_*.txt, r06_prefill_w64_v1_offset_blocks.txt, r06_attn_stream_probe.txt): two waves per SIMD do
// not overlap matrix and vector work on this chip (the arbitration loser runs at 0.3-0.4 rate); a lone wave hides <= 7
// issue slots per 32-cycle MFMA, and compiler-generated softmax slices between asm MFMAs cost 12. The generated stream's
// steady-state step runs 39 cycles per MFMA on synthetic data (the probe), against ~60 for the lockstep 8-wave loop.
//
// Structure: workgroup = 4 waves = 256 query rows of one (sequence, q-head); wave w owns rows 64 w .. 64 w + 63 as two 32-row
// blocks A / B. Per 64-key tile t:   phase 1  S(t+1) = K(t+1) Q^T for both blocks (a K fragment feeds two MFMAs) beside
// finish(t) (exp2, row sums, bf16 pack);   phase 2  O += V(t)^T P(t) (a V fragment feeds two MFMAs) beside start(t+1) (row
// maxima, x = S c - m against the STALE maximum: the deferred rescale of attn_prefill.hip, decided after the step's last
// P.V). Scores double-buffered in v[0:127] (P overwrites the scores it came from), O in a[0:127], Q in a[128:191], staged rows
// in a[192:223]; K and V double-buffered in LDS, one barrier per tile. This file: item lookup, LDS prologue, the operands of
// the asm statement, the tail of waves whose rows end early, the epilogue.
#include "common.h"
#include "attn_prefill64_core.inc"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kQRows = 256;  // query rows per workgroup
constexpr int kKBlk = 64;    // keys per tile
constexpr int kKRowB = 256;  // K tile row bytes in LDS (16-byte XOR swizzle)
constexpr int kVRowB = 320;  // V tile row bytes in LDS (256 + 64 pad: conflict-free tr reads)
constexpr int kKBufB = kKBlk * kKRowB;   // 16 KiB
constexpr int kVBufB = kKBlk * kVRowB;   // 20 KiB
constexpr int kLdsBytes = 2 * kKBufB + 2 * kVBufB;

typedef __attribute__((ext_vector_type(32))) float f32x32_t;

// PAGED: K / V tiles come from the paged cache [block][kv head][block_size][128] through the sequence's block-table row
// (prefix-cache hits, chunk continuations): a 64-key tile never straddles a block (block_size % 64 == 0), its byte offset
// inside either cache is looked up once per item into an LDS table that the asm loop reads one entry per step.
template <bool PAGED>
__global__ __launch_bounds__(256, 1) void prefill_w64_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t k_tok_stride,
    int64_t v_tok_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ block_tables, int64_t bt_stride, int block_size, bf16_t* __restrict__ out,
    int num_seqs, int hq, int hkv, float scale_log2e, float* __restrict__ lse, float rescale_thr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qcol = lane & 31, hi = lane >> 5;
  int head = blockIdx.x;
  const int tile_rank = blockIdx.y;

  // ---- which (sequence, 256-row q block)? lane i holds sequence i's bounds (<= 64 sequences), longest blocks first ----
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
  int qi[2], qi_c[2];
#pragma unroll
  for (int X = 0; X < 2; ++X) {
    qi[X] = row0 + X * 32 + qcol;
    qi_c[X] = qi[X] < lq ? qi[X] : lq - 1;
  }
  // tiles this wave has work in (the causal frontier of its last row); past them it only helps staging
  const int ntw = row0 < lq ? min(nt, (min(row0 + 63, lq - 1) + off) / kKBlk + 1) : 0;
  // the first tile whose start needs the causal mask: tile tn with tn * 64 + 63 > (first row of block A) + off
  const int tmask = (min(row0, lq - 1) + off + 1) / kKBlk;

  // ---- staging (prologue and tail; the asm loop stages its own tiles with the same addresses) ------------------------------
  const int srow = tid >> 4, sc16 = tid & 15;
  const int64_t kstride = PAGED ? 128 : k_tok_stride, vstride = PAGED ? 128 : v_tok_stride;
  const unsigned int koff0 = ((unsigned int)(srow * kstride) + sc16 * 8) * 2u, voff0 = ((unsigned int)(srow * vstride) + sc16 * 8) * 2u;
  const int ktile = (int)(kKBlk * kstride * 2), vtile = (int)(kKBlk * vstride * 2);      // bytes per 64-key tile
  // PAGED: byte offset of tile i inside either cache (tiles past the end repeat the last one: staged, never read)
  int64_t* tab = reinterpret_cast<int64_t*>(smem + kLdsBytes);
  auto tile_off = [&](int i) -> int64_t {
    const int kt = min(i, nt - 1) * kKBlk;
    const int blk = block_tables[(int64_t)seq * bt_stride + kt / block_size];
    return ((((int64_t)blk * hkv + kvh) * block_size + (kt % block_size)) * 128) * 2;
  };
  if constexpr (PAGED) {
    for (int i = tid; i < nt + 3; i += 256) tab[i] = tile_off(i);
  }
  __amdgpu_buffer_rsrc_t krs, vrs;
  if constexpr (!PAGED) {
    krs = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)k0 * k_tok_stride + kvh * 128), 0, (int)(lk * k_tok_stride * 2), 0x00020000);
    vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)k0 * v_tok_stride + kvh * 128), 0, (int)(lk * v_tok_stride * 2), 0x00020000);
  }
  const int kwr = srow * kKRowB + ((sc16 ^ (srow & 15)) << 4);              // chunk n: + n * 16 rows (same swizzle)
  const int vwr = 2 * kKBufB + srow * kVRowB + (sc16 << 4);
  auto stage_rows = [&](auto is_k, int tt, int buf) __attribute__((always_inline)) {   // rows past the end of the sequence read as zeros
    constexpr bool IS_K = decltype(is_k)::value;
    __amdgpu_buffer_rsrc_t rs;
    int soff;
    if constexpr (PAGED) {
      const int64_t o = __builtin_amdgcn_readfirstlane((int)(tile_off(tt) & 0xffffffffll)) & 0xffffffffll;
      const int64_t oh = __builtin_amdgcn_readfirstlane((int)(tile_off(tt) >> 32));
      rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)(IS_K ? k : v) + ((oh << 32) | o)), 0, kKBlk * 256, 0x00020000);
      soff = 0;
    } else {
      rs = IS_K ? krs : vrs;
      soff = tt * (IS_K ? ktile : vtile);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(rs, (IS_K ? koff0 : voff0) + n * ((IS_K ? ktile : vtile) >> 2), soff, 0);
      if constexpr (IS_K) *reinterpret_cast<u32x4_t*>(smem + buf * kKBufB + kwr + n * 16 * kKRowB) = x;
      else *reinterpret_cast<u32x4_t*>(smem + buf * kVBufB + vwr + n * 16 * kVRowB) = x;
    }
  };
  auto stage_k = [&](int tt, int buf) __attribute__((always_inline)) { stage_rows(std::true_type{}, tt, buf); };
  auto stage_v = [&](int tt, int buf) __attribute__((always_inline)) { stage_rows(std::false_type{}, tt, buf); };
  stage_k(0, 0);
  stage_v(0, 0);
  stage_k(1, 1);
  __syncthreads();

  f32x32_t o0, o1, o2, o3;      // O: block A d 0..63, A d 64..127, B d 0..63, B d 64..127 (a[0:127])
  float m_run[2] = {0.f, 0.f}, l_run[2] = {1.f, 1.f};
  int t = 0;
  if (ntw > 0) {
    // wave-uniform scalars travel in the lanes of one VGPR (the statement has 30 operands at most)
    const int ipk = lane == 0 ? __float_as_int(scale_log2e) : lane == 1 ? __float_as_int(rescale_thr) : lane == 2 ? ntw :
                    lane == 3 ? tmask : lane == 4 ? ktile : lane == 5 ? vtile : lane == 6 ? 2 * ktile : vtile;
    const int i16 = lane & 15;
    const int vl = 2 * kKBufB + (4 * hi + (i16 >> 2)) * kVRowB + (16 * ((lane >> 4) & 1) + (i16 & 3) * 4) * 2;
    const int kk = (hi ^ (qcol & 15)) << 4, rb = qcol * kKRowB;
    const int kma = qi_c[0] + off - 4 * hi, kmb = qi_c[1] + off - 4 * hi;
    const bf16_t* qa = q + ((int64_t)(q0 + qi_c[0]) * hq + head) * 128 + hi * 8;
    const bf16_t* qb = q + ((int64_t)(q0 + qi_c[1]) * hq + head) * 128 + hi * 8;
    if constexpr (PAGED) {
      const int ipk2 = lane == 8 ? kLdsBytes + 8 : ipk;           // LDS address of table entry 1
      const unsigned long long kb = (unsigned long long)k, vb = (unsigned long long)v;
      const unsigned int kb0 = (unsigned int)kb, kb1 = (unsigned int)(kb >> 32), vb0 = (unsigned int)vb, vb1 = (unsigned int)(vb >> 32);
      asm volatile(NVL_PF64_CORE_ASM_PAGED
                   : "={a[0:31]}"(o0), "={a[32:63]}"(o1), "={a[64:95]}"(o2), "={a[96:127]}"(o3), [ma] "=&v"(m_run[0]),
                     [mb] "=&v"(m_run[1]), [la] "=&v"(l_run[0]), [lb] "=&v"(l_run[1])
                   : [pk] "v"(ipk2), [vl] "v"(vl), [kwr] "v"(kwr), [vwr] "v"(vwr), [kma] "v"(kma), [kmb] "v"(kmb), [kk] "v"(kk),
                     [rb] "v"(rb), [ko] "v"(koff0), [vo] "v"(voff0), [qa] "v"(qa), [qb] "v"(qb), [kb0] "s"(kb0), [kb1] "s"(kb1),
                     [vb0] "s"(vb0), [vb1] "s"(vb1)
                   : "memory", "vcc", "scc", NVL_PF64_CORE_CLOBBERS);
    } else {
      asm volatile(NVL_PF64_CORE_ASM
                   : "={a[0:31]}"(o0), "={a[32:63]}"(o1), "={a[64:95]}"(o2), "={a[96:127]}"(o3), [ma] "=&v"(m_run[0]),
                     [mb] "=&v"(m_run[1]), [la] "=&v"(l_run[0]), [lb] "=&v"(l_run[1])
                   : [pk] "v"(ipk), [vl] "v"(vl), [kwr] "v"(kwr), [vwr] "v"(vwr), [kma] "v"(kma), [kmb] "v"(kmb), [kk] "v"(kk),
                     [rb] "v"(rb), [ko] "v"(koff0), [vo] "v"(voff0), [qa] "v"(qa), [qb] "v"(qb), [ksrd] "s"(krs), [vsrd] "s"(vrs)
                   : "memory", "vcc", "scc", NVL_PF64_CORE_CLOBBERS);
    }
    t = ntw;
  } else {
    __syncthreads();              // the barrier behind the first tile (attn_prefill64_core.inc)
#pragma unroll
    for (int r = 0; r < 32; ++r) { o0[r] = 0.f; o1[r] = 0.f; o2[r] = 0.f; o3[r] = 0.f; }
  }
  for (; t < nt; ++t) {           // tiles above this wave's rows: staging only, one barrier per tile like the asm steps
    stage_k(t + 2, t & 1);
    stage_v(t + 1, (t & 1) ^ 1);
    __syncthreads();
  }

  // ---- epilogue: normalise and store O[query][d] (attn_prefill.hip: permlane swaps, dwordx4 stores) -------------------
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
    for (int db = 0; db < 4; ++db) {
      const f32x32_t& oo = X == 0 ? (db < 2 ? o0 : o1) : (db < 2 ? o2 : o3);
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        const int r0 = (db & 1) * 16 + rp * 8;
        const unsigned int ax = pack_bf16x2(oo[r0 + 0] * inv, oo[r0 + 1] * inv);
        const unsigned int ay = pack_bf16x2(oo[r0 + 2] * inv, oo[r0 + 3] * inv);
        const unsigned int bx = pack_bf16x2(oo[r0 + 4] * inv, oo[r0 + 5] * inv);
        const unsigned int by = pack_bf16x2(oo[r0 + 6] * inv, oo[r0 + 7] * inv);
        const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        if (q_valid) *reinterpret_cast<u32x4_t*>(op + db * 32 + 16 * rp) = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
      }
    }
  }
}

}  // namespace

// Launch of the 64-rows-per-wave shape; arguments as validated by nvl_attn_prefill_varlen (attn_prefill.hip), which calls
// this for bf16 launches of <= 64 sequences with long prompts (packed K / V, or the paged cache). Returns 0 / NVL_E*, or
// NVL_W64_DECLINED when the launch does not fit the shape (the caller takes the 8-wave loop).
int nvl_prefill_w64_launch(const void* q, const void* k, const void* v, int64_t k_tok_stride, int64_t v_tok_stride,
                           const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, const int32_t* block_tables, int64_t bt_stride,
                           int block_size, void* out, int64_t total_q, int num_seqs, int num_q_heads, int num_kv_heads,
                           float scale_log2e, float* lse, float rescale_thr, hipStream_t s) {
  const int64_t tiles = (total_q + kQRows - 1) / kQRows + num_seqs;  // upper bound on sum ceil(Lq / 256)
  const bool paged = block_tables != nullptr;
  // paged: the per-item offset table (one entry per 64-key tile a block-table row can address, + 3) rides behind the tile buffers
  const size_t lds = kLdsBytes + (paged ? ((size_t)bt_stride * block_size / kKBlk + 3) * 8 : 0);
  if (tiles > 65535 || num_seqs > 64 || lds > 160 * 1024) return 1;      // declined
  static bool attr_done[NVL_MAX_DEVICES] = {};
  bool& done = attr_done[nvl_device_slot()];
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_w64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&prefill_w64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      nvl_set_error("nvl_attn_prefill_varlen: cannot reserve LDS for the 64-row shape");
      return NVL_ELAUNCH;
    }
    done = true;
  }
  dim3 grid((unsigned)num_q_heads, (unsigned)tiles);
  if (paged)
    hipLaunchKernelGGL(prefill_w64_kernel<true>, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       k_tok_stride, v_tok_stride, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride, block_size, (bf16_t*)out,
                       num_seqs, num_q_heads, num_kv_heads, scale_log2e, lse, rescale_thr);
  else
    hipLaunchKernelGGL(prefill_w64_kernel<false>, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       k_tok_stride, v_tok_stride, cu_seqlens_q, cu_seqlens_k, block_tables, bt_stride, block_size, (bf16_t*)out,
                       num_seqs, num_q_heads, num_kv_heads, scale_log2e, lse, rescale_thr);
  return nvl_check_launch("nvl_attn_prefill_varlen");
}
