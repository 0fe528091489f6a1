/*
 * nvl.h — C ABI of libnvl_hip.so, the MI355X (gfx950) kernels behind the
 * nano-vllm hot path (paged-KV forward step).
 *
 * Every entry point replaces one piece of arithmetic that the reference
 * delegates to a third-party CUDA dependency (flash-attn, Triton, inductor).
 * Citations are file:line in GeeeekExplorer/nano-vllm @ v0.2.0.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the null stream).
 *   - all pointers are DEVICE pointers unless a name ends in `_host`.
 *   - bf16 activations / weights / KV cache, fp32 tables and temperatures,
 *     int32 slots / context lengths / block tables / cu_seqlens,
 *     int64 token ids / positions / sampled ids. 64-bit address arithmetic
 *     everywhere (the reference's Triton store overflows int32 at 288 GB,
 *     layers/attention.py:28).
 *   - enqueue-only on `stream`: no allocation, no synchronisation, no host
 *     callbacks => every call is hipGraph-capturable. The caller owns all
 *     buffers including workspaces. (Exceptions, init-time only:
 *     nvl_allreduce_create / _connect / _status / _destroy.)
 *   - return 0 on success, negative NVL_E* on error (arguments are validated
 *     on the host before anything is launched); nvl_last_error() returns a
 *     thread-local message. Nothing throws across the ABI.
 *   - head_dim must be 128 (every Qwen3 size; models/qwen3.py:36).
 *
 * Paged KV-cache layout (ours; the reference's token-major
 * [nblk, block, Hkv, D] of engine/model_runner.py:115 is never exposed to
 * callers): per layer, K and V each are
 *        [num_blocks][num_kv_heads][block_size][128] bf16
 * i.e. head-major inside a block, so one (block, kv-head) tile is one
 * contiguous block_size*256-byte run. slot = block*block_size + offset, as in
 * the reference (engine/model_runner.py:151-161,181).
 *
 * `kv_dtype` (entry points that touch the cache): NVL_KV_BF16 = the reference's cache
 * precision (every parity run); NVL_KV_FP8 = opt-in OCP fp8 e4m3 cache (128 bytes per row,
 * K/V rounded to nearest on store, exact on load): halves the bytes the decode step is bound
 * by; an extension outside the reference's numerics. Group sizes 1, 2, 4, 8.
 */
#ifndef NVL_H_
#define NVL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVL_OK 0
#define NVL_EINVAL (-1)   /* bad argument (message in nvl_last_error) */
#define NVL_ELAUNCH (-2)  /* HIP launch error */
#define NVL_EUNSUPPORTED (-3)

#define NVL_KV_BF16 0
#define NVL_KV_FP8 1

/* ABI version: bumped on any signature change. */
int nvl_abi_version(void);
/* Thread-local description of the last error returned on this thread. */
const char* nvl_last_error(void);
/* Number of compute units of the current device (sizing persistent grids). */
int nvl_device_cu_count(void);

/* ---- RMSNorm ---------------------------------------------------------
 * Replaces RMSNorm.rms_forward (layers/layernorm.py:16-26, inductor).
 *   y = bf16( float(x) * rsqrt(mean(float(x)^2) + eps) * float(w) )
 * (single rounding: what the reference's @torch.compile'd graph computes).
 * Rows are addressed as (outer, inner): x row = x + outer*x_outer_stride +
 * inner*hidden, y likewise; this covers both [N, hidden] and the strided
 * q/k head views [N, H, 128] of models/qwen3.py:79-84. Strides in elements. */
int nvl_rmsnorm(const void* x, int64_t x_outer_stride,
                const void* weight,
                void* y, int64_t y_outer_stride,
                int64_t n_outer, int n_inner, int hidden, float eps,
                void* stream);

/* Replaces RMSNorm.add_rms_forward (layers/layernorm.py:28-40).
 *   s = float(x) + float(residual); residual <- bf16(s);
 *   y = bf16( s * rsqrt(mean(s^2)+eps) * float(w) )   (norm uses un-rounded s)
 * x, residual, y: contiguous [rows, hidden]; residual is updated in place;
 * y may alias x. */
int nvl_add_rmsnorm(const void* x, void* residual, const void* weight, void* y,
                    int64_t rows, int hidden, float eps, void* stream);

/* ---- SiLU * mul ---------------------------------------------------------
 * Replaces SiluAndMul.forward (layers/activation.py:8-11).
 *   y[r, i] = bf16( silu(float(x[r, i])) * float(x[r, inter + i]) )
 * x: [rows, 2*inter] with row stride x_row_stride; y: contiguous [rows, inter]. */
int nvl_silu_mul(const void* x, int64_t x_row_stride, void* y,
                 int64_t rows, int inter, void* stream);

/* ---- Skinny (decode-step) linear layers ---------------------------------------
 * Replace F.linear as called by LinearBase.forward and its subclasses
 * (layers/linear.py:54-156) when the row count is decode-sized (one row per
 * running sequence; the reference reaches cuBLAS through torch here), with the
 * reference's elementwise glue folded into the epilogue:
 *   mode 0  out[m, n]   = bf16( x[m, :] . W[n, :] )                        bf16 [m, n]
 *   mode 1  out[m, j]   = bf16( silu(bf16(x.W[j])) * bf16(x.W[n/2 + j]) )  bf16 [m, n/2]
 *           (= SiluAndMul over the merged gate|up projection, layers/activation.py:8-11)
 *   mode 2  part[s][m, n] = partial sums over K-slice s, fp32 [splits, m, n]; the
 *           consumer nvl_add_rmsnorm_splitk rounds bf16(sum_s part[s]) — the same
 *           rounding point as the bf16 GEMM output it replaces.
 * x: [m, k] bf16 contiguous; weight: [n, k] bf16 contiguous (torch Linear layout).
 * nvl_linear_decode_splits returns the number of K-slices the kernel will emit
 * for (m, n, k, mode) (1 for modes 0/1) or 0 when the shape is not covered — the
 * caller then keeps the library GEMM. */
int nvl_linear_decode_splits(int64_t m, int n, int k, int mode);
int nvl_linear_decode(const void* x, const void* weight, void* out,
                      int64_t m, int n, int k, int mode, int weight_layout, void* stream);
/* weight_layout (nvl_linear_decode, nvl_linear_wide): 0 = `weight` is the reference's row-major
 * [n, k] parameter; 1 = the tile-packed copy made by nvl_pack_weight_tiles (below). */

/* Decode-time linear layers on DEEP reductions (Qwen3-8B / 32B projections, full width and
 * per-rank TP shapes): the same contract as nvl_linear_decode (modes 0 / 1 / 2, reference
 * layers/linear.py:54-156 + layers/activation.py:8-11), a different decomposition: a workgroup
 * covers 128-256 output columns, stages the x tile of a 128-wide k step once in LDS for its 8
 * waves and streams its weight rows HBM -> VGPR; small-N shapes are additionally split over K
 * across workgroups.
 *   nvl_linear_wide_plan: 1 if the shape is covered (0 = not: use another path). *splits = K
 *     splits of the chosen plan (mode 2: `out` holds that many [m, n] fp32 slabs, consumed by
 *     nvl_add_rmsnorm_splitk); *workspace_bytes = scratch the call needs for modes 0 / 1 when
 *     the plan splits K (0 otherwise).
 *   nvl_linear_wide: NVL_EUNSUPPORTED when the plan query says 0. k % 128 == 0, n % 16 == 0
 *     (mode 1: n % 32 == 0). weight_layout 0: `weight` is the reference's row-major [n, k]
 *     parameter; 1: the tile-packed copy nvl_pack_weight_tiles made of it (what the engine
 *     passes: every wave load of the weight stream is then one contiguous KiB).
 *   nvl_pack_weight_tiles: packed[n/16][k/32][64][8] <- weight[n][k] (bf16, n % 16 == 0,
 *     k % 32 == 0): the 16 x 32 sub-matrix of (tile t, k-block b) is stored as the 64 lanes x
 *     8 elements of the v_mfma_f32_16x16x32_bf16 A operand (lane = 16 * (k % 32 / 8) + row % 16).
 *     Done once at model-load time (utils/loader.py:12-28 fills the row-major parameter). */
int nvl_linear_wide_plan(int64_t m, int n, int k, int mode, int* splits, size_t* workspace_bytes);
int nvl_linear_wide(const void* x, const void* weight, void* out, int64_t m, int n, int k, int mode,
                    int weight_layout, void* workspace, size_t workspace_bytes, void* stream);
int nvl_pack_weight_tiles(const void* weight, void* packed, int64_t n, int64_t k, void* stream);

/* Fused split-K reduction + residual add + RMSNorm: replaces
 * RMSNorm.add_rms_forward (layers/layernorm.py:28-40) when its input is the
 * fp32 partials of nvl_linear_decode mode 2:
 *   xs = bf16(sum_s partials[s][row]); s = float(xs) + float(residual);
 *   residual <- bf16(s); y = bf16( s * rsqrt(mean(s^2)+eps) * float(w) ).
 * partials: [splits, rows, hidden] fp32 contiguous. */
int nvl_add_rmsnorm_splitk(const float* partials, int splits, void* residual,
                           const void* weight, void* y,
                           int64_t rows, int hidden, float eps, void* stream);

/* ---- Rotary embedding -----------------------------------------------------
 * Replaces RotaryEmbedding.forward / apply_rotary_emb
 * (layers/rotary_embedding.py:6-14, 37-48): neox (half-split) rotation from a
 * fp32 table cos_sin[max_pos][128] = cos(64) || sin(64), fp32 math, one
 * rounding to bf16. x: [n_tok, n_heads, 128] with token stride x_tok_stride
 * (head stride 128); out: same shape, token stride out_tok_stride. */
int nvl_rope_neox(const int64_t* positions, const float* cos_sin, int64_t max_pos,
                  const void* x, int64_t x_tok_stride,
                  void* out, int64_t out_tok_stride,
                  int64_t n_tok, int n_heads, void* stream);

/* ---- KV-cache store ---------------------------------------------------------
 * Replaces store_kvcache_kernel / store_kvcache (layers/attention.py:10-40,
 * Triton). k, v: [n_tok, Hkv, 128] with token strides (v is a strided view of
 * the qkv GEMM output in the reference); slot_mapping[i] == -1 skips token i
 * (layers/attention.py:23). Caches in the layout described at the top. */
int nvl_store_kvcache(const void* k, int64_t k_tok_stride,
                      const void* v, int64_t v_tok_stride,
                      void* k_cache, void* v_cache,
                      const int32_t* slot_mapping,
                      int64_t n_tok, int num_kv_heads, int block_size,
                      int64_t num_blocks, int kv_dtype, void* stream);

/* ---- Fused q/k-RMSNorm -> RoPE -> KV-cache store ("KF") -------------------
 * One launch for the four reference launches of models/qwen3.py:83-85 +
 * layers/attention.py:63:   q = rope(q_norm(q)); k = rope(k_norm(k));
 * store(k, v). Numerics identical to running nvl_rmsnorm, nvl_rope_neox and
 * nvl_store_kvcache separately: q/k are rounded to bf16 after the norm and
 * again after the rotation (they are separate compiled graphs in the
 * reference). qkv: [n_tok, (Hq + 2*Hkv)*128] (q | k | v), token stride
 * qkv_tok_stride. q_norm_w / k_norm_w may both be NULL (no q/k norm: the
 * reference skips it when attention_bias is set, models/qwen3.py:68-70,82).
 * q_out: contiguous [n_tok, Hq, 128] (required). k_out: optional contiguous
 * [n_tok, Hkv, 128] (rotated keys, for the non-paged prefill path), may be
 * NULL. k_cache/v_cache may be NULL (warm-up: no cache yet,
 * layers/attention.py:62), then slot_mapping is ignored. */
int nvl_qknorm_rope_kvstore(const void* qkv, int64_t qkv_tok_stride,
                            const int64_t* positions,
                            const void* q_norm_w, const void* k_norm_w, float eps,
                            const float* cos_sin, int64_t max_pos,
                            const int32_t* slot_mapping,
                            void* q_out, void* k_out,
                            void* k_cache, void* v_cache,
                            int64_t n_tok, int num_q_heads, int num_kv_heads,
                            int block_size, int64_t num_blocks, int kv_dtype, void* stream);

/* ---- Paged decode attention ------------------------------------------------
 * Replaces flash_attn_with_kvcache as called at layers/attention.py:72-74:
 * single-query attention of q[b] over the first context_lens[b] tokens of the
 * paged cache, GQA (kv_head = q_head / (Hq/Hkv)), fp32 scores/softmax,
 * P rounded to bf16 before P.V, bf16 output. Rows with context_lens[b] == 0
 * (graph padding, engine/model_runner.py:208) produce zeros.
 * q: [batch, Hq, 128] contiguous; out: [batch, Hq, 128] contiguous.
 * Any group size Hq/Hkv from 1 to 16 (models/qwen3.py:29-38 takes any ratio:
 * Qwen3-14B is 40 / 8 = 5, at every TP degree); larger ones return NVL_EINVAL.
 * block_tables: [batch, bt_stride] int32, entries past the context unused
 * (-1 padded, engine/model_runner.py:125). HBM-bound: reads
 * context_len * 2 * Hkv * 256 bytes per sequence.
 * Workspace holds split-KV partials; size from the _workspace_bytes query
 * (depends on max_context, not on the per-step lengths => graph-safe).
 * plan (optional, may be NULL): the per-step work plan written by
 *   nvl_decode_plan for the SAME (context_lens, batch, Hq, Hkv, max_context);
 *   with it the kernel skips its own prefix scan / search (the plan is the same
 *   for every layer of a decode step: make it once, pass it to every layer).
 *   The library remembers the (batch, Hkv, max_context) each plan buffer was
 *   built for and REFUSES (NVL_EINVAL) a launch whose geometry differs, or a
 *   buffer nvl_decode_plan never filled on this device: the kernel indexes the
 *   plan's per-wave records by wave id.
 * lse (optional, may be NULL): fp32 [batch, Hq] log-sum-exp of the scaled
 *   scores, natural log — flash-attn's `softmax_lse` (return_softmax_lse=True);
 *   -inf for padded rows. */
size_t nvl_paged_attn_decode_workspace_bytes(int64_t max_batch, int num_q_heads,
                                             int64_t max_context);
int nvl_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache,
                          const int32_t* block_tables, int64_t bt_stride,
                          const int32_t* context_lens,
                          void* out,
                          int64_t batch, int num_q_heads, int num_kv_heads,
                          int block_size, int64_t num_blocks, int64_t max_context,
                          float softmax_scale,
                          void* workspace, size_t workspace_bytes,
                          int kv_dtype, const void* plan, float* lse, void* stream);

/* Per-step plan of the decode attention launches: where every wave of the
 * attention grid starts in the step's flattened (sequence, kv-head, 32-token
 * tile) work list. Depends only on context_lens / batch / head counts /
 * max_context / the device, i.e. it is identical for all layers of a decode
 * step (the reference calls flash_attn_with_kvcache once per layer,
 * layers/attention.py:72-74, and each call re-derives its own split
 * schedule). Enqueue-only, graph-capturable; `plan` is caller-owned,
 * nvl_decode_plan_bytes() bytes, 16-byte aligned.
 * shared_prefix (optional, may be NULL; ABI v5): DEVICE pointer to int32[1 + batch]. [0] = the number of leading KV
 *   blocks that the MEMBER sequences of the step have in common (identical block_tables[b][0 .. n) for every member
 *   b), [1 + b] != 0 marks sequence b as a member — what the reference's prefix cache produces when requests start
 *   with the same prompt prefix (engine/block_manager.py:58-82 hands a cache hit the block id of the earlier request;
 *   requests prefilled before the first one was registered, :110-120, hold private copies and are not members).
 *   A plan built with it makes the attention calls that consume it run a SHARED-PREFIX PASS: those blocks are read
 *   once per pack of floor(16 / (Hq/Hkv)) consecutive sequences instead of once per member, and the per-sequence kernel
 *   starts a member behind them; results are merged like any split (same value up to the fp32 summation order).
 *   (The packs are served by extra workgroups of the SAME attention launch; the plan's per-sequence grid is made that much
 *   smaller, which is one more reason why a plan is only valid for the calls it was built for.)
 *   The array is read on the device when the plan kernel AND the attention kernels run (graph replays see the current
 *   values; it must stay valid as long as the plan is used, like context_lens). [0] = 0: no shared prefix, the pass
 *   is a no-op. The count is clamped so that the tile holding a member's newest token always stays in that
 *   sequence's own share. Needs the matrix-core decode kernel (Hq/Hkv in 2 ... 16) and block_size % 128 == 0
 *   (`block_size` is only read when shared_prefix != NULL). Whether the pass is launched is a property of the plan
 *   BUFFER (remembered like its geometry): re-plan the same buffer without the pointer to switch it off. */
size_t nvl_decode_plan_bytes(void);
int nvl_decode_plan(const int32_t* context_lens, int64_t batch,
                    int num_q_heads, int num_kv_heads, int64_t max_context,
                    const int32_t* shared_prefix, int block_size, int shared_prefix_groups,
                    void* plan, size_t plan_bytes, void* stream);
/* shared_prefix_groups (ABI 6; read only when shared_prefix != NULL): the member flags are GROUP ids — rows with the
 * same id > 0 (< 32) start with the same leading blocks, rows of different ids with different ones (two system prompts
 * in one batch); [0] is the block count EVERY group shares at least (the pass covers that many blocks of each). The
 * value is the number of different groups ONE pack of floor(16 / (Hq/Hkv)) consecutive rows may hold: the pass's grid
 * has that many slots per pack (1 = a single group per step, flags 0 / 1: the ABI 5 behaviour; a pack holding more
 * groups than slots leaves the surplus groups' prefix to... nobody: size it for the worst pack, <= 8). */

/* Decode-step fusion of the three reference launches that precede the attention
 * call on a decode step — q/k RMSNorm (models/qwen3.py:82-84), rotary embedding
 * (:85) and the KV-cache store (layers/attention.py:63) — into the attention
 * kernel itself. qkv: the raw qkv GEMM output [batch, (Hq + 2*Hkv)*128] (q | k | v),
 * token stride qkv_tok_stride. Sequence b's new token is at position
 * context_lens[b] - 1 and is stored at the slot the block table gives for that
 * position (what engine/model_runner.py:176-181 puts in positions / slot_mapping);
 * rows with context_lens[b] == 0 are padding: nothing is stored, output row zero.
 * Same numerics as nvl_qknorm_rope_kvstore followed by nvl_paged_attn_decode
 * (q and k rounded to bf16 after the norm and again after the rotation). The
 * caches are written (one K and one V row per (sequence, kv-head)). */
int nvl_paged_attn_decode_fused(const void* qkv, int64_t qkv_tok_stride,
                                const void* q_norm_w, const void* k_norm_w, float eps,
                                const float* cos_sin, int64_t max_pos,
                                void* k_cache, void* v_cache,
                                const int32_t* block_tables, int64_t bt_stride,
                                const int32_t* context_lens,
                                void* out,
                                int64_t batch, int num_q_heads, int num_kv_heads,
                                int block_size, int64_t num_blocks, int64_t max_context,
                                float softmax_scale,
                                void* workspace, size_t workspace_bytes,
                                int kv_dtype, const void* plan, float* lse,
                                int qkv_splits, int64_t qkv_split_stride, void* stream);
/* qkv_splits (ABI 4): 0 = `qkv` is the bf16 output [batch, qkv_tok_stride] of the qkv projection
 * (QKVParallelLinear.forward, layers/linear.py:96-128). 1..8 = `qkv` is the fp32 split-K slab stack
 * [qkv_splits][batch][qkv_tok_stride] that nvl_linear_wide mode 2 leaves for a deep-K projection
 * (slab s at element offset s * qkv_split_stride): the attention prologue sums a row piece over the
 * slabs in slab order and rounds it to bf16 once — the value the separate slab-reduce launch would
 * have produced, bit for bit — before the norm / rotation. Matrix-core kernel only (Hq / Hkv in
 * 2 ... 16; group size 1 returns NVL_EUNSUPPORTED). */

/* ---- Varlen causal prefill attention (MFMA) ----------------------------------
 * Replaces flash_attn_varlen_func as called at layers/attention.py:67-70:
 * packed variable-length causal attention, mask bottom-right aligned (query i
 * of Lq sees keys j <= i + Lk - Lq), GQA, fp32 softmax, bf16 P, bf16 out.
 * q: [sum Lq, Hq, 128] contiguous; out likewise.
 * K/V source, exactly as the reference chooses it (layers/attention.py:65-66):
 *   block_tables == NULL : k, v are packed [sum Lk, Hkv, 128] with token
 *                          strides k_tok_stride / v_tok_stride;
 *   block_tables != NULL : k, v are the paged caches (layout at the top),
 *                          block_tables [num_seqs, bt_stride] (prefix cache /
 *                          chunked prefill continuation).
 * cu_seqlens_q / cu_seqlens_k: int32 [num_seqs + 1] (device).
 * lse (optional, may be NULL): fp32 [sum Lq, Hq] log-sum-exp of the scaled
 * scores, natural log (flash-attn's softmax_lse). */
int nvl_attn_prefill_varlen(const void* q, const void* k, const void* v,
                            int64_t k_tok_stride, int64_t v_tok_stride,
                            const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                            const int32_t* block_tables, int64_t bt_stride,
                            void* out,
                            int64_t total_q, int num_seqs, int max_seqlen_q,
                            int num_q_heads, int num_kv_heads,
                            int block_size, int64_t num_blocks,
                            float softmax_scale, int kv_dtype, float* lse, void* stream);

/* ---- Token sampler ---------------------------------------------------------
 * Replaces Sampler.forward (layers/sampler.py:7-12):
 *   argmax_i softmax(l/T)_i / E_i,  E_i ~ Exp(1) clamped >= 1e-10
 * computed in one pass as argmax_i (l_i/T - log E_i) (the softmax normaliser
 * cancels), E_i from a counter-based Philox4x32-10 keyed by
 * (seed, offset, row, column) so the draw is independent of the launch shape.
 * Extension over the reference (sampling_params.py:11 forbids it):
 * temperature == 0 => plain argmax, lowest index on ties.
 * logits: [batch, vocab] bf16, row stride logits_row_stride; temperatures fp32
 * [batch]; out int64 [batch]. `offset_dev` (optional, may be NULL) is a device
 * uint64 added to `offset`, so a captured graph can advance the stream.
 * workspace: nvl_sample_workspace_bytes(batch). */
size_t nvl_sample_workspace_bytes(int64_t max_batch);
int nvl_sample(const void* logits, int64_t logits_row_stride,
               const float* temperatures, int64_t* out,
               int64_t batch, int64_t vocab,
               uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
               const uint64_t* row_keys,
               void* workspace, size_t workspace_bytes, void* stream);
/* `row_keys` (nvl_sample, nvl_sample_shard; optional, may be NULL): device
 * uint64 [batch]. With it, row r of the batch draws as "row" = low 32 bits of row_keys[r] at
 * offset + (row_keys[r] >> 32) instead of as batch row r: a caller that passes
 * sequence_id | position << 32 gets draws that depend on (seed, sequence, position, column)
 * only — not on the sequence's row in the batch or on what else is being decoded (the
 * reference's torch generator is consumed in batch order, layers/sampler.py:11: there the
 * tokens of a request change with its neighbours). */

/* Vocab-parallel form of the sampler (the reference gathers every rank's [batch, vocab/tp]
 * logits on rank 0, layers/embed_head.py:62-65, and samples there; here each rank reduces ITS
 * shard to one {key, index} pair per row and only those 8 bytes per row travel):
 *   nvl_sample_shard : logits = this rank's [batch, vocab_local] slice whose column 0 is global
 *                      vocabulary index col_offset (multiple of 8). Same Philox stream as
 *                      nvl_sample over the full row (keyed by GLOBAL column), so merging the
 *                      shards' winners reproduces nvl_sample on the concatenated logits exactly.
 *                      best_packed: [batch][2] 32-bit words {float key bits, global index}.
 *   nvl_sample_merge : best_packed = `parts` such arrays, part_stride_bytes apart; out[b] =
 *                      index of the largest key (lowest index on ties), int64. */
int nvl_sample_shard(const void* logits, int64_t logits_row_stride,
                     const float* temperatures, void* best_packed,
                     int64_t batch, int64_t vocab_local, int64_t col_offset,
                     uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                     const uint64_t* row_keys,
                     void* workspace, size_t workspace_bytes, void* stream);
int nvl_sample_merge(const void* best_packed, int parts, int64_t part_stride_bytes,
                     int64_t* out, int64_t batch, void* stream);

/* Feed the previous step's sampled ids back as this step's input ids ON THE DEVICE:
 *   ids[i] = src_row[i] >= 0 ? prev_tokens[src_row[i]] : ids[i]
 * The reference round-trips every sampled id through the host (`.tolist()` at
 * engine/model_runner.py:218, then `torch.tensor(input_ids...)` at :183 on the next
 * step); with this node at the head of the captured decode graph the engine can
 * enqueue step N+1 before step N's ids have reached the host. src_row: int32 [n]
 * (row of the sequence in the previous decode batch, -1 = take the staged id). */
int nvl_feed_tokens(int64_t* ids, const int32_t* src_row, const int64_t* prev_tokens,
                    int64_t n, void* stream);

/* ---- Tensor-parallel collectives over xGMI (decode-sized messages) -----------------------
 * Replace dist.all_reduce after the row-parallel GEMMs (layers/linear.py:153-156) and after
 * the vocab-parallel embedding (layers/embed_head.py:41), and dist.gather of the logits
 * (layers/embed_head.py:62-65), which the reference sends through NCCL. One process per GPU;
 * every rank's comm buffer is mapped into every process with hipIpc and the kernels read the
 * peers directly over the fully connected xGMI links (csrc/comm.hip describes the protocol).
 *   create  : allocates this rank's buffer (the ONLY allocation; sized for max_bytes per call)
 *   uid     : 64-byte token (IPC memory handle) the peers need; exchange them out of band
 *   connect : uids = world x 64 bytes, rank order; maps the peers. A host-side barrier across
 *             ranks must separate connect from the first collective.
 *   run / add_rmsnorm / gather : enqueue-only on `stream`, hipGraph-capturable, results
 *             identical on every rank and run to run (fixed summation order, one bf16 rounding):
 *             run          out[rows, hidden] = sum over ranks of in (bf16; out may alias in)
 *             add_rmsnorm  the all-reduce fused with RMSNorm.add_rms_forward
 *                          (layers/layernorm.py:28-40): s = bf16(sum_r x_partial) + residual;
 *                          residual <- bf16(s); y = bf16(s * rsqrt(mean s^2 + eps) * w)
 *             gather       out[world][bytes_per_rank] = every rank's `in` (<= 4 KiB each)
 *             Every rank must issue the same sequence of calls with the same shapes.
 *   status  : NVL_OK, or NVL_ELAUNCH if a kernel gave up waiting for a peer (spins are bounded;
 *             synchronises the device — not for use inside a capture). */
int nvl_allreduce_create(int rank, int world, int64_t max_bytes, void** comm_out);
int nvl_allreduce_uid(void* comm, void* uid_out_64B);
int nvl_allreduce_connect(void* comm, const void* uids);
int64_t nvl_allreduce_max_bytes(void* comm);
/* This rank's shared input region (device pointer, nvl_allreduce_max_bytes long): a producer that writes its
 * [rows, hidden] partial sums THERE (the row-parallel GEMM's output) and passes the same pointer as `in` /
 * `x_partial` saves the kernel its copy-in phase. set_fences(0) selects the lean hand-off (per-wave store drains
 * instead of system-scope fences: sufficient because the shared buffer is uncached; default 1). */
void* nvl_allreduce_buffer(void* comm);
int nvl_allreduce_set_fences(void* comm, int on);
int nvl_allreduce_run(void* comm, const void* in, void* out, int64_t rows, int hidden, void* stream);
int nvl_allreduce_add_rmsnorm(void* comm, const void* x_partial, void* residual, const void* weight,
                              void* y, int64_t rows, int hidden, float eps, void* stream);
int nvl_allreduce_gather(void* comm, const void* in, void* out, int64_t bytes_per_rank, void* stream);
int nvl_allreduce_status(void* comm);
/* status_async (ABI 6): enqueue-only form for the serving path — one tiny kernel writes into *status_out (uint32 in
 * DEVICE memory) whether ANY rank of the group has latched a spin timeout so far (every rank's flag region is mapped
 * on every rank). Capturable: the engine puts it at the end of every decode step and copies the word to the host with
 * the step's sampled ids, so a timed-out collective raises from step() / generate() instead of returning the garbage
 * tokens that followed it (the reference's dist.all_reduce, layers/linear.py:153-156, would hang or raise in NCCL). */
int nvl_allreduce_status_async(void* comm, void* status_out, void* stream);
int nvl_allreduce_destroy(void* comm);

/* Host-side reference of the sampler's RNG (same Philox stream as the
 * kernel): fills e[n] with the Exp(1) draws for columns [col0, col0+n) of
 * `row`. Lets tests replay a GPU draw bit-exactly on the CPU. */
void nvl_sample_exponentials_host(uint64_t seed, uint64_t offset, int64_t row,
                                  int64_t col0, int64_t n, float* e_host);

#ifdef __cplusplus
}
#endif
#endif /* NVL_H_ */
