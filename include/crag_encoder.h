/*
 * crag_encoder.h — C ABI of the hand-written HIP operators of the Qwen3-Embedding encoder lane
 * (libcrag_dense.so, gfx950).  They replace the arithmetic the reference delegates to an external
 * Triton/ONNX gateway (/root/reference/app/embeddings.py:53-59 ->
 * P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:618-649,683-716): everything of the decoder forward
 * except the plain linear layers (library GEMMs), plus the gateway's pooling / slice / L2-norm
 * post-processing (RUNBOOK:703-715).
 *
 * Conventions as in crag_dense.h: 0 = ok, negative = error (crag_last_error()); all pointers are
 * DEVICE pointers on the current device; `stream` is a hipStream_t as void*.  bf16 tensors are
 * raw uint16 storage, row-major.  Sequences are PACKED (no padding): T tokens total,
 * cu_seqlens[B+1] int32 prefix sums.
 */
#ifndef CRAG_ENCODER_H
#define CRAG_ENCODER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRAG_HEAD_DIM 128 /* Qwen3 head_dim; the attention / rope kernels are specialised for it */

/* x[t, :] = table[ids[t], :]            (embed_tokens) */
int crag_enc_embed_gather(const int32_t *ids, const uint16_t *table, uint16_t *out, int64_t n_tokens,
                          int hidden, int64_t vocab, void *stream);

/* RMSNorm with optional fused residual add (the pre-norm residual stream):
 *   s = x (+ residual_in);  residual_out = s (if non-NULL);  out = weight * bf16(s * rsqrt(mean(s^2)+eps))
 * rows x hidden, hidden % 8 == 0, hidden <= 8192. */
int crag_enc_rmsnorm(const uint16_t *x, const uint16_t *residual_in, const uint16_t *weight,
                     uint16_t *out, uint16_t *residual_out, int64_t rows, int hidden, float eps,
                     void *stream);

/* In-place per-head RMSNorm (q_norm / k_norm weights, [128]) + rotary embedding (rotate-half
 * form) on the q and k parts of the fused projection qkv[T, (hq + 2*hkv) * 128];
 * cos_sin[max_pos, 64, 2] fp32 table; positions[T] int32 (position inside its sequence). */
int crag_enc_qk_norm_rope(uint16_t *qkv, const uint16_t *q_norm_w, const uint16_t *k_norm_w,
                          const float *cos_sin, const int32_t *positions, int64_t n_tokens, int hq,
                          int hkv, float eps, void *stream);

/* V part of qkv -> Vt[hkv][128][t_pad] (keys on the fast axis, each sequence starting at a
 * 32-aligned padded offset, pads zero).  tok_of_pad[t_pad] int32: packed token of a padded slot
 * or -1.  Inside every 32-slot block the slots are stored in the order the attention kernel's PV
 * fragments consume them: stored index 16*s2 + 8*h + 4*g + r holds slot 16*s2 + 8*g + 4*h + r
 * (s2, h, g in {0,1}, r in 0..3), i.e. one lane's 8 keys of one MFMA are 16 contiguous bytes. */
int crag_enc_v_transpose(const uint16_t *qkv, uint16_t *vt, const int32_t *tok_of_pad, int64_t t_pad,
                         int hq, int hkv, void *stream);

/* Causal grouped-query flash attention over packed sequences, head_dim 128, bf16 in/out,
 * fp32 softmax and accumulation.  One workgroup per (q block of 32 rows, kv head): its 4 waves are
 * the hq/hkv = 4 query heads of the group.  blk_seq / blk_q0 [n_blocks] int32: sequence and first
 * row (inside the sequence) of every q block.  qkv rows must extend 32 rows past T (any finite or
 * non-finite content).  out[T, hq*128]. */
int crag_enc_attention(const uint16_t *qkv, const uint16_t *vt, uint16_t *out, const int32_t *cu_seqlens,
                       const int32_t *cu_pad, const int32_t *blk_seq, const int32_t *blk_q0,
                       int n_blocks, int64_t t_pad, int hq, int hkv, float scale, void *stream);

/* out[t, i] = silu(gu[t, i]) * gu[t, inter + i]   (gate | up fused projection) */
int crag_enc_swiglu(const uint16_t *gate_up, uint16_t *out, int64_t rows, int inter, void *stream);

/* Pooling + post-processing of the gateway: per sequence take the last token (mode 0) or the
 * mean over its tokens (mode 1) of the FINAL-NORMED hidden state, keep the first out_dim
 * components, L2-normalise in fp32 with max(norm, 1e-12).  For mode 0 `hidden_states` is the
 * un-normed residual stream and the final RMSNorm (weight, eps) is applied to the pooled rows
 * only; for mode 1 it must already be normed (weight may be NULL).  out[B, out_dim] fp32. */
int crag_enc_pool_normalize(const uint16_t *hidden_states, const uint16_t *final_norm_w,
                            const int32_t *cu_seqlens, float *out, int n_seqs, int hidden, int out_dim,
                            int mode, float eps, void *stream);
/* Same, with the last sub-block's output `delta` (nullable; mode 0 only) added to the residual stream for the
 * pooled rows only: hidden = bf16(hidden_states + delta) -- the model's final `hidden + mlp(...)` without a pass
 * over every token. */
int crag_enc_pool_normalize_add(const uint16_t *hidden_states, const uint16_t *delta, const uint16_t *final_norm_w,
                                const int32_t *cu_seqlens, float *out, int n_seqs, int hidden, int out_dim,
                                int mode, float eps, void *stream);
/* Last-token pooling with the pooled rows given as DATA: sequence b pools row rows[b] of hidden_states (+ delta, nullable)
 * -- what a graph replay over padded sequences needs (the real last token of every sequence) without gathering the rows
 * first. */
int crag_enc_pool_normalize_rows(const uint16_t *hidden_states, const uint16_t *delta, const uint16_t *final_norm_w,
                                 const int64_t *rows, float *out, int n_seqs, int hidden, int out_dim, float eps,
                                 void *stream);

/* Linear layer for m_rows <= 32 tokens -- ONE query per /retrieve request (retrieve.py:427) -- as a weight stream:
 * out[m_rows, n] = x[m_rows, k] @ W[n, k]^T (bf16 in, fp32 accumulate, bf16 out).
 *   x      [m_pad, k] bf16, m_pad = 16 or 32 rows allocated (rows >= m_rows are padding and never stored)
 *   wsw    the weight in MFMA A-fragment order: [n/16][k/32][lane = 16 (kk/8) + row][8] = W[16 tile + row][32 step +
 *          8 (lane >> 4) + e]  (torch: W.view(n/16, 16, k/32, 4, 8).permute(0, 2, 3, 1, 4).contiguous())
 *   epilogue 0: out [m_rows, n].
 *   epilogue 1 (gate|up projection): the rows of W are interleaved per 8 features -- tile t = gate rows 8t..8t+7,
 *          then up rows 8t..8t+7 -- and out [m_rows, n/2] = silu(gate) * up with the model's bf16 roundings
 *          (crag_enc_swiglu's arithmetic).
 * Built for the Qwen3-Embedding-4B widths: k = 2560, 4096 (epilogue 0) or 9728 (epilogue 0); k = 2560 (epilogue 1). */
int crag_enc_skinny_gemm(const uint16_t *x, const uint16_t *wsw, uint16_t *out, int m_rows, int m_pad, int n, int k,
                         int epilogue, void *stream);

/* crag_enc_qk_norm_rope and crag_enc_v_transpose in ONE launch (disjoint columns of the fused qkv rows). */
int crag_enc_qk_rope_vt(uint16_t *qkv, const uint16_t *q_norm_w, const uint16_t *k_norm_w, const float *cos_sin,
                        const int32_t *positions, int64_t n_tokens, int hq, int hkv, float eps, uint16_t *vt,
                        const int32_t *tok_of_pad, int64_t t_pad, void *stream);

/* ---- the decoder layer at 16 / 32 token rows (one short query per /retrieve request, retrieve.py:427) as five
 * launches: csrc/crag_encoder_small.hip ----
 *
 * crag_enc_small_gemm: out[m_rows, n (or n/2)] = X[m_pad, k] @ W[n, k]^T with the weights streamed once from HBM.
 *   wsw: W in tile order [n / rows_per_tile][k / 32][4][rows_per_tile][8] (a tile's k-step is one contiguous block
 *        whose 16-byte pieces are the MFMA A-operand registers of a lane);  rows_per_tile in {10, 12, 16}.
 *   norm_w != NULL (RMSNorm prologue, k = 2560): X = weight * bf16((x + delta) * rsqrt(mean((x + delta)^2) + eps)),
 *        the sum x + delta rounded to bf16 first (the residual stream); delta is required (zeros for none);
 *        res_out (nullable, must not alias x or delta) receives x + delta.
 *   norm_w == NULL: X = x; delta and res_out must be NULL.
 *   epilogue 1 (rows_per_tile 16): a tile is 8 gate rows then the 8 up rows of the same features; out [m_rows, n/2]
 *        = silu(gate) * up with crag_enc_swiglu's roundings.
 * Built for the Qwen3-Embedding-4B widths: (k 2560, prologue, 12-row tiles), (k 2560, prologue, SwiGLU, 16-row
 * tiles), (k 4096, 10-row tiles), (k 9728, 10-row tiles).  m_pad = 16 or 32 rows are read, m_rows written. */
int crag_enc_small_gemm(const uint16_t *x, const uint16_t *delta, const uint16_t *norm_w, uint16_t *res_out,
                        const uint16_t *wsw, uint16_t *out, int m_rows, int m_pad, int n, int k, int rows_per_tile,
                        int epilogue, float eps, void *stream);

/* Per-head q/k RMSNorm + RoPE (crag_enc_qk_norm_rope's arithmetic) + causal attention for n_tokens <= 32 packed
 * token rows in one launch, one workgroup per q head.  qkv[T, (hq + 2 hkv) * 128] holds the raw projections and is
 * not modified; positions[T] (position inside its sequence) also tells the sequences apart: tokens t and u belong
 * to the same sequence iff t - positions[t] == u - positions[u].  cos_sin: the [max_pos, 64, 2] table of
 * crag_enc_qk_norm_rope, or with cos_sin_by_token != 0 its rows gathered per token, [T, 64, 2] (done once per forward
 * it takes a dependent load out of every layer).  out[T, hq * 128]. */
int crag_enc_small_attention(const uint16_t *qkv, const uint16_t *q_norm_w, const uint16_t *k_norm_w,
                             const float *cos_sin, int cos_sin_by_token, const int32_t *positions, uint16_t *out,
                             int n_tokens, int hq, int hkv, float eps, float scale, void *stream);
/* The same kernel over a packed batch of n_seqs SHORT sequences (every one <= max_len <= 32 tokens; cu_seqlens[n_seqs + 1]
 * = their first rows, as crag_enc_attention takes it): one workgroup per (q head, sequence), grid (hq, n_seqs) -- q/k-norm
 * + RoPE + attention of a batch of short queries (3 to 8 of them: the gateway's batch sizes) in ONE launch instead of
 * crag_enc_qk_rope_vt + crag_enc_attention.  positions[T], cos_sin as above; qkv is not modified. */
int crag_enc_small_attention_seqs(const uint16_t *qkv, const uint16_t *q_norm_w, const uint16_t *k_norm_w,
                                  const float *cos_sin, int cos_sin_by_token, const int32_t *positions,
                                  const int32_t *cu_seqlens, int n_seqs, int max_len, uint16_t *out, int hq, int hkv,
                                  float eps, float scale, void *stream);
/* The same with the qkv projection handed over as the split-K partial tiles of crag_enc_wide_gemm_rows
 * (qkv_partial_rows[splitk][m_pad][(hq + 2 hkv) * 128] fp32): every head vector is bf16(sum over the splits, in split
 * order) -- what crag_enc_wide_reduce would have written -- summed while it is loaded. */
int crag_enc_small_attention_seqs_parts(const float *qkv_partial_rows, int splitk, int m_pad, const uint16_t *q_norm_w,
                                        const uint16_t *k_norm_w, const float *cos_sin, int cos_sin_by_token,
                                        const int32_t *positions, const int32_t *cu_seqlens, int n_seqs, int max_len,
                                        uint16_t *out, int hq, int hkv, float eps, float scale, void *stream);

/* ---- the linear layers at 32 / 64 / 96 / 128 token rows: what the gateway's batcher hands the model (RUNBOOK:304,331-334:
 * max_batch_size 8, preferred_batch_size [1, 2, 4, 8]) -- csrc/crag_encoder_wide.hip ----
 *
 * crag_enc_wide_gemm: partial[s] = X[m_pad, K_s] @ W[n, K_s]^T for the K ranges s = 0 .. splitk-1 (fp32), the weights
 * streamed once from HBM, the activations staged through LDS; 4 waves = 128 rows of W per workgroup, grid
 * (n / 128, splitk).
 *   x   [m_pad, k] bf16, m_pad = 32, 64, 96 or 128 rows allocated and READ (padding rows must hold finite values)
 *   ww  W in MFMA A-fragment order for v_mfma_f32_32x32x16_bf16: [n/32][k/16][lane = 32 (kk/8) + row][8]
 *       (torch: W.view(n/32, 32, k/16, 2, 8).permute(0, 2, 3, 1, 4).contiguous())
 *   partial  crag_enc_wide_partial_bytes(m_pad, n, splitk) bytes of scratch: the accumulators in register order
 *   n % 128 == 0, k % 128 == 0, 1 <= splitk <= k / 128.
 * crag_enc_wide_reduce: out = bf16(sum over the splits, in split order) -- one rounding, as an unsplit GEMM --,
 *   epilogue 0: out [m_rows, n];  epilogue 1 (gate|up: every 32 rows of W are 16 gate rows then the 16 up rows of the
 *   same features): out [m_rows, n/2] = silu(gate) * up with crag_enc_swiglu's roundings. */
int64_t crag_enc_wide_partial_bytes(int m_pad, int n, int splitk);
int crag_enc_wide_gemm(const uint16_t *x, const uint16_t *ww, float *partial, int m_pad, int n, int k, int splitk,
                       void *stream);
/* The unsplit form in ONE launch: out = bf16(X @ W^T) (epilogue 0) or silu(gate) * up (epilogue 1) written by the GEMM
 * kernel itself -- no partial tiles, no reduce launch; n / 128 workgroups (gate|up: 152). */
int crag_enc_wide_gemm_direct(const uint16_t *x, const uint16_t *ww, uint16_t *out, int m_rows, int m_pad, int n, int k,
                              int epilogue, void *stream);
int crag_enc_wide_reduce(const float *partial, uint16_t *out, int m_rows, int m_pad, int n, int splitk, int epilogue,
                         void *stream);
/* A split-K projection whose consumer is the residual add + RMSNorm of the next sub-block (down -> ln1): the partial
 * tiles TOKEN-MAJOR, partial_rows[splitk][m_pad][n] fp32 (crag_enc_wide_partial_bytes bytes as well), and the norm that
 * reads them -- delta = bf16(sum over the splits, in split order), the rounding of crag_enc_wide_reduce, then
 * crag_enc_rmsnorm's arithmetic with residual_in (required) / residual_out (nullable; may alias residual_in): two
 * launches (reduce, norm) become one.  hidden = n <= 4096. */
int crag_enc_wide_gemm_rows(const uint16_t *x, const uint16_t *ww, float *partial_rows, int m_pad, int n, int k, int splitk,
                            void *stream);
int crag_enc_rmsnorm_partials(const float *partial_rows, int splitk, int m_pad, const uint16_t *residual_in,
                              const uint16_t *weight, uint16_t *out, uint16_t *residual_out, int rows, int hidden,
                              float eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CRAG_ENCODER_H */
