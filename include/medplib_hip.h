/* medplib_hip.h — C ABI of libmedplib_hip.so (gfx950 / MI355X only).
 *
 * The reference (ShawnHuang497/MedPLIB) has no native code and no FFI: its hot path is PyTorch modules calling
 * cuBLAS/cuDNN through torch (SURVEY.md §0.2, §8b "Face 2").  This header is therefore the boundary a maintainer
 * would bind from the reference's Python modules (ctypes stub in INTEGRATION.md); each entry point cites the
 * reference site whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - return 0 on success, negative MP_ERR_* otherwise (never throws / aborts); mp_last_error_string() explains;
 *   - plain device pointers (tensor.data_ptr()), sizes and strides in ELEMENTS unless stated; no torch types;
 *   - no allocation, no ownership transfer; scratch comes from the caller (mp_*_workspace sizes);
 *   - asynchronous on `stream` (pass torch.cuda.current_stream().cuda_stream); thread-safe across streams;
 *   - dtype tags: MP_BF16 = 0, MP_F32 = 1; bf16 buffers are `void*`.
 */
#ifndef MEDPLIB_HIP_H
#define MEDPLIB_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

#define MP_OK 0
#define MP_ERR_SHAPE (-1)
#define MP_ERR_DTYPE (-2)
#define MP_ERR_WORKSPACE (-3)
#define MP_ERR_LAUNCH (-4)
#define MP_ERR_ARG (-5)
#define MP_BF16 0
#define MP_F32 1
/* activation tags for mp_gemm_bf16_nt */
#define MP_ACT_NONE 0
#define MP_ACT_RELU 1
#define MP_ACT_GELU 2
#define MP_ACT_QUICK_GELU 3
#define MP_ACT_SILU 4
/* fused SwiGLU epilogue: W rows = gate/up interleaved in blocks of 32; C = silu(gate)*up is [M, N/2] (LlamaMLP) */
#define MP_ACT_SWIGLU_PAIR 5

int mp_version(void);
const char* mp_arch(void);
const char* mp_last_error_string(void);
/* Measurement aid: an empty kernel named mp_profile_marker_kernel with `tag` (1..1024) workgroups — a cut mark in a rocprofv3 kernel
 * trace (bench.py brackets its timed steps with tags 1 / 2; scripts/rocpd_stats.py keeps what lies between). */
int mp_profile_marker(int tag, hipStream_t stream);

/* ---- bf16 trunk (CLIP ViT-L, Llama-7B(-MoE), SAM-Med2D ViT-B encoder) ------------------------------------------ */

/* C[M,N] = act(alpha * A[M,K] @ W[N,K]^T + bias[N]) + residual[M,N]   — every nn.Linear on the trunk
 * (HF LlamaAttention/LlamaMLP via medplib_moe_llama.py:127-141; CLIP layers via clip_encoder.py:46-57;
 * mm_projector multimodal_projector/builder.py:39-46; SAM qkv/proj/MLP image_encoder.py:273-296, common.py:13-28).
 * K % 64 == 0; m_dev (optional) is a device int overriding M (expert row counts). */
int mp_gemm_bf16_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
                    const void* residual, int64_t ldr, int M, int N, int K, int act, int out_dtype, float alpha,
                    const int* m_dev, hipStream_t stream);
/* Fused qkv projection + rotary embedding (HF LlamaAttention q/k/v_proj + apply_rotary_pos_emb, medplib_moe_llama.py:127-135, SURVEY
 * A.1): C[M, 3*hidden] (standard layout: q | k | v, heads of 128) = A[M,K] @ Wi^T with q and k rotated in the GEMM epilogue at position
 * (row % seq) + pos_offset (cos_t / sin_t: fp32 [positions, 64]).  Wi = the fused qkv weight with the rows of every q / k head
 * interleaved in blocks of 32: [dims 0..31 | 64..95 | 32..63 | 96..127] (v rows unchanged).  Bit-identical with mp_gemm_bf16_nt on the
 * plain weight followed by mp_rope_qk_bf16.  head_dim 128, hidden % 256 == 0. */
int mp_gemm_qkv_rope_bf16(const void* A, int64_t lda, const void* Wi, int64_t ldw, void* C, int64_t ldc, const float* cos_t,
                          const float* sin_t, int M, int N, int K, int seq, int pos_offset, int head_dim, hipStream_t stream);
/* The gate|up projection of a training forward (HF LlamaMLP, medplib_moe_llama.py:127-141, with peft adapters folded into K: see
 * mp_lora_down_bf16): act_out[M, N/2] = silu(gate) * up as with MP_ACT_SWIGLU_PAIR, and gu_out[M, N] = the bf16 gate|up values in W's
 * interleaved row order (what mp_swiglu_pair_bwd_bf16 reads).  Same values as mp_gemm_bf16_nt + mp_swiglu_pair_fwd_bf16. */
int mp_gemm_swiglu_keep_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* act_out, int64_t ld_act, void* gu_out,
                             int64_t ld_gu, int M, int N, int K, hipStream_t stream);
/* 320, 256 or 128: the tile size of the kernel the calling thread's last mp_gemm_bf16_nt* call dispatched to (0 before the first call).
 * Measurement aid: bench.py attributes its HIP-event samples to gemm256v3_bf16_nt_kernel / gemm_bf16_nt_kernel with it. */
int mp_gemm_last_kernel(void);
/* Tile choice of the calling thread's dense mp_gemm_bf16_nt / mp_gemm_qkv_rope_bf16 calls: 1 (default) = 320x256 tiles where the wave model
 * says they beat 256x256 tiles (M = 5112: one whole wave instead of 1.25 for every N = 4096 projection), 0 = never, 2 = whenever the call
 * is eligible (dense, bf16 out, N % 256 == 0, act NONE / QUICK_GELU / RoPE), 3 = as 2 and the 320-row kernel never splits a tail (the
 * frozen towers, which run on streams beside the decoder: a split tail's units wait for each other, and only one kernel per device may
 * do that at a time), -1 = back to the process default (MP_GEMM320).  mp_gemm_last_kernel() then reports 320. */
int mp_gemm_tile_policy(int mode);
/* The 320-row kernel's split tail (units of one tile reduce their shares together): how many shader cycles a unit waits for its siblings
 * before the tile falls back to "the last unit out sums all partials alone" — same ascending split order, so the same bits either way, and
 * forward progress never depends on the units being co-resident (another process on the GPU, a CU mask).  cycles >= 0 sets it for the
 * process (0 = never wait), < 0 only reads; returns the previous value (default 150000, MP_GEMM320_TAIL_WAIT). */
int64_t mp_gemm_tail_wait(int64_t cycles);
/* `batch` independent GEMMs at fixed strides — the per-expert SwiGLU GEMMs of DeepSpeed `Experts`
 * (call site medplib_moe_llama.py:604-614). m_dev[b] = rows routed to expert b. */
int mp_gemm_bf16_nt_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW,
                            void* C, int64_t ldc, int64_t strideC, const float* bias, int64_t strideBias, int batch,
                            int M, int N, int K, int act, int out_dtype, const int* m_dev, hipStream_t stream);

/* ... with a batched bf16 residual added after the product is rounded to bf16 (per-expert LoRA delta onto the expert output). */
int mp_gemm_bf16_nt_batched_res(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW, void* C,
                                int64_t ldc, int64_t strideC, const void* residual, int64_t ldr, int64_t strideR, int batch, int M, int N,
                                int K, const int* m_dev, hipStream_t stream);

/* The same expert GEMMs with the MOELayer's dispatch / combine einsums folded in (top-1 routing): A rows are GATHERED from the shared
 * [tokens, K] activations by a_rows[b * rows_stride + r] (= mp_moe_route_top1's slot_token), and with c_rows the C rows are SCATTERED
 * to the shared [tokens, N] output as residual[row] + c_scale[row] * bf16(acc) — the combine weights and the decoder layer's residual
 * add (sharded_moe.py MOELayer.forward; medplib_moe_llama.py:144-147).  act: MP_ACT_NONE or MP_ACT_SWIGLU_PAIR (no scatter). */
int mp_gemm_bf16_nt_batched_rows(const void* A, int64_t lda, int64_t strideA, const int* a_rows, const void* W, int64_t ldw,
                                 int64_t strideW, void* C, int64_t ldc, int64_t strideC, const int* c_rows, const float* c_scale,
                                 const void* residual, int64_t ldr, int rows_stride, int batch, int M, int N, int K, int act,
                                 const int* m_dev, hipStream_t stream);

/* y[m, n] = epilogue(x[m, :] . W[n, :]) for M <= 8 rows: the single-token decode steps of evaluate() (HF generate with a KV cache,
 * MedPLIB.py:592-606), bound by the HBM stream of W.  Same epilogues as mp_gemm_bf16_nt (incl. SWIGLU_PAIR).
 * MoE decode: w_index[m] (device int) picks the expert's matrix (W + w_index[m] * strideW) per row; then the output is
 * residual + row_scale[m] * bf16(acc), with scale 0 for rows whose row_keep[m] < 0 (capacity-dropped tokens). */
int mp_gemv_bf16(const void* x, int64_t ldx, const void* W, int64_t ldw, int64_t strideW, void* y, int64_t ldy, const float* bias,
                 const void* residual, int64_t ldr, const int* w_index, const float* row_scale, const int* row_keep, int M, int N, int K,
                 int act, int out_dtype, float alpha, hipStream_t stream);
/* mp_rmsnorm_bf16 (HF LlamaRMSNorm, medplib_moe_llama.py:121 / :286) folded into the GEMV that consumes it: y[m, :] = rmsnorm(x[m, :]) @ W^T,
 * bit-identical with the two separate calls.  Decode steps: input_layernorm -> the fused q|k|v projection, and (dense layers)
 * post_attention_layernorm -> the interleaved gate|up projection with act = SWIGLU_PAIR (y [M, N/2]).  1 <= M <= 2; K a multiple of 512
 * (1, 2, 4, 8 or 16 x 512). */
int mp_gemv_rmsnorm_bf16(const void* x, int64_t ldx, const float* norm_w, float eps, const void* W, int64_t ldw, void* y, int64_t ldy,
                         int M, int N, int K, int act, int out_dtype, hipStream_t stream);
/* The decode step's q|k|v projection with both neighbours folded in: input_layernorm (as mp_gemv_rmsnorm_bf16) in front, RoPE at position
 * *pos_dev and the KV-cache append (as mp_decode_rope_append_bf16) behind.  Writes the rotated q into the q third of qkv [M, 3*H*D] (the k and v
 * thirds are not written), the rotated k and v into the caches [B, max_len, H, D] at that position.  Bit-identical with the three launches. */
int mp_gemv_rmsnorm_rope_append_bf16(const void* x, int64_t ldx, const float* norm_w, float eps, const void* W_qkv, int64_t ldw, void* qkv,
                                     int64_t ldy, const float* cos_t, const float* sin_t, void* cache_k, void* cache_v, const int* pos_dev,
                                     int M, int heads, int head_dim, int K, int64_t cache_batch_stride, int64_t cache_seq_stride,
                                     hipStream_t stream);

/* Optional scratch for the 256x256 kernel's tail split-K (the last partial wave of tiles is cut along K so it does not hold
 * the machine for a whole tile-time): `ws` >= 64 MiB of device memory, `tickets` >= 384 ZEROED device ints.  The library never
 * allocates; without a workspace the GEMMs run unsplit.  One workspace serves one stream at a time.  Pass ws = NULL to clear. */
int mp_gemm_set_workspace(void* ws, int64_t ws_bytes, int* tickets, int n_tickets);
/* A stream that issues GEMMs concurrently with others gets its own scratch; other streams of the device use its default entry.
 * Both calls act on the CURRENT device (hipGetDevice); the directory is keyed by (device, stream) and has no size limit. */
int mp_gemm_set_stream_workspace(hipStream_t stream, void* ws, int64_t ws_bytes, int* tickets, int n_tickets);

/* Fused attention forward; variant 0 = hardware transpose-read V path, 1 = scalar-transposed V (cross-check).
 * Llama causal + key padding (HF-4.31 eager), CLIP (A.2), SAM window/global attention with decomposed
 * rel-pos bias (image_encoder.py:280-296, 381-421).  Strides: batch, sequence (head stride = D). */
int mp_attention_fwd_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss,
                          const void* V, int64_t v_sb, int64_t v_ss, void* O, int64_t o_sb, int64_t o_ss,
                          const uint8_t* key_valid, const float* rel_h, const float* rel_w, int kh, int kw, int B, int H,
                          int Sq, int Sk, int D, int causal, float scale, int variant, const int* sk_dev, hipStream_t stream);
/* (sk_dev, optional: the number of valid keys, <= Sk, read from device memory — the KV-cache length of a decode step that lives in
 * a HIP graph, where launch arguments cannot change from token to token.) */

/* LlamaRMSNorm (HF 4.31; medplib_moe_llama.py:121,138,286). */
int mp_rmsnorm_bf16(const void* x, int64_t ldx, const float* w, void* y, int64_t ldy, int64_t rows, int dim, float eps,
                    hipStream_t stream);
/* nn.LayerNorm over the last dim (CLIP eps 1e-5; SAM eps 1e-6, build_sam.py:91; LayerNorm2d in NHWC, common.py:31). */
int mp_layernorm_bf16(const void* x, int64_t ldx, const float* w, const float* b, void* y, int64_t ldy, int64_t rows,
                      int dim, float eps, hipStream_t stream);
/* Half-split RoPE on the q and k thirds of a fused [tokens, 3*H*D] buffer. */
int mp_rope_qk_bf16(void* qkv, int64_t ld, const float* cos_t, const float* sin_t, int64_t tokens, int seq, int heads,
                    int head_dim, int pos_offset, hipStream_t stream);
/* One decode step of the KV-cache path (prepare_inputs_for_generation + LlamaAttention with past_key_value, medplib_moe_llama.py:
 * 451-485; HF 4.31): RoPE of the single new token of each sequence at position *pos_dev (device), q rotated in place inside the
 * fused qkv row, rotated k and v appended to the caches at that position.  mp_advance_ints bumps the device-side counters. */
int mp_decode_rope_append_bf16(void* qkv, int64_t ld, const float* cos_t, const float* sin_t, void* cache_k, void* cache_v,
                               const int* pos_dev, int B, int heads, int head_dim, int64_t cache_batch_stride, int64_t cache_seq_stride,
                               hipStream_t stream);
int mp_advance_ints(int* p, int n, int delta, hipStream_t stream);
/* greedy next-token pick over fp32 logits (HF generate do_sample=False, MedPLIB.py:592-606). */
int mp_argmax_rows_f32(const float* x, int64_t ld, int64_t rows, int cols, int64_t* out, hipStream_t stream);
/* out = silu(gu[:, :ff]) * gu[:, ff:]  (LlamaMLP). */
int mp_swiglu_bf16(const void* gu, int64_t ldgu, void* out, int64_t ldo, int64_t rows, int ff, hipStream_t stream);
int mp_cast_f32_to_bf16(const float* x, void* y, int64_t n, hipStream_t stream);
int mp_cast_bf16_to_f32(const void* x, float* y, int64_t n, hipStream_t stream);
/* y[r,:] = x[r,:] + addend[r % period,:]  (position embeddings: CLIP, SAM image_encoder.py:152-154). */
int mp_add_rows_bf16(const void* x, const void* addend, void* y, int64_t rows, int dim, int64_t period, hipStream_t stream);
int mp_add3_bf16(const void* a, const void* b, const void* c, void* y, int64_t n, hipStream_t stream);

/* ---- fp32 trainable tail (SAM-Med2D mask decoder, text_hidden_fcs) ---------------------------------------------- */

/* C = act(alpha * op(A) op(B) + beta*C + bias); two-level batching by element strides
 * (transformer.py:185-244, mask_decoder.py:141-148,158-186, MedPLIB.py:152-164). */
int mp_sgemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C, int64_t ldc,
                 const float* bias, int M, int N, int K, float alpha, float beta, int act, int nb0, int nb1, int64_t sA0,
                 int64_t sA1, int64_t sB0, int64_t sB1, int64_t sC0, int64_t sC1, int split_k, hipStream_t stream);
int mp_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int64_t rows,
                         int dim, float eps, hipStream_t stream);
int mp_layernorm_bwd_f32(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, float* dx,
                         float* dw_accum, float* db_accum, int64_t rows, int dim, hipStream_t stream);
int mp_softmax_fwd_f32(const float* x, float* y, int64_t rows, int cols, float scale, hipStream_t stream);
int mp_softmax_bwd_f32(const float* p, const float* dp, float* dx, int64_t rows, int cols, float scale, hipStream_t stream);
int mp_add_f32(const float* a, const float* b, float* y, int64_t n, int64_t period, hipStream_t stream);
int mp_act_fwd_f32(const float* x, float* y, int64_t n, int act, hipStream_t stream);
int mp_act_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, int act, hipStream_t stream);
int mp_colsum_f32(const float* x, float* out, int64_t rows, int cols, int accumulate, hipStream_t stream);
/* ConvTranspose2d(k=2,s=2) = GEMM + pixel shuffle (mask_decoder.py:53-59 output_upscaling). */
int mp_convt2x2_shuffle_fwd_f32(const float* G, const float* bias, float* Y, int B, int h, int w, int Co, hipStream_t stream);
int mp_convt2x2_shuffle_bwd_f32(const float* dY, float* dG, int B, int h, int w, int Co, hipStream_t stream);
/* last_hidden_state[seg_token_mask] (MedPLIB.py:461) and expand_embedding (MedPLIB.py:292-308). */
int mp_gather_rows_bf16_to_f32(const void* src, int64_t ld, const int64_t* idx, float* out, int64_t n_rows, int dim,
                               hipStream_t stream);
int mp_gather_rows_f32(const float* src, const int64_t* idx, float* out, int64_t n_rows, int64_t dim, hipStream_t stream);
int mp_scale_f32(float* x, int64_t n, float s, hipStream_t stream);

/* The whole trainable fp32 tail of one direction as ONE launch: text_hidden_fcs (MedPLIB.py:152-164) -> TwoWayTransformer
 * (transformer.py:62-106,151-182,185-244) -> hypernetwork / IoU heads (mask_decoder.py:113-153), forward or backward, lowered by the host
 * (medplib_amd/tail_program.py) into a table of 256-byte op descriptors that a persistent grid executes phase by phase with grid barriers.
 * ops: n_ops x 256 bytes on the device {int type, flags, ntiles, tile_begin, M, N, K, i0..i3, pad; float f0..f3; int64 ld[12]; uint64 p[12]},
 * operand address = (slot << 56) | byte offset, 0 = absent; op types 1 GEMM, 2 REDUCE, 3 LN_FWD, 4 LN_BWD, 5 ATTN_FWD, 6 ATTN_BWD, 7 COPY2D.
 * phase_ops [n_phases + 1], phase_tiles [n_phases]: device int arrays.  slots: EIGHT base addresses in HOST memory, slots[0] == 0.
 * sync: 128 device bytes (zeroed here per call; 32-bit word 1 != 0 afterwards = a barrier gave up).  stamps: n_phases x 8 device bytes or NULL
 * (100 MHz clock per phase end).  grid: workgroups, clamped to the compute-unit count. */
int mp_tail_program_run(const void* ops, const int* phase_ops, const int* phase_tiles, int n_phases, const uint64_t* slots, void* sync,
                        void* stamps, int grid, hipStream_t stream);

/* ---- mask head: resize, losses, metrics -------------------------------------------------------------------------- */

/* Fused inference upsampler: ConvT2x2/s2(256->64) + LayerNorm2d + GELU + ConvT2x2/s2(64->32) + GELU [+ hyper_in @ upscaled]
 * in one pass over HBM (mask_decoder.py:53-59,141-148).  src [B, h*w, 256] bf16 tokens (NHWC); w1_packed [256][256] =
 * W1.permute(2,3,1,0), w2_packed [128][64] = W2.permute(2,3,1,0); up [B,32,4h,4w] bf16 (NCHW, may be null);
 * mask [B,4h,4w] f32 = sum_c hyper[b,c] * up[b,c] (may be null). */
int mp_mask_upsample_fused_bf16(const void* src, const void* w1_packed, const float* b1, const float* ln_w, const float* ln_b,
                                const void* w2_packed, const float* b2, const float* hyper, void* up, float* mask, int B, int h,
                                int w, float ln_eps, hipStream_t stream);
/* Backward of the fused upsampler + hypernetwork product (training form): recomputes the forward of every 16-token group and returns
 * dx2 [2][B*h*w][256] (the two row-parity shares of d src; the caller adds them), the operands of the two weight-gradient `tn` products
 * — dy1 [B*h*w][256] (column (kh*2+kw)*64 + c), a1 [B*h*w*4][64] and dy2 [B*h*w*4][128] (rows in (group, kh, kw, token) order, column
 * (kh2*2+kw2)*32 + c2):  dW1_packed = dy1^T @ src,  dW2_packed = dy2^T @ a1 — and part [B*h*w/8][256], one row per (group, kh) task:
 * db1[64] | dln_w[64] | dln_b[64] | db2[32] | dhyper[32] (column sums over all tasks give the first four; dhyper sums over the tasks of
 * one image).  w1_t / w2_t are the packed weights transposed ([256][256], [64][128]).  No atomics: deterministic. */
int mp_mask_upsample_fused_bwd_bf16(const void* src, const void* w1_packed, const float* b1, const float* ln_w, const float* ln_b,
                                    const void* w2_packed, const float* b2, const float* hyper, const void* w1_t, const void* w2_t,
                                    const float* dmask, float* dx2, float* dy1, float* a1, float* dy2, float* part,
                                    int B, int h, int w, float ln_eps, hipStream_t stream);
/* The bf16 operand images of the two ConvTranspose2d weights (w [Cin, Cout, 2, 2] fp32, mask_decoder.py:53-59) in one launch:
 * w?p [(kh, kw, cout), cin] = what mp_mask_upsample_fused_bf16 reads, w?t = its transpose (mp_mask_upsample_fused_bwd_bf16; may be NULL). */
int mp_upsampler_pack_bf16(const float* w1, const float* w2, void* w1p, void* w2p, void* w1t, void* w2t, int ci1, int co1, int ci2, int co2,
                           hipStream_t stream);
/* postprocess_masks (MedPLIB.py:682-701): crop window (already resolved with Python slice semantics by the caller)
 * then F.interpolate(bilinear, align_corners=False) to (out_h, out_w). */
int mp_bilinear_resize_fwd(const void* in, int in_dtype, float* out, int n, int in_h, int in_w, int crop_y0, int crop_x0,
                           int crop_h, int crop_w, int out_h, int out_w, hipStream_t stream);
int mp_bilinear_resize_bwd(const float* dout, float* din_zeroed, int n, int in_h, int in_w, int crop_y0, int crop_x0,
                           int crop_h, int crop_w, int out_h, int out_w, hipStream_t stream);
/* BCE + Dice + IoU-MSE + Focal and their weighted combination (MedPLIB.py:26-124, 515-572); out10 in the
 * reference's dict order; stats[n,8] feeds the backward.  `offsets` (device int64 [n_masks+1], element offsets into the flat
 * pred / gt buffers) makes the batch ragged — masks of different H x W, which the reference handles by looping; NULL = n_masks
 * masks of `hw` elements each. */
size_t mp_mask_losses_workspace(int n_masks);
int mp_mask_losses_fwd(const float* pred, const float* gt, const float* pred_iou, const float* ce_loss, int n_masks, int64_t hw,
                       const int64_t* offsets, float w_ce, float w_bce, float w_dice, float w_iou, float w_focal, float* stats,
                       float* out10, void* workspace, size_t workspace_bytes, hipStream_t stream);
int mp_mask_losses_bwd(const float* pred, const float* gt, const float* stats, const float* grad_scale, float* dpred,
                       float* dpred_iou, int n_masks, int64_t hw, const int64_t* offsets, float w_bce, float w_dice, float w_iou,
                       float w_focal, hipStream_t stream);
/* (sigmoid(x) > thr) and |pred|,|gt|,|and|,|or| counts (train_ds_medplib.py:702-719,750; vqa_infer.py:565-588). */
int mp_mask_threshold_iou(const void* pred, int pred_dtype, const float* gt, uint8_t* bin_out,
                          unsigned long long* counts_zeroed, int n_masks, int64_t hw, float threshold, hipStream_t stream);

/* ---- index-driven glue of the trunk ------------------------------------------------------------------------------ */

/* prepare_inputs_labels_for_multimodal (medplib_arch.py:296-527) as one gather: src_code[r] >= 0 -> embed_tokens row,
 * == INT64_MIN -> zero pad row, otherwise feats row (-1 - code).  The plan (codes, labels, masks) is built on the host. */
int mp_splice_rows_bf16(const void* embed, const void* feats, const int64_t* src_code, void* out, int64_t rows, int dim,
                        hipStream_t stream);
/* Conv2d(k=p,s=p) patch embedding as im2col (+ zero K padding) for the GEMM: CLIP 14x14, SAM 16x16
 * (image_encoder.py:424-455). */
int mp_patch_im2col(const void* img, int img_dtype, void* out, int B, int C, int H, int W, int patch, int k_padded,
                    hipStream_t stream);
/* NHWC tap-gather im2col: Adapter Conv3x3/s2 and ConvT4x4/s2 parity classes (image_encoder.py:32-37), neck Conv3x3
 * (image_encoder.py:133-149).  dy/dx are host arrays of n_taps offsets. */
int mp_im2col_nhwc_bf16(const void* x, void* out, int B, int H, int W, int C, int OH, int OW, int stride_y, int stride_x,
                        int n_taps, const int* dy, const int* dx, hipStream_t stream);
int mp_scatter_parity_bf16(const void* src, const void* add, void* dst, int B, int OH, int OW, int C, int sy, int sx, int py,
                           int px, int DH, int DW, hipStream_t stream);
/* window_partition / window_unpartition (+ shortcut add) (image_encoder.py:299-345, Block.forward :217-230). */
int mp_window_partition_bf16(const void* x, void* win, int B, int H, int W, int C, int ws, hipStream_t stream);
int mp_window_unpartition_add_bf16(const void* win, const void* shortcut, void* out, int B, int H, int W, int C, int ws,
                                   hipStream_t stream);
/* add_decomposed_rel_pos tables rel_h/rel_w from the (unscaled) q of a fused qkv buffer (image_encoder.py:381-421). */
int mp_relpos_tables_bf16(const void* qkv, int64_t ld, const float* rel_pos_h, const float* rel_pos_w, float* rel_h, float* rel_w,
                          int Bw, int heads, int hh, int ww, int head_dim, hipStream_t stream);
/* ---- the SAM-Med2D image encoder's own kernels (csrc/sam_encoder.hip; round 6).  Geometry: 16 x 16 tokens, 768 channels, heads of 64.
 * mp_sam_attention_bf16 replaces window_partition -> Attention.forward -> window_unpartition of Block.forward (image_encoder.py:217-230,
 * 280-296, 299-345) and add_decomposed_rel_pos (:381-421) on the UN-partitioned [B * grid * grid, 3 * heads * 64] qkv tensor (image order):
 * window = 14: the four 14 x 14 windows of the zero-padded 28 x 28 map; a padded token's k / v row is the bf16-rounded qkv bias (what the
 * projection of a zero row is), a padded query has no output (cropped by window_unpartition); window = 0: global attention.  The rel-pos
 * terms q . rel_pos_h[qy - ky + n - 1], q . rel_pos_w[qx - kx + n - 1] (fp32 tables [2 n - 1, 64]) are computed in the kernel.
 * out [B * grid * grid, heads * 64] in image order (the input of Attention.proj, whose GEMM then carries the shortcut add). */
int mp_sam_attention_bf16(const void* qkv, int64_t ld_qkv, const float* qkv_bias, const float* rel_pos_h, const float* rel_pos_w, void* out,
                          int64_t ld_out, int B, int heads, int grid, int window, float scale, hipStream_t stream);
/* y = LayerNorm(x [+ addend[row % period]]); with an addend the bf16 sum is also written to xsum (patch embedding + pos_embed -> blocks[0].norm1,
 * image_encoder.py:128-131, 217-218). */
int mp_sam_add_layernorm_bf16(const void* x, const void* addend, int period, void* xsum, const float* w, const float* b, float eps, void* y,
                              int rows, int dim, hipStream_t stream);
/* Block.norm2 and, in the same pass, the column sums of its output per slab of 16 rows: part [rows / 16, 768] fp32 (the numerator of
 * Adapter_Layer's AdaptiveAvgPool2d(1), image_encoder.py:24,49). */
int mp_sam_layernorm_colsum_bf16(const void* x, const float* w, const float* b, float eps, void* xn, float* part, int rows, int dim, hipStream_t stream);
/* Adapter_Layer.channel: gate [B, C] = sigmoid(W2 relu(W1 mean)), mean = the slab sums of the image / tokens (image_encoder.py:25-30,49);
 * the two Linear weights TRANSPOSED: w1t [C, hidden] = W1^T, w2t [hidden, C] = W2^T, fp32. */
int mp_sam_channel_gate_f32(const float* part, int slabs, int tokens, const float* w1t, const float* w2t, float* gate, int B, int C, int hidden,
                            hipStream_t stream);
/* im2col of Adapter_Layer.spatial[0] (Conv2d k 3, s 2, p 1) on gate * x: cols [B * (grid/2)^2, 9 * C], column order (ky, kx, c) (image_encoder.py:33,49-50). */
int mp_sam_im2col_scaled_bf16(const void* x, const float* gate, void* cols, int B, int grid, int C, hipStream_t stream);
/* the four output-parity tap gathers of Adapter_Layer.spatial[2] (ConvTranspose2d k 4, s 2, p 1) in one launch: cols4 [4, B * half^2, 4 * C]. */
int mp_sam_im2col_parity4_bf16(const void* s1, void* cols4, int B, int half, int C, hipStream_t stream);
/* The end of Block.forward for an adapter block, one pass over the rows: t = xn + y4[parity][pixel] (x + x_spatial), ad = Adapter.norm(t),
 * x_out = x + mlp + ad (image_encoder.py:52-56, 232-234), h_out = the NEXT block's norm1(x_out) (next_w null: skipped).  y4 [4, B * (grid/2)^2, dim]. */
int mp_sam_block_tail_bf16(const void* y4, const void* xn, const void* x, const void* mlp, const float* ad_w, const float* ad_b, float ad_eps,
                           const float* next_w, const float* next_b, float next_eps, void* x_out, void* h_out, int B, int grid, int dim,
                           hipStream_t stream);
/* nn.AdaptiveAvgPool1d over the token axis of a token-major [n, len_in, C] tensor: TokenCompressor 576 -> 256 and
 * MaskTokenEncoder 441 -> 64 (medplib_arch.py:67-77, 98-108). */
int mp_adaptive_avgpool_tokens_bf16(const void* x, void* out, int n, int len_in, int len_out, int C, hipStream_t stream);
/* First layer of MaskTokenEncoder: Conv2d(1, CO, k3, s2, p1) + GELU on [n, H, W] masks (bf16 or f32, rounded to bf16 like the
 * reference's cast, medplib_arch.py:103-104) -> NHWC [n, OH, OW, CO] bf16; w [CO, 9] f32, bias [CO] f32 (medplib_arch.py:84-85). */
int mp_conv3x3s2_c1_gelu_bf16(const void* img, int img_dtype, const float* w, const float* bias, void* out, int n, int H, int W,
                              int CO, hipStream_t stream);
/* extract_region_feature (medplib_arch.py:580-613): out[m, :] = mean over the points offsets[m]..offsets[m+1] of the bilinear
 * (grid_sample align_corners=True, zero padding) read-out of feature map `map_index[m]` ([h, w, C] token-major) at the
 * normalised (x, y) pairs in `xy`; each sample rounded to bf16 before the fp32 mean, like the reference's dtype round trip. */
int mp_region_point_mean_bf16(const void* fmap, const float* xy, const int64_t* offsets, const int* map_index, void* out,
                              int n_masks, int h, int w, int C, hipStream_t stream);
/* Adapter_Layer channel gate: global average pool and per-channel scale (image_encoder.py:43-47). */
int mp_token_mean_bf16(const void* x, float* out, int B, int T, int C, hipStream_t stream);
int mp_scale_channels_bf16(const void* x, const float* gate, void* y, int B, int T, int C, hipStream_t stream);
/* HF CLIPVisionEmbeddings: [cls; patches] + position embedding. */
int mp_clip_embed_bf16(const void* patch, const void* cls, const void* pos, void* out, int B, int n_patches, int C,
                       hipStream_t stream);
/* feature_select 'patch': drop the CLS row of every image (clip_encoder.py:31-39). */
int mp_copy_rows_bf16(const void* src, void* dst, int64_t rows, int dim, int rows_per_batch, int src_batch_rows, int src_row0,
                      hipStream_t stream);

/* Rows of the LAST decoder layer that something reads (the supervised rows of the filtered CE, medplib_moe_llama.py:392-408, and the <SEG> rows,
 * MedPLIB.py:456-466): that layer's MLP runs on those rows only — row-wise identical results, unread rows are not computed.
 * mp_gather_rows_bf16: out[r] = src[idx[r]] (scatter = 0) or out[idx[r]] = src[r] (scatter = 1), bf16 rows of `dim`.
 * mp_moe_filter_slots: per expert, the routed slots whose token is marked in needed[tokens] (uint8), compacted in slot order. */
int mp_gather_rows_bf16(const void* src, int64_t ld_src, const int64_t* idx, void* out, int64_t ld_out, int64_t n_rows, int dim, int scatter,
                        hipStream_t stream);
int mp_moe_filter_slots(const int* slot_token, const int* kept, const uint8_t* needed, int* slot_token_out, int* kept_out, int n_experts,
                        int capacity, hipStream_t stream);

/* ---- CE and MoE routing ------------------------------------------------------------------------------------------- */

/* per-row -log softmax(logits)[label] on fp32 logits (medplib_moe_llama.py:388-408). */
int mp_cross_entropy_rows_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t n_rows, int vocab, float* row_loss,
                              hipStream_t stream);
/* out = mean(x)*scale + add_scale*sum(add): CE mean (+ router_aux_loss_coef * sum l_aux, medplib_moe_llama.py:410-421). */
int mp_mean_plus_f32(const float* x, int64_t n, float scale, const float* add, int n_add, float add_scale, float* out,
                     hipStream_t stream);
/* DeepSpeed TopKGate: fp32 gate logits + softmax over ALL tokens incl. padding. */
int mp_moe_gate_bf16(const void* x, int64_t ldx, const float* wg, float* logits, float* gates, int64_t tokens, int dim,
                     int n_experts, hipStream_t stream);
/* DeepSpeed top1gating: argmax expert, capacity, random-token-selection from injected uniforms, slots, l_aux. */
int mp_moe_route_top1(const float* gates, const float* rts_uniform, int tokens, int n_experts, int capacity, int* expert,
                      int* slot, float* weight, int* kept_counts, long long* exp_counts, float* l_aux, int* slot_token,
                      hipStream_t stream);
/* (slot_token, optional: [n_experts * capacity], slot_token[e * capacity + s] = token in slot s of expert e — the row index the
 * expert GEMMs gather their A rows and scatter their C rows by, mp_gemm_bf16_nt_batched_rows.)
 * out[t] = x[t] for the tokens no expert took: the residual-only rows when the combine is fused into the down projection. */
int mp_moe_fill_dropped_bf16(const void* x, const int* slot, void* out, int64_t tokens, int dim, hipStream_t stream);
/* Decode rows (tokens <= 8): the post-attention LlamaRMSNorm, the TopKGate logits / softmax and top1gating in one launch —
 * bit-identical with mp_rmsnorm_bf16 + mp_moe_gate_bf16 + mp_moe_route_top1 (the three launches cost 21 us per layer of a decode
 * step, nearly all latency).  h [tokens, dim] receives the normed rows; gates (optional) [tokens, n_experts]. */
int mp_decode_norm_gate_route(const void* x, int64_t ldx, const float* ln_w, float eps, const float* wg, void* h, int64_t ldh,
                              const float* rts_uniform, int tokens, int dim, int n_experts, int capacity, float* gates, int* expert,
                              int* slot, float* weight, int* kept_counts, long long* exp_counts, float* l_aux, hipStream_t stream);
/* DeepSpeed residual MoE (MoE(use_residual=True).forward, deepspeed/moe/layer.py; off in the shipped scripts,
 * train_ds_medplib.py:131): out = x + (moe * c0 + mlp * c1), (c0, c1) = softmax of the two `coefficient` logits of the row
 * (coef [tokens, ldcoef >= 2] bf16), with the bf16 module's rounding points. */
int mp_moe_residual_mix_bf16(const void* x, const void* moe, const void* mlp, const void* coef, int64_t ldcoef, void* out,
                             int64_t tokens, int dim, hipStream_t stream);
/* DeepSpeed top2gating (sharded_moe.py, deepspeed==0.13.1): first choice = argmax gates, second = argmax of
 * logits + noise (Gumbel draws, or NULL) with the first masked; locations by cumsum in token order, second choices behind all
 * first choices; choices at location >= capacity dropped; surviving gate pair renormalised.  Entry layout of
 * expert/slot/weight ([2*tokens]): first choices at [0, tokens), second at [tokens, 2*tokens). */
int mp_moe_route_top2(const float* gates, const float* logits, const float* noise, int tokens, int n_experts, int capacity,
                      int* expert, int* slot, float* weight, int* kept_counts, long long* exp_counts, float* l_aux,
                      hipStream_t stream);
/* Stateless draws for the gate: U(0,1) (RTS, top1gating) or Gumbel(0,1) (gumbel != 0; top2gating second-expert sampling)
 * from a hash of (seed, offset + i).  DeepSpeed uses torch's generator for these: same distribution, different stream. */
int mp_gate_noise_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, int gumbel, hipStream_t stream);
/* MOELayer dispatch / combine as index gathers (replaces einsum "sec,sm->ecm" / "sec,ecm->sm"); top_k (1 or 2) entries per
 * token in the layout above. */
/* Post-attention RMSNorm and the MoE gate in one pass over the rows (LlamaRMSNorm + TopKGate's `logits = x.float() @ wg.float()^T`,
 * softmax — medplib_moe_llama.py:137-147): h = rmsnorm(x) * ln_w (bf16), logits / gates fp32 [tokens, E] computed from that
 * bf16 h.  Bit-identical with mp_rmsnorm_bf16 followed by mp_moe_gate_bf16.  n_experts = 0: the norm alone.  dim 2048 / 4096 / 8192. */
int mp_rmsnorm_gate_bf16(const void* x, int64_t ldx, const float* ln_w, float eps, void* h, int64_t ldh, const float* wg, float* logits,
                         float* gates, int64_t tokens, int dim, int n_experts, hipStream_t stream);
/* The folded-norm form of the above (config.fold_input_norm): rstd [tokens] fp32 = 1 / sqrt(mean(x^2) + eps) instead of the normalised rows — the
 * consumer GEMM reads x itself with ln_w folded into its weight's columns and multiplies by rstd in its epilogue (mp_gemm_qkv_rope_scaled_bf16,
 * mp_gemm_bf16_nt_batched_rows_scaled).  logits / gates exactly as mp_rmsnorm_gate_bf16 computes them (from the HF-rounded bf16 h, which never
 * leaves the registers): the routing does not move.  n_experts = 0: rstd alone. */
int mp_rmsnorm_gate_rstd_bf16(const void* x, int64_t ldx, const float* ln_w, float eps, const float* wg, float* logits, float* gates,
                              float* rstd, int64_t tokens, int dim, int n_experts, hipStream_t stream);
/* The two consumer GEMMs of the folded input norms: the fused qkv projection + RoPE and the experts' gate|up projection + SwiGLU on the RAW residual
 * stream, the norm weight multiplied into the weight's columns on the host (once, at load), rstd applied to the fp32 accumulators in the epilogue
 * before the projection's own bf16 rounding (LlamaRMSNorm -> q/k/v_proj, -> gate/up_proj; medplib_moe_llama.py:121-148).  320-row kernel shapes only
 * (mp_gemm_fold_ok: M >= 1024, N % 256 == 0, K % 64 == 0) — anything else is an error, the caller keeps the unfolded kernels for it. */
int mp_gemm_qkv_rope_scaled_bf16(const void* A, int64_t lda, const void* Wi, int64_t ldw, void* C, int64_t ldc, const float* cos_t,
                                 const float* sin_t, const float* row_scale, int M, int N, int K, int seq, int pos_offset, int head_dim,
                                 hipStream_t stream);
int mp_gemm_bf16_nt_batched_rows_scaled(const void* A, int64_t lda, const int* a_rows, const float* a_row_scale, const void* W, int64_t ldw,
                                        int64_t strideW, void* C, int64_t ldc, int64_t strideC, int rows_stride, int batch, int M, int N,
                                        int K, const int* m_dev, hipStream_t stream);
int mp_gemm_fold_ok(int M, int N, int K);
int mp_moe_dispatch_bf16(const void* x, int64_t ldx, const int* expert, const int* slot, void* buf, int64_t ldbuf, int64_t tokens, int dim,
                         int capacity, int top_k, hipStream_t stream);   /* buf: [E, capacity, ldbuf >= dim] slabs (ldbuf > dim: row-padded, e.g. a K-extension) */
int mp_moe_combine_bf16(const void* y, const int* expert, const int* slot, const float* weight, const void* residual, void* out,
                        int64_t tokens, int dim, int capacity, int top_k, hipStream_t stream);

/* ---- attention backward (first piece of the decoder backward for LoRA training) ------------------------ */
/* mp_attention_fwd_bf16 (transposed-formulation kernel, no rel-pos) that also returns lse2 [B*H, Sq]: the fp32 row log-sum-exp of
 * the scaled, masked scores in the log2 domain (+inf for a row with no visible key). */
int mp_attention_fwd_lse_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss, const void* V,
                              int64_t v_sb, int64_t v_ss, void* O, int64_t o_sb, int64_t o_ss, const uint8_t* key_valid, int B, int H,
                              int Sq, int Sk, int D, int causal, float scale, float* lse2, hipStream_t stream);
/* delta[B*H, Sq] = rowsum(dO o O) (fp32). */
int mp_attention_delta_bf16(const void* O, int64_t o_sb, int64_t o_ss, const void* dO, int64_t do_sb, int64_t do_ss, float* delta, int B,
                            int H, int Sq, int D, hipStream_t stream);
/* Autograd of softmax(scale q k^T + causal / key-padding mask, fp32) v (HF-4.31 LlamaAttention eager): dQ, dK, dV (bf16)
 * from dO, the forward's lse2 and delta; nothing [S, S]-sized is materialised, no atomics (each output element has one owner). */
int mp_attention_bwd_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss, const void* V, int64_t v_sb,
                          int64_t v_ss, const void* dO, int64_t do_sb, int64_t do_ss, const float* lse2, const float* delta, void* dQ,
                          int64_t dq_sb, int64_t dq_ss, void* dK, int64_t dk_sb, int64_t dk_ss, void* dV, int64_t dv_sb, int64_t dv_ss,
                          const uint8_t* key_valid, int B, int H, int Sq, int Sk, int D, int causal, float scale, hipStream_t stream);
/* The same with delta computed inside the dQ kernel from the forward's output O (no separate pass over O and dO); delta_ws is a
 * [B*H, Sq] fp32 workspace the dQ kernel fills for the dK/dV kernel that follows it on the stream.  rope_cos / rope_sin (optional, [positions, D / 2] fp32): dQ and dK are stored ROTATED by the given tables at the row's position — with the
 * forward's cos and the NEGATED sin this is the transpose of RoPE, i.e. mp_rope_qk_bf16 on the result folded into the store (the same bits). */
int mp_attention_bwd_fused_bf16(const void* Q, int64_t q_sb, int64_t q_ss, const void* K, int64_t k_sb, int64_t k_ss, const void* V,
                                int64_t v_sb, int64_t v_ss, const void* O, int64_t o_sb, int64_t o_ss, const void* dO, int64_t do_sb,
                                int64_t do_ss, const float* lse2, float* delta_ws, void* dQ, int64_t dq_sb, int64_t dq_ss, void* dK,
                                int64_t dk_sb, int64_t dk_ss, void* dV, int64_t dv_sb, int64_t dv_ss, const uint8_t* key_valid, int B,
                                int H, int Sq, int Sk, int D, int causal, float scale, const float* rope_cos, const float* rope_sin, hipStream_t stream);

/* ---- decoder backward pieces (LoRA training; train_ds_medplib.py:262-303, scripts/train_stage3.sh) -------- */
/* Autograd of LlamaRMSNorm w.r.t. its input: dx = rs * (dy*w - xhat * mean(dy*w*xhat)) [+ add], xhat = x*rs; rs_out (optional, [rows])
 * receives the row scales for the weight gradient. */
int mp_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const float* w, const void* dy, int64_t ldy, const void* add, int64_t lda, void* dx,
                        int64_t ldo, int64_t rows, int dim, float eps, float* rs_out, hipStream_t stream);
/* ... and w.r.t. its weight (`input_layernorm,post_attention_layernorm` in --sft_modules, scripts/train_stage2.sh): dw[c] =
 * sum_t dy[t, c] * bf16(x[t, c] * rs[t]), rs [rows] from mp_rmsnorm_bwd_bf16 (rs_out); partial >= ceil(rows / 256) * dim floats,
 * chunks added in ascending order. */
int mp_rmsnorm_wgrad_f32(const void* x, int64_t ldx, const void* dy, int64_t ldy, const float* rs, float* dw, float* partial,
                         int64_t partial_floats, int64_t rows, int dim, hipStream_t stream);
/* silu(gate) * up and its autograd on the gate|up GEMM output [tokens, 2*ff] whose columns are interleaved in blocks of 32 (the
 * fused weight layout); act / dact [tokens, ff]. */
/* (counts / cap, optional: the rows are capacity slabs [E, cap, .]; only the first counts[e] rows of slab e are processed) */
int mp_swiglu_pair_fwd_bf16(const void* gu, void* act, int64_t ldact, int64_t tokens, int ff, const int* counts, int cap, hipStream_t stream);
int mp_swiglu_pair_bwd_bf16(const void* gu, const void* dact, void* dgu, int64_t tokens, int ff, const int* counts, int cap, hipStream_t stream);
/* out[n, j] = scale * sum_t X[t, n] * G[t, j] (fp32 [N, R], R in {8, 16, 32}): the LoRA weight gradients dB = dY^T (x A^T) and
 * dA^T = x^T (dY B) — reads X once (p > 0: X is the UNdropped adapter input and mp_dropout_bf16's mask over the contiguous [tokens, N]
 * tensor is applied on the way; G must be readable for 16 columns per 16 ranks: the padded [tokens, 64] adapter tensors are); `partial` (>= ceil(tokens / 256) * N * R floats) holds per-chunk sums that are added in ascending
 * order (fixed summation order). */
/* keep_bits (optional, with p > 0; every kernel below that takes them): the mask as bytes [tokens, ld_bits], bit j of byte c = element 8 c + j
 * is kept — what mp_lora_down_bf16 wrote in the forward; the kernel then reads them instead of regenerating the mask from the seed (same mask,
 * same results; the generator is two 64-bit hashes and eight compares per 8 elements, which made these passes VALU-bound). */
int mp_tn_skinny_f32(const void* X, int64_t ldx, const void* G, int64_t ldg, float* out, float* partial, int64_t partial_floats,
                     int64_t tokens, int N, int R, float scale, float p, uint64_t seed, const int* rows_dev, const uint8_t* keep_bits,
                     int64_t ld_bits, hipStream_t stream);   /* rows_dev (optional): device-side row count <= tokens */
/* The two products of an adapter's backward that read its output gradient dY = X [tokens, N], in ONE pass over it (peft lora.Linear backward:
 * lora_B.weight.grad = dY^T (x A^T) and the gradient flowing into lora_A's output, dY B; call sites train_ds_medplib.py:262-303):
 * out / partial as mp_tn_skinny_f32(X, G = the forward's t) — the same bits — and dt[token, 0..63] = bf16(alpha * X[token, :] . Bt[j, :]) for the
 * rank rows of Bt (B^T padded to [>= 16 * ceil(R / 16), N]: mp_lora_pack's BT), zeros beyond, as mp_lora_down_bf16 with p = 0 computes it (here
 * the fp32 partials are summed over ceil(N / 256) column blocks instead of eight K ranges: equal to rounding).  dt_partial: >= ceil(N / 256) *
 * tokens * 16 * ceil(R / 16) floats of scratch. */
int mp_tn_skinny_down_f32(const void* X, int64_t ldx, const void* G, int64_t ldg, float* out, float* partial, int64_t partial_floats,
                          const void* Bt, int64_t ldb, void* dt, int64_t lddt, float* dt_partial, int64_t dt_partial_floats, int64_t tokens,
                          int N, int R, float scale, float alpha, hipStream_t stream);
/* The dense LlamaMLP's backward between its two input-gradient GEMMs in ONE kernel, for adapters on gate / up / down_proj (HF modeling_llama.py
 * LlamaMLP.forward via medplib_moe_llama.py:127-141; peft lora.Linear, call sites train_ds_medplib.py:262-303): mp_lora_up_add_swiglu_bwd_bf16
 * (dact, gu, dt_down, AT_down, p, seed [, keep_bits] -> dgu [tokens, 2 ff]) and mp_tn_skinny_down_f32 on that dgu (t_gu, Bt_gu -> out / partial = the
 * gate|up adapter's dB, dt_gu) — dgu is written for the main GEMM but not read back.  Same bits as the two calls. */
int mp_swiglu_bwd_skinny_f32(const void* dact, int64_t lddact, const void* gu, void* dgu, const void* dt_down, int64_t lddtd, const void* AT_down,
                             int R_down, float p, uint64_t seed, const uint8_t* keep_bits, int64_t ld_bits, const void* t_gu, int64_t ldg,
                             float* out, float* partial, int64_t partial_floats, const void* Bt_gu, int64_t ldb, void* dt_gu, int64_t lddt,
                             float* dt_partial, int64_t dt_partial_floats, int64_t tokens, int ff, int R_gu, float scale, float alpha,
                             hipStream_t stream);
/* d_logits = gconst * gscale[0] * (softmax(logits) - onehot(labels)) for the supervised rows (medplib_moe_llama.py:392-408), bf16
 * [rows, ldo] with the columns V..ldo-1 zeroed (ldo = V padded to the GEMM's K granularity). */
int mp_ce_rows_bwd(const float* logits, int64_t ldl, const int64_t* labels, const float* gscale, float gconst, void* dlogits, int64_t ldo,
                   int64_t rows, int V, hipStream_t stream);
/* out[rows[i], :] = bf16(g[i, :]): backward of the row gathers in front of lm_head / text_hidden_fcs (out pre-zeroed, rows unique). */
int mp_scatter_rows_f32_bf16(const float* g, const int64_t* rows, void* out, int64_t n, int dim, hipStream_t stream);
/* One adapter (fp32 A [r, fin], B [fout, r]) written into its group's padded bf16 GEMM operands, both orientations: A [64, fin],
 * A^T [fin, 64] at rank offset k0; B * bscale in B [W, 64], B^T [64, W] at the output rows `rows[o]` of the fused projection. */
/* Gradient counterpart of mp_lora_pack: gB[fout, r] += dB[rows[o], k0 + j], gA[r, fin] += dAT[c, k0 + i] (dB [W, R], dAT [fin, R] fp32: the fused
 * group's padded weight gradients from mp_tn_skinny_f32) -- accumulates one adapter's gradients where the optimizer reads them. */
int mp_lora_grad_unpack_f32(const float* dB, const float* dAT, const int64_t* rows, int R, int k0, int r, int fin, int fout, float* gB, float* gA,
                            hipStream_t stream);
int mp_lora_pack(const float* a, const float* b, const int64_t* rows, void* A, void* AT, void* B, void* BT, int r, int fin, int fout, int k0,
                 int W, float bscale, void* Bx, int64_t ldbx, float xscale, hipStream_t stream);
/* mp_lora_grad_unpack_f32 fed with the CHUNK PARTIALS of the two weight-gradient products (mp_tn_skinny_f32 called with out = NULL leaves
 * partial[chunks][N * R], chunks = ceil(tokens / 256)): gB += scaleB * sum_c dBp[c][rows[o], k0 + j], gA += scaleA * sum_c dATp[c][col, k0 + i],
 * ascending c — the same sums in the same order as the reduce it replaces.  W = rows of the padded B (the stride of a dBp chunk is W * R). */
int mp_lora_grad_unpack_partials_f32(const float* dBp, const float* dATp, int chunksB, int chunksA, float scaleB, float scaleA, const int64_t* rows,
                                     int R, int k0, int r, int fin, int fout, int W, float* gB, float* gA, hipStream_t stream);
/* mp_lora_pack for `n` adapters in one launch: descs = n x 104 bytes on the device {const float* a, b; const int64_t* rows; bf16* A, AT, B, BT, Bx;
 * int64 ldbx; int r, fin, fout, k0, W; float bscale, xscale; int pad}, max_elems = the largest r * fin + fout * r among them. */
int mp_lora_pack_batched(const void* descs, int n, int64_t max_elems, hipStream_t stream);
/* Backward of the adapter branch into the projection's input gradient in one pass: out = dx + dropout(bf16(dt A)) with the forward's mask
 * (dt [tokens, >= R] = scaling * dY B, AT [K, 64] = A^T padded: mp_lora_pack; p = 0: no mask).  Replaces a thin GEMM, mp_dropout_bf16 and
 * mp_add3_bf16 with the same rounding points.  R in {8, 16, 32}; out may alias dx. */
int mp_lora_up_add_bf16(const void* dt, int64_t lddt, const void* AT, const void* dx, int64_t lddx, void* out, int64_t ldo, int tokens,
                        int K, int R, float p, uint64_t seed, const int* rows_dev, const uint8_t* keep_bits, int64_t ld_bits,
                        hipStream_t stream);   /* rows_dev (optional, both kernels): device-side row count */
/* mp_lora_up_add_bf16 folded into the mp_rmsnorm_bwd_bf16 that reads its result (the adapter on gate / up_proj: its input gradient is the
 * post-attention norm's output gradient): dx = rmsnorm_bwd(x, w, dy', add) with dy' = bf16(dy + dropout(bf16(dt A))).  Bit-identical with
 * the two calls.  dim must be 4096, R 8 or 16 (the rank vectors live in LDS: 128 KiB at R = 16); keep_bits: see mp_tn_skinny_f32. */
int mp_rmsnorm_bwd_up_bf16(const void* x, int64_t ldx, const float* w, const void* dy, int64_t ldy, const void* add, int64_t lda, void* dx,
                           int64_t ldo, int64_t rows, int dim, float eps, const void* dt, int64_t lddt, const void* AT, int R, float p,
                           uint64_t seed, const uint8_t* keep_bits, int64_t ld_bits, hipStream_t stream);
/* The same followed by the SwiGLU backward in ONE pass (dense LlamaMLP with an adapter on down_proj, HF modeling_llama.py LlamaMLP.forward via
 * medplib_moe_llama.py:127-141): dgu[tokens, 2 ff] (gate|up interleaved in blocks of 32, the layout of mp_gemm_swiglu_keep_bf16's gu_out) from
 * d_act' = bf16(dact + dropout(bf16(dt A))) — bit-identical with mp_lora_up_add_bf16 followed by mp_swiglu_pair_bwd_bf16. */
int mp_lora_up_add_swiglu_bwd_bf16(const void* dt, int64_t lddt, const void* AT, const void* dact, int64_t lddact, const void* gu, void* dgu,
                                   int tokens, int ff, int R, float p, uint64_t seed, const uint8_t* keep_bits, int64_t ld_bits, hipStream_t stream);
/* The adapter's down-projection with lora_dropout inline, written as the K-extension of the projection's input (peft lora.Linear.forward,
 * tuners/lora/layer.py: result + lora_B(lora_A(dropout(x))) * scaling; call sites train_ds_medplib.py:262-303): t[token, 0..63] =
 * bf16(drop(x)[token, :] . A[j, :]) for the R rank rows of A (A: [>= 16 * ceil(R / 16), K], rows >= R zero), zeros beyond; xd (optional) =
 * drop(x), the wgrad's operand.  With t stored in columns K..K+63 of the row-padded input and scaling * B in columns K..K+63 of the weight
 * (mp_lora_pack's Bx), one mp_gemm_bf16_nt over K + 64 gives base + adapter.  Same mask as mp_dropout_bf16 on the contiguous tensor.
 * t is scaled by alpha before its rounding; with p = 0 and A = B^T the same kernel is the backward's dt = scaling * dY B (reads dY once). */
int mp_lora_down_bf16(const void* x, int64_t ldx, const void* A, int64_t lda, void* t, int64_t ldt, void* xd, int64_t ldxd, int tokens,
                      int K, int R, float p, uint64_t seed, float alpha, const int* rows_dev, float* partial, int64_t partial_floats,
                      uint8_t* keep_bits, int64_t ld_bits,   /* optional OUTPUT with p > 0 (LDS-staged form only): the mask bytes the backward kernels read */
                      hipStream_t stream);   /* partial (optional, >= 8 * tokens * 16 * ceil(R / 16) floats): room for the K-split sums of the LDS-staged kernel (K % 256 == 0) */
/* MoE layer backward, top-1 / top-2 (autograd of DeepSpeed MOELayer + top1gating / top2gating; entries = choice * tokens +
 * token; top-2 weights are the kept pair renormalised; l_aux uses the first choices' counts): the combine's d_y[e, slot] = w d_out and
 * d_w = <d_out, y[e, slot]> (d_y pre-zeroed); the gate's d_logits from d_w (chosen expert of kept tokens) and from l_aux
 * (c_aux[0] * aux_coef = d loss / d l_aux); the gate's input gradient d_x += d_logits wg. */
int mp_moe_combine_bwd_bf16(const void* dout, const void* y, const int* expert, const int* slot, const float* weight, void* dy, float* dw,
                            int64_t tokens, int dim, int capacity, int top_k, hipStream_t stream);
int mp_moe_gate_bwd_f32(const float* gates, const int* expert, const int* slot, const float* dw, const long long* first_choice_counts,
                        const float* c_aux, float aux_coef, float* dlogits, int64_t tokens, int n_experts, int top_k, hipStream_t stream);
int mp_moe_gate_dgrad_bf16(const float* dlogits, const float* wg, void* dx, int64_t tokens, int dim, int n_experts, hipStream_t stream);
/* Gradient of the token-embedding table (`embed_tokens` in --sft_modules): out[ids[u], :] = sum of g[row, :] over the rows of segment
 * u of rows_sorted (seg [n_unique + 1]), in list order; out fp32 [vocab, dim] pre-zeroed. */
int mp_embed_grad_f32(const void* g, const int64_t* rows_sorted, const int64_t* seg, const int64_t* ids, float* out, int64_t n_unique, int dim,
                      hipStream_t stream);
/* GELU on a bf16 tensor (the GEMM epilogue's function) and its derivative: the mm_projector's activation when the projector
 * trains (`mm_projector` in --sft_modules, scripts/train_stage2.sh; multimodal_projector/builder.py:39-46). */
int mp_gelu_fwd_bf16(const void* x, void* y, int64_t n, hipStream_t stream);
int mp_gelu_bwd_bf16(const void* x, const void* dy, void* dx, int64_t n, hipStream_t stream);
/* Backward of mp_region_point_mean_bf16 w.r.t. the feature maps (`region_fea_adapter` in --sft_modules, scripts/train_stage4.sh:33):
 * dfmap [n_maps, h*w, C] bf16 from dout [n_masks, C]; wt = scratch of n_masks * h * w floats; no atomics. */
int mp_region_point_mean_bwd_bf16(const float* xy, const int64_t* offsets, const int* map_index, const void* dout, void* dfmap, float* wt,
                                  int n_maps, int n_masks, int h, int w, int C, hipStream_t stream);
/* MaskTokenEncoder training (`mask_encoder` in --sft_modules, scripts/train_medplib_icl.sh:12; medplib_arch.py:80-108): layer 1 without
 * its GELU (the pre-activation is kept) and its weight [CO, 9] / bias gradients; AdaptiveAvgPool1d-over-tokens backward; col2im of
 * a k3 / s2 / p1 convolution in gather form (dcols [n*OH*OW, 9*C] tap-major -> dx [n, H, W, C]). */
int mp_conv3x3s2_c1_pre_bf16(const void* img, int img_dtype, const float* w, const float* bias, void* out, int n, int H, int W, int CO,
                             hipStream_t stream);
int mp_conv3x3s2_c1_wgrad_f32(const void* img, int img_dtype, const void* dpre, float* dw, float* db, int n, int H, int W, int CO,
                              hipStream_t stream);
int mp_adaptive_avgpool_tokens_bwd_bf16(const void* dout, void* dx, int n, int len_in, int len_out, int C, hipStream_t stream);
int mp_col2im_k3s2p1_bf16(const void* dcols, void* dx, int n, int H, int W, int C, hipStream_t stream);
/* peft lora_dropout on the adapter input: y = x * keep / (1 - p), keep from a stateless hash of (seed, index). */
int mp_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, hipStream_t stream);

/* ---- image preprocessing in front of the path (datasets/LazySupervisedDataset.py:535-556) -------- */
/* Window bounds + 22-bit fixed-point coefficients of one axis of PIL's bilinear ImagingResample (Pillow Resample.c
 * precompute_coeffs + normalize_coeffs_8bpc), which is what ResizeLongestSide.apply_image ends in
 * (model/segment_anything/utils/transforms.py:25-34).  HOST function (no GPU): bounds int[out_size][2] = (first, count), coefs
 * int[out_size][ksize], ksize = mp_pil_bilinear_ksize(in_size, out_size). */
int mp_pil_bilinear_ksize(int in_size, int out_size);
int mp_pil_bilinear_coeffs(int in_size, int out_size, int* bounds, int* coefs, int ksize);
/* One 8-bits-per-channel resampling pass over a device uint8 array viewed as [outer, in_len, inner] -> [outer, out_len, inner]
 * (horizontal pass of an HWC image: outer = H, inner = C; vertical: outer = 1, inner = W*C); bounds / coefs on the device.
 * Bit-exact with PIL. */
int mp_resample_axis_u8(const void* src, void* dst, int64_t outer, int in_len, int out_len, int64_t inner, const int* bounds,
                        const int* coefs, int ksize, hipStream_t stream);
/* uint8 HWC [h, w, C] -> float / bf16 CHW [C, size_h, size_w]: dst[c, top + y, left + x] = table[c][src[y, x, c]], pad[c] outside.
 * With the host-built 256-entry tables this is LazySupervisedDataset.preprocess (:480-505) + pad_tensor_channelwise (:446-477):
 * SAM `(x - pixel_mean) / pixel_std` then zero pad; CLIP integer-mean pad then CLIPImageProcessor rescale + normalise. */
int mp_image_table_pad_chw(const void* src, int h, int w, int C, const float* table, const float* pad, void* dst, int size_h,
                           int size_w, int top, int left, int out_dtype, hipStream_t stream);
/* uint8 HWC [n_pixels, 3] image + uint8 [n_pixels] mask -> out: where mask > 0, trunc(clip(pixel * 0.45 + tint * 0.55)) in float32
 * (three separately rounded operations, as numpy does it), else the pixel.  ICLLazySupervisedDataset._overlay_mask (:46-50). */
int mp_overlay_mask_u8(const void* img, const void* mask, void* out, int64_t n_pixels, float tint_r, float tint_g, float tint_b,
                       hipStream_t stream);

/* ---- optimizer (train_ds_medplib.py:383-420: AdamW betas (0.9,0.95), wd 0, clip 1.0) ------------------------------ */
/* out_accum[0] += sum(x^2); out_accum must hold 1 + 256 floats (out_accum[1..256] = per-block partials, summed in index order: the
 * result is bit-reproducible). */
int mp_sumsq_accum_f32(const float* x, int64_t n, float* out_accum, hipStream_t stream);
int mp_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int step, float max_norm, const float* grad_sumsq,
                      float grad_scale, hipStream_t stream);

/* ---- communication over RCCL / xGMI (one process per GPU) ----------------------------------------------------------------------
 * What the reference gets from `deepspeed.init_distributed` + ZeRO-2's bucketed gradient reduction (ds_config
 * train_ds_medplib.py:412-419: overlap_comm, reduce_bucket_size) and from DeepSpeed MOELayer's `_AllToAll` (expert dispatch /
 * combine, call sites medplib_moe_llama.py:604-614).  RCCL is bound with dlopen at the first call; without it these return
 * MP_ERR_ARG.  The communicator handle belongs to the caller (no library-side registry).
 * Bootstrap: rank 0 fills `mp_comm_unique_id_bytes()` bytes with mp_comm_unique_id and hands them to the other ranks by any means
 * (the launcher's store); every rank then calls mp_comm_init with the same bytes.  Collectives are asynchronous on `stream`. */
int mp_comm_unique_id_bytes(void);
int mp_comm_unique_id(void* out, int64_t bytes);
int mp_comm_init(int rank, int world, const void* unique_id, void** comm);
int mp_comm_destroy(void* comm);
/* what the communicator itself reports (ncclCommCount / ncclCommUserRank): the number of ranks it spans and this rank's index in it;
 * either pointer may be NULL.  bench.py prints it as `rccl_ranks` so a line can never claim more GPUs than RCCL connected. */
int mp_comm_count(void* comm, int* world, int* rank);
/* in-place SUM all-reduce of one gradient bucket (`count` elements of MP_F32 / MP_BF16) */
int mp_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype_tag, hipStream_t stream);
/* equal-split all-to-all of routed token slabs: chunk p (count_per_peer elements) of `send` goes to rank p, chunk p of `recv` arrives
 * from rank p; send != recv */
int mp_alltoall_tokens(void* comm, const void* send, void* recv, int64_t count_per_peer, int dtype_tag, hipStream_t stream);
/* variable all-to-all = grouped point-to-point messages (the expert exchange with ROUTED ROWS ONLY instead of capacity-padded slabs):
 * n_send messages (send_peer[i], send_ptr[i] = device address as an integer, send_count[i] elements) and n_recv messages likewise — HOST
 * arrays; messages between one pair of ranks match in array order.  DeepSpeed's `_AllToAll` always ships the padded [E, C, M] buffer
 * (sharded_moe.py); this is the byte-saving alternative, paid for with a host read of the counts per layer. */
int mp_alltoallv_tokens(void* comm, int n_send, const int* send_peer, const int64_t* send_ptr, const int64_t* send_count, int n_recv,
                        const int* recv_peer, const int64_t* recv_ptr, const int64_t* recv_count, int dtype_tag, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MEDPLIB_HIP_H */
