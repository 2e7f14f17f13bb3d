/* vlr.h - C ABI of libvlr_hip.so, the MI355X (gfx950) DPO-step library.
 *
 * Drop-in boundary.  The reference (TideDra/VL-RLHF) has no FFI: its hot path is Python that
 * bottoms out in torch/transformers/trl native code.  These entry points are what a binding for that path binds
 * instead; each cites the reference interface (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless said otherwise; "bf16" tensors are raw uint16 bit patterns;
 *   - row-major, leading dimensions (`ld*`) in ELEMENTS; no allocation inside: the caller passes workspaces;
 *   - every function enqueues on `stream` and returns immediately: 0 = ok, non-zero = error, message from
 *     vlr_last_error() (thread-local).  Argument errors map to Python ValueError in the host mirror;
 *   - thread-compatible, not thread-safe per stream; one process per GPU.
 */
#ifndef VLR_H
#define VLR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vlr_stream_t; /* == hipStream_t */

const char* vlr_last_error(void);
int vlr_abi_version(void);
/* in-library kernel timing (HIP events on the launch stream): kernels 0 gemm NT, 1 gemm NN, 2 gemm TN, 3 attention fwd,
 * 4 attention bwd, 5 the 256x256 eight-phase GEMM kernel alone (the sub-launches of 0-2).  vlr_prof_collect fills out[k*3 + {0,1,2}] = {launches, total ms, algorithmic FLOPs}.
 * vlr_prof_enable(0) stops; (N >= 1) starts and brackets one launch in N, picked pseudo-randomly per kernel id - launches and FLOPs
 * stay exact, `total ms` is the sampled mean x launches (an event pair is two queue packets around the kernel: N = 1 perturbs the step). */
int vlr_prof_enable(int on);
int vlr_prof_collect(double* out_host, int n_kernels);

/* ---- GEMM (replaces torch.nn.functional.linear / its autograd on cuBLAS: every nn.Linear under
 *      LlavaForRL.forward, src/vlrlhf/models/Llava/__init__.py:178,191,232) --------------------------------------
 * layout 0 (NT): C[M,N] = A[M,K] . B[N,K]^T         forward   y = x W^T
 * layout 1 (NN): C[M,N] = A[M,K] . B[K,N]           dgrad     dx = dy W
 * layout 2 (TN): C[M,N] = A[K,M]^T . B[K,N]         wgrad     dW = dy^T x
 * epilogue: v = act(acc + bias[n]) + residual[m,n] (+ C[m,n] if accumulate); act 0 none, 1 quick_gelu, 2 gelu(erf).
 * C is bf16 (out_f32 = 0) or fp32. */
int vlr_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* bias, const void* residual, int M,
                  int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate, int out_f32,
                  vlr_stream_t stream);
/* same with v = act(alpha * acc + bias) + ...  (peft LoRA merge W_eff = W + (lora_alpha / r) * B A; call site of the
 * adapters: /root/reference src/vlrlhf/utils/auto_load.py:559-571) */
int vlr_gemm_bf16_scaled(int layout, const void* A, const void* B, void* C, const void* bias, const void* residual, int M,
                         int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate, int out_f32,
                         float alpha, vlr_stream_t stream);
/* Two weight-gradient GEMMs of one decoder layer in one launch (layout 2 of vlr_gemm_bf16, both over the same K token rows):
 * C0 [M0][N0] = A0^T B0 and C1 [M1][N1] = A1^T B1 (A_i [K][lda_i], B_i [K][ldb_i], bf16; accumulate as in vlr_gemm_bf16).  The output tiles
 * of both problems form ONE persistent grid, so their tile counts add up before they are rounded to whole rounds of the CUs - the
 * dW of gate|up and of down_proj of LLaMA-7B are 6 + 3 rounds apart and 8 together.  Same arithmetic and results as two vlr_gemm_bf16
 * calls (which it falls back to when the shapes do not qualify).  Replaces two of the cuBLAS wgrad calls under autograd's backward of
 * transformers LlamaMLP (call site src/vlrlhf/models/Llava/__init__.py:232-243). */
int vlr_gemm_bf16_tn_pair(const void* A0, const void* B0, void* C0, int M0, int N0, int lda0, int ldb0, int ldc0, const void* A1,
                          const void* B1, void* C1, int M1, int N1, int lda1, int ldb1, int ldc1, int K, int accumulate,
                          vlr_stream_t stream);
/* fp32 residual stream (vlr_llama_cfg.resid_f32): C fp32 [M][ldc] = A . B (layout as above) + residual fp32 [M][ldr] (NULL: none);
 * in place (C == residual) is allowed.  Replaces the `hidden_states = residual + hidden_states` adds of transformers
 * LlamaDecoderLayer.forward (call site src/vlrlhf/models/Llava/__init__.py:232) without the bf16 rounding of the sum. */
int vlr_gemm_bf16_f32res(int layout, const void* A, const void* B, float* C, const float* residual, int M, int N, int K,
                         int lda, int ldb, int ldc, int ldr, vlr_stream_t stream);
/* Switches of the persistent (continuous-pipeline) GEMM launches, a sum of: 8 = the adapter-segment K tiles take the general staging
 * path, 16 = the two wave groups of a workgroup run their epilogues one after the other (the order before round 4; bit-identical
 * results), 32 = the shared-panel tile map (csrc/gemm_tilemap.h, round 5: the XCD blocks of a round are stacked into one super-block so
 * that the eight XCDs share panels through the Infinity Cache; same results, other order of the output tiles).  Production = 32.
 * -1 = back to the default (environment VLR_GEMM_SCHED, else 32).  The tile schedules 1-7 of ABI v5 (stream-K tail, XCD rotation, XCD
 * round barrier) measured slower or neutral and are rejected with VLR_ERR_ARG. */
int vlr_gemm_set_sched(int mode);
/* Diagnostics only: the tile timeline of the persistent GEMM launches.  With a non-NULL buffer of >= 256 KiB every later launch of the
 * 256x256 continuous-pipeline kernel leaves, for each workgroup b, words [b*256 + 4*i .. +3] = {tile id, and the 100 MHz clock after the
 * first K tile / after the K loop / after the epilogue's stores were issued} of its i-th tile, and [b*256 + 252 ..] = {tiles, start
 * clock, K tiles per tile}.  Recorded only by the library built with -DVLR_GEMM_TRACE (`build_hip.py --trace`,
 * tools/gemm_tile_trace.py); the production build returns VLR_ERR_ARG.  (NULL, 0) switches it off. */
int vlr_gemm_set_trace(void* buf, long bytes);
/* Optional fp32 scratch for split-K: problems with few output tiles and a long reduction (the LoRA adapter gradients;
 * the ragged last tile rows of the decoder GEMMs) are split along K into fp32 partials and reduced by a second kernel
 * that applies the epilogue.  Without it they run un-split.  The buffer is cut in 128 MiB slots (at most eight; below 256 MiB in
 * total: 64 MiB slots, which cover the split-K partials of the 7B shapes), one per
 * distinct stream that launches such a GEMM (policy pass, reference pass on a side stream, ...); further streams run un-split.
 * Registering again forgets the stream assignment.  (NULL, 0) unregisters. */
int vlr_gemm_set_splitk_workspace(void* workspace, long bytes);
/* Fused forward projections of the decoder layer (transformers LlamaMLP / LlamaAttention; call site
 * src/vlrlhf/models/Llava/__init__.py:232).  The elementwise op that follows the projection runs on the fp32 accumulators
 * in the GEMM epilogue - one rounding to bf16 instead of two and no extra pass over the tensor.
 *  vlr_gemm_swiglu  : gu [M][2I] = x [M][K] . wgu[2I][K]^T (gate | up), act [M][I] = silu(gate) * up; store_gu = 0 skips writing
 *                     gu where the fused kernel runs (no-grad passes; gu must still be a valid scratch buffer).
 *  vlr_gemm_qkv_rope: qkv [M][N] = x . wqkv[N][K]^T with rotate-half RoPE by pos[row] on the first rope_cols columns (the q and k
 *                     heads; head_dim 128 takes the fused path, other sizes the plain GEMM + vlr_rope). */
int vlr_gemm_swiglu(const void* x, const void* wgu, void* gu, void* act, int M, int I, int K, int ldx, int store_gu,
                    vlr_stream_t stream);
int vlr_gemm_qkv_rope(const void* x, const void* wqkv, void* qkv, const int* pos, const float* cos_t, const float* sin_t, int M,
                      int N, int rope_cols, int K, int ldx, int head_dim, int max_pos, vlr_stream_t stream);
/*  vlr_gemm_qkv_rope_bias: the same with a bias [N] on the projection (Qwen c_attn, QwenVL/modeling_qwen.py:104), NULL = none */
int vlr_gemm_qkv_rope_bias(const void* x, const void* wqkv, const void* bias, void* qkv, const int* pos, const float* cos_t,
                           const float* sin_t, int M, int N, int rope_cols, int K, int ldx, int head_dim, int max_pos, vlr_stream_t stream);
/*  vlr_gemm_swiglu_bwd: backward of the MLP's first half in one pass - d act = dy [M][H] . wdown [H][I] stays in the
 *                     accumulators and gu [M][2I] (gate | up of the forward) is replaced in place by d gate | d up.  dact_ws
 *                     [M][I] is scratch for the rows / shapes that take the plain GEMM + vlr_swiglu_bwd. */
int vlr_gemm_swiglu_bwd(const void* dy, const void* wdown, void* gu_inout, void* dact_ws, int M, int I, int H,
                        vlr_stream_t stream);
/* The same projections with the LoRA adapters of the group riding the K loop of the base GEMM (peft lora.Linear.forward,
 * result = base(x) + lora_B(lora_A(dropout(x))) * scaling; reference targets src/vlrlhf/models/Llava/__init__.py:273-286):
 * after the K tiles of x . W^T the kernel runs on into u . Bl^T, u [M][ldu] = scaling * dropout_t(x) A_t^T of the group's
 * targets side by side (r columns each, written by the caller's lora_A GEMMs), Bl = the lora_B rows of the group [N][r].  The
 * adapter term is block-diagonal: an output tile of target t multiplies only u_t.  One rounding, fused SwiGLU / RoPE / residual
 * epilogues as above; shapes / rows the persistent kernel does not take run base GEMM + one skinny GEMM per target.
 *  vlr_gemm_lora         : y [M][ldy] = x [M][K] . W[N][K]^T + u . Bl^T (+ residual [M][ldr])          (o_proj, down_proj)
 *  vlr_gemm_swiglu_lora  : gate | up with u = u_gate | u_up, Bl = [B_gate ; B_up] [2I][r]; gu is always stored
 *  vlr_gemm_qkv_rope_lora: q | k | v with u = u_q | u_k | u_v, Bl = [B_q ; B_k ; B_v]; q_cols + 2 * kv_cols = N.  kv_cols = 0: ONE
 *                          adapter over the whole fused projection (u [M][r], Bl [N][r]).  bias [N] (NULL = none) is added before
 *                          the rotation. */
/*  vlr_gemm_dropout_acc  : dx [M][in] += scaling / (1 - p) * mask .* (v [M][ldv] . A [r][in]) - the input-gradient term of one
 *                          target under lora_dropout; the mask is regenerated from (seed, row * in + col) as in vlr_dropout and the
 *                          product never reaches HBM (scratch [M][in] only for shapes the fused kernel does not take) */
/*  vlr_gemm_grouped     : `groups` (<= 8) equally shaped skinny GEMMs as ONE launch - group g computes C + g gC = alpha * (A + g gA) . (B + g gB)
 *                          (layout as vlr_gemm_bf16, strides in elements, bf16 out, optional += C) - the per-target lora_A / lora_B products of a
 *                          fused group (q, k, v / gate, up), split along K when that fills the chip.  mask_on 1 (NT) / 2 (TN): the activation
 *                          operand (A rows / B rows, dense [rows][mask_ld]) is multiplied by the keep mask of vlr_dropout(seed + g) while it is
 *                          staged - lora_dropout without a dropped copy of x; the caller folds 1 / (1 - p) into alpha. */
int vlr_gemm_grouped(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int groups,
                     long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed, float p_drop, int mask_ld,
                     vlr_stream_t stream);
int vlr_gemm_dropout_acc(const void* v, int ldv, const void* A, void* dx, void* scratch, int M, int in, int r, float p,
                         uint64_t seed, float scale, vlr_stream_t stream);
/*  vlr_gemm_dropout_acc_multi: the same for the n targets that share dx, in ONE pass over it: dx (+)= scaling / (1 - p) * sum_t mask_t .*
 *                          (v_t . A_t), v [M][ldv] = the n blocks of r columns side by side, A = the n stacked [r][in] matrices, mask_t of
 *                          vlr_dropout(seed + t); accumulate = 0 WRITES dx (the adapter term alone) */
int vlr_gemm_dropout_acc_multi(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p, uint64_t seed,
                               float scale, int accumulate, vlr_stream_t stream);
/*  ..._bits (ABI v5)     : the three calls above with the keep masks DRAWN BEFOREHAND by vlr_dropout_bits (packed, bit e of byte i =
 *                          element 8 i + e of the dense [rows][mask_ld] operand; the mask of group / target g starts at bits + g *
 *                          gstride bytes, seed + g as before): the forward, the dA and the dx kernels of a target all need the same mask
 *                          and the packed form lets the masked operand travel by LDS-DMA (gemm128p.hip masks the fragments it reads).
 *                          vlr_gemm_grouped_bits: mask_on 1 / 2 read the row-major masks, mask_on 3 (layout 2) the K-tile-blocked transposed
 *                          ones of vlr_dropout_bits2.  bits = NULL: hash in the kernel, exactly the calls above. */
int vlr_gemm_grouped_bits(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int groups,
                          long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed, float p_drop, int mask_ld,
                          const void* mask_bits, long mask_gstride, vlr_stream_t stream);
int vlr_gemm_dropout_acc_bits(const void* v, int ldv, const void* A, void* dx, void* scratch, int M, int in, int r, float p,
                              uint64_t seed, float scale, const void* bits, vlr_stream_t stream);
int vlr_gemm_dropout_acc_multi_bits(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p, uint64_t seed,
                                    float scale, int accumulate, const void* bits, long bits_gstride, vlr_stream_t stream);
/*  ..._rows (ABI v7)     : the adapter restricted to a ROW SET (InternLM-XComposer2's PLoRA acts on the image rows only, reference
 *                          models/InternLMXC2/build_mlp.py:194-202).  rowmask [M] bytes, 1 = the row takes part.
 *                          vlr_gemm_grouped_bits_rows (layouts 0 / 1, accumulate 0): 128-row output tiles without a marked row are not
 *                          computed - the caller zeroes the unmarked rows afterwards (vlr_rows_mask), which it does anyway.
 *                          vlr_gemm_dropout_acc_multi_rows: the caller guarantees zero v rows on the unmarked rows; with accumulate = 1
 *                          every 64-row slab without a marked row is skipped (dx += 0).  rowmask = NULL: the calls above. */
int vlr_gemm_grouped_bits_rows(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int groups,
                               long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed, float p_drop,
                               int mask_ld, const void* mask_bits, long mask_gstride, const unsigned char* rowmask, vlr_stream_t stream);
/*  vlr_rows_tile_list    : out[0] = n, out[1 .. n] = the 64-row tiles of [0, M) with a marked row, ascending (out: M / 64 + 2 ints).
 *  vlr_gemm_grouped_bits_ktiles (layout 2, C = A^T B over K = token rows, K % 64 == 0): contracts only over the listed K tiles - the
 *                          caller guarantees that every other row contributes zero (PLoRA's dB = dy^T u, dA = v^T drop(x)).  The list
 *                          is read on the device; split-K slices cut the LIST, so the summation order differs from the dense call. */
int vlr_rows_tile_list(const unsigned char* rowmask, int M, int* out, vlr_stream_t stream);
/* (ABI v8) vlr_rows_tile_flags: flags[t] = 1 when NO row of the tile_rows-row tile t of [0, M) is marked (ceil(M / tile_rows) bytes).
 * vlr_gemm_seg_rowskip(tile_flags, keep): the NEXT adapter-segment GEMM call of this thread (vlr_gemm_lora[_f32res] / vlr_gemm_swiglu_lora /
 * vlr_gemm_qkv_rope_lora) may run the 256-row tiles whose flag is set over only the first `keep` (multiple of 64; 0 = none) K elements of
 * every sub-target's r-wide block of the segment - the caller guarantees that the REST of u is zero on the rows of those tiles, so the
 * result is bit-identical to the full segment.  InternLM-XComposer2: the [u_lora | u_plora] segment of the two-adapter passes with keep =
 * r_lora (PLoRA acts on the image rows only: reference models/InternLMXC2/build_mlp.py:194-202), and PLoRA alone with keep = 0.
 * tile_flags = NULL clears a pending setting. */
int vlr_rows_tile_flags(const unsigned char* rowmask, int M, int tile_rows, unsigned char* flags, vlr_stream_t stream);
int vlr_gemm_seg_rowskip(const unsigned char* tile_flags, int keep);
int vlr_gemm_grouped_bits_ktiles(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int groups,
                                 long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed, float p_drop,
                                 int mask_ld, const void* mask_bits, long mask_gstride, const int* ktlist, vlr_stream_t stream);
int vlr_gemm_dropout_acc_multi_rows(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p, uint64_t seed,
                                    float scale, int accumulate, const void* bits, long bits_gstride, const unsigned char* rowmask,
                                    vlr_stream_t stream);
/* (ABI v9) The row-slab adapter products, streamed (csrc/lora_rows.hip): a workgroup owns 64 token rows and reads its rows of the
 * activation ONCE at HBM rate straight into MFMA fragments, the small matrix travels through an LDS ring; no split-K partials, no
 * reduction launch, bit-reproducible.  r in {64, 128, 256}; in / outs multiples of 64; rows 16-byte aligned.
 *  vlr_lora_rows_u: u [M][t * ustride .. + r] = alpha * (keep_t . x [M][in]) . A_t [r][in]^T for the n <= 4 targets that share x (A = the
 *                   stacked A_t; peft lora.Linear.forward: lora_A(dropout(x)) * scaling, reference utils/auto_load.py:559-571).  bits = the
 *                   packed keep masks of vlr_dropout_bits (target t at bits + t * bits_gstride; x dense, ldx == in) or NULL (no dropout).
 *  vlr_lora_rows_v: v [M][t * r .. + r] = dy [M][ofs_t .. + outs[t]] . B_t [outs[t]][r], B = the stacked B_t, ofs_t = outs[0] + .. + outs[t-1]
 *                   (outs: HOST array) - the input of the dA / dx terms of the backward.
 *  rowmask [M] bytes or NULL: a row-restricted adapter (InternLM-XComposer2's PLoRA, reference models/InternLMXC2/build_mlp.py:194-202) -
 *                   unmarked rows come out ZERO, 64-row slabs without a marked row are not read. */
int vlr_lora_rows_u(int n, const void* x, int ldx, const void* A, void* u, int ldu, int ustride, int M, int in, int r, float alpha,
                    const void* bits, long bits_gstride, const unsigned char* rowmask, vlr_stream_t stream);
int vlr_lora_rows_v(int n, const void* dy, int lddy, const int* outs, const void* B, void* v, int ldv, int M, int r,
                    const unsigned char* rowmask, vlr_stream_t stream);
/*  vlr_gemm_swiglu_bwd_add: vlr_gemm_swiglu_bwd with an addend on d act before the SwiGLU backward (d act = dy . wdown + dact_add, bf16
 *                          [M][I]; may be the dact_ws buffer) - the LoRA adapter term of down_proj */
int vlr_gemm_swiglu_bwd_add(const void* dy, const void* wdown, void* gu_inout, void* dact_ws, const void* dact_add, int M, int I, int H,
                            vlr_stream_t stream);
int vlr_gemm_lora(const void* x, int ldx, const void* W, void* y, int ldy, const void* residual, int ldr, int M, int N, int K,
                  const void* u, int ldu, const void* Bl, int r, vlr_stream_t stream);
/*  vlr_gemm_lora_f32res  : the same on the fp32 residual stream - y fp32 [M][ldy] = x W^T + u Bl^T + residual fp32 [M][ldr] */
int vlr_gemm_lora_f32res(const void* x, int ldx, const void* W, float* y, int ldy, const float* residual, int ldr, int M, int N, int K,
                         const void* u, int ldu, const void* Bl, int r, vlr_stream_t stream);
int vlr_gemm_swiglu_lora(const void* x, const void* wgu, void* gu, void* act, int M, int I, int K, int ldx, const void* u, int ldu,
                         const void* Bl, int r, vlr_stream_t stream);
int vlr_gemm_qkv_rope_lora(const void* x, const void* wqkv, const void* bias, void* qkv, const int* pos, const float* cos_t,
                           const float* sin_t, int M, int N, int rope_cols, int K, int ldx, int head_dim, int max_pos, const void* u,
                           int ldu, const void* Bl, int r, int q_cols, int kv_cols, vlr_stream_t stream);

/* ---- normalisation / activations (transformers LlamaRMSNorm, CLIP LayerNorm, SwiGLU, GELU; call sites
 *      Llava/__init__.py:178-191,232) ------------------------------------------------------------------------- */
int vlr_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int H, float eps, vlr_stream_t stream);
/* x fp32 [M][H] (fp32 residual stream); y bf16 */
int vlr_rmsnorm_fwd_f32(const float* x, const void* w, void* y, float* rstd, int M, int H, float eps, vlr_stream_t stream);
int vlr_rmsnorm_bwd_workspace_bytes(int H);
int vlr_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                    void* dw, int dw_accumulate, void* workspace, int M, int H, vlr_stream_t stream);
/* the same with the forward input x in fp32; dy / dres / dx (the gradient stream) are bf16 */
int vlr_rmsnorm_bwd_f32(const void* dy, const float* x, const void* w, const float* rstd, const void* dres, void* dx,
                        void* dw, int dw_accumulate, void* workspace, int M, int H, vlr_stream_t stream);
int vlr_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int M, int D, float eps, vlr_stream_t stream);
/* backward of y = LayerNorm(x) * w + b (Qwen-VL resampler: ln_q / ln_kv / ln_post, QwenVL/visual.py:122-123,391): mean and rstd are
 * recomputed from x; dx may be NULL, dw / db (bf16 [D], `accumulate`: +=) may be NULL; workspace = vlr_layernorm_bwd_workspace_bytes(D) */
int vlr_layernorm_bwd_workspace_bytes(int D);
int vlr_layernorm_bwd(const void* dy, const void* x, const void* w, float eps, void* dx, void* dw, void* db, int accumulate,
                      void* workspace, int M, int D, vlr_stream_t stream);
int vlr_vit_embed_ln(const void* patch_embeds, const void* cls, const void* pos, const void* w, const void* b, void* y,
                     int n_img, int T, int D, float eps, vlr_stream_t stream);
int vlr_im2col(const float* pixel_values, void* patches, int n_img, int image_size, int patch, int Kp, vlr_stream_t stream);
int vlr_rope_table(float* cos_t, float* sin_t, int max_pos, int head_dim, float theta, vlr_stream_t stream);
int vlr_rope(void* qkv, const int* pos, const float* cos_t, const float* sin_t, int M, int H, int head_dim, int ld,
             int max_pos, int backward, vlr_stream_t stream);
/* same on the first n_heads heads of every row (q heads followed by the k heads; grouped-query layouts have fewer k heads) */
int vlr_rope_heads(void* qkv, const int* pos, const float* cos_t, const float* sin_t, int M, int n_heads, int head_dim, int ld,
                   int max_pos, int backward, vlr_stream_t stream);
int vlr_swiglu_fwd(const void* gate_up, void* act, int M, int I, vlr_stream_t stream);
int vlr_swiglu_bwd(void* gate_up_inout, const void* dact, int M, int I, vlr_stream_t stream);
int vlr_gelu_fwd(const void* z, void* h, long n, vlr_stream_t stream);
int vlr_gelu_bwd(const void* z, const void* dh, void* dz, long n, vlr_stream_t stream);
int vlr_colsum_workspace_bytes(int C);
int vlr_colsum(const void* X, int R, int C, int ld, void* out, int accumulate, void* workspace, vlr_stream_t stream);
int vlr_colsum_f32(const void* X, int R, int C, int ld, float* out, void* workspace, vlr_stream_t stream);
int vlr_gather_rows(const void* src, const int* rows, void* dst, int R, int H, vlr_stream_t stream);
int vlr_scatter_rows(const void* src, const int* rows, void* dst, int R, int H, vlr_stream_t stream);
/* strided row-list operations on a column block of a wider matrix (PLoRA of InternLM-XComposer2, reference
 * models/InternLMXC2/build_mlp.py:194-202: `res[im_mask] += Plora_B(Plora_A(x[im_mask]))`):
 *   vlr_rows_gather: dst [R][W] (contiguous) = src[rows[r]][0:W], src row stride lds
 *   vlr_rows_add   : dst[rows[r]][0:W] += src [R][W], dst row stride ldd; `rows` must not repeat */
int vlr_rows_gather(const void* src, int lds, const int* rows, void* dst, int R, int W, vlr_stream_t stream);
int vlr_rows_add(const void* src, const int* rows, void* dst, int ldd, int R, int W, vlr_stream_t stream);
int vlr_cast_f32_to_bf16(const float* src, void* dst, long n, vlr_stream_t stream);
int vlr_cast_bf16_to_f32(const void* src, float* dst, long n, vlr_stream_t stream);
int vlr_rowdot(const void* X, const float* v, float* out, int M, int H, vlr_stream_t stream);

/* ---- attention (transformers LlamaAttention / CLIPAttention, eager semantics: fp32 softmax, causal + key padding
 *      for the decoder, full for the ViT; replaces flash-attn 2.5.8 too, utils/auto_load.py:49-56,534) -------------
 * q/k/v: column blocks of a fused [batch*S][ld] buffer (head h = columns h*head_dim..).  lse: [batch][heads][Sp]
 * floats, Sp = S rounded up to 64, log2 domain.  key_mask: [batch][S] int32 (0 = padded key) or NULL. */
int vlr_attn_fwd(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse,
                 const int* key_mask, int batch, int S, int heads, int head_dim, int causal, float scale,
                 vlr_stream_t stream);
int vlr_attn_bwd(const void* q, const void* k, const void* v, int ld, const void* o, const void* dout, int ldo,
                 const float* lse, float* delta_ws, const int* key_mask, void* dq, void* dk, void* dv, int ldd,
                 int batch, int S, int heads, int head_dim, int causal, float scale, vlr_stream_t stream);

/* grouped-query attention (Mistral / InternLM2: kv_heads < heads; k, v column blocks hold kv_heads heads, query head h reads
 * K/V head h / (heads / kv_heads); dk, dv are summed over the group's query heads in a fixed order - no atomics).
 * vlr_attn_fwd / vlr_attn_bwd are the kv_heads == heads case. */
int vlr_attn_fwd_gqa(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse,
                     const int* key_mask, int batch, int S, int heads, int kv_heads, int head_dim, int causal, float scale,
                     vlr_stream_t stream);
int vlr_attn_bwd_gqa(const void* q, const void* k, const void* v, int ld, const void* o, const void* dout, int ldo,
                     const float* lse, float* delta_ws, const int* key_mask, void* dq, void* dk, void* dv, int ldd,
                     int batch, int S, int heads, int kv_heads, int head_dim, int causal, float scale, vlr_stream_t stream);

/* ---- image/text merge (LlavaForRL._merge_input_ids_with_image_features, Llava/__init__.py:36-109) ---------------
 * info[0] receives the number of image slots found; the caller compares it with n_feat_rows*dup and raises the
 * reference's ValueError (:90-94) on mismatch.  `dup` = how many batch halves share one feature table (the
 * reference duplicates every image, base/trainer.py:138-142; the frozen ViT runs once per distinct image here). */
int vlr_merge_index(const long* input_ids, const long* attention_mask, const long* labels, int Bn, int T, int S, int P,
                    int image_token, int pad_token, int n_feat_rows, int dup, int* src, int* out_mask,
                    long* out_labels, int* out_pos, unsigned char* img_map, int* inv_map, int* info, vlr_stream_t stream);
int vlr_merge_fwd(const int* src, const long* input_ids, const void* embed_table, const void* feats, void* out, int Bn,
                  int T, int S, int H, vlr_stream_t stream);
/* the same into an fp32 residual stream (vlr_llama_cfg.resid_f32): out fp32 [Bn][S][H]; embed_table bf16 (rows widened exactly);
 * feats fp32 [rows][H] when feats_f32 (the projector's unrounded output), else bf16 */
int vlr_merge_fwd_f32(const int* src, const long* input_ids, const void* embed_table, const void* feats, int feats_f32, float* out,
                      int Bn, int T, int S, int H, vlr_stream_t stream);
int vlr_merge_bwd(const void* dmerged, const int* src, const int* inv_map, const long* input_ids, void* dfeats,
                  void* dembed_table, int Bn, int T, int S, int H, int n_feat_rows, int dup, vlr_stream_t stream);

/* ---- log-probabilities (VLDPOTrainer.get_batch_logps, base/trainer.py:148-188) ------------------------------- */
int vlr_build_rows(const long* labels, const unsigned char* shared_mask, int Bn, int S, int label_pad, int* rows,
                   int* tgt, int* seq_off, vlr_stream_t stream);
int vlr_logp_rows(const float* logits, const int* row_idx, const int* tgt, int R, int V, long ld, float* tok_logp,
                  float* lse, vlr_stream_t stream);
int vlr_dlogits_rows(const float* logits, const int* tgt, const float* lse, const int* seq_off, int nseq,
                     const float* dlogps, int average, int R, int V, long ld, void* dlogits, long ldd,
                     vlr_stream_t stream);
int vlr_seq_sum(const float* tok_logp, const int* seq_off, int nseq, int average, float* out, vlr_stream_t stream);
/* Fused lm-head + log-softmax + label pick on the R response rows (SURVEY 8b `vlr_lmhead_logps_fwd/bwd`): hg [R][H] bf16 are the
 * gathered final hidden states, w_lm [V][H]; tok_logp[r] = logit[r][tgt[r]] - lse[r].  When vlr_lmhead_is_fused(R, V, H) the [R][V]
 * logits never reach HBM: the forward GEMM's epilogue leaves per-wave (max, sum exp) partials in `workspace`
 * (vlr_lmhead_workspace_bytes) that a second kernel folds in a fixed order, the backward recomputes the GEMM and writes
 * d logits [R][V] bf16 = g_r * ([v == tgt_r] - softmax) straight from its epilogue (g_r = dlogps of the row's sequence, / row count
 * when `average`).  Otherwise `logits_ws` ([R][V] fp32) carries the logits through vlr_logp_rows / vlr_dlogits_rows. */
long vlr_lmhead_workspace_bytes(int R, int V);
int vlr_lmhead_is_fused(int R, int V, int H);
int vlr_lmhead_logps_fwd(const void* hg, const void* w_lm, const int* tgt, float* tok_logp, float* lse, void* workspace,
                         float* logits_ws, int R, int V, int H, vlr_stream_t stream);
int vlr_lmhead_logps_bwd(const void* hg, const void* w_lm, const int* tgt, const float* lse, const int* seq_off, int nseq,
                         const float* dlogps, int average, void* dlogits, void* workspace, float* logits_ws, int R, int V, int H,
                         vlr_stream_t stream);

/* ---- DPO loss, forward + backward (VLDPOTrainer.dpo_loss, base/trainer.py:244-301).
 * loss_type 0 sigmoid|ddpo, 1 hinge, 2 ipo, 3 kto_pair (losses has 2n entries).  dpc/dpr = d(sum_i g_i*loss_i)/d
 * policy_{chosen,rejected}_logps with g = grad_losses or 1/len(losses) when NULL (trl: loss = losses.mean()). */
int vlr_dpo_loss(const float* pc, const float* pr, const float* rc, const float* rr, int n, float beta,
                 float label_smoothing, int loss_type, int reference_free, float* losses, float* chosen_rewards,
                 float* rejected_rewards, float* dpc, float* dpr, float* loss_mean, const float* grad_losses,
                 vlr_stream_t stream);

/* ---- optimizer on flat buffers (torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW as driven by the HF Trainer;
 *      hyper-parameters scripts/dpo_llava.sh:35-41).  out3 = {grad norm, clip coefficient * gscale, raw sum g^2};
 *      vlr_adamw_step reads coef[1] on the device (no host sync). */
int vlr_grad_sqnorm_workspace_bytes(void);
int vlr_grad_sqnorm(const void* grads, long n, float max_norm, float gscale, float extra_sq, void* workspace,
                    float* out3, vlr_stream_t stream);
int vlr_adamw_step(float* master, float* m, float* v, const void* grads, void* params_bf16, long n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, const float* coef,
                   vlr_stream_t stream);

/* ---- composed passes (transformers LlamaDecoderLayer / CLIPEncoderLayer forward + autograd backward) ---------- */
typedef struct {
    int hidden, inter, heads, head_dim;
    float rms_eps;
    int max_pos;            /* rows of the rope tables */
    const float* rope_cos;  /* [max_pos][head_dim/2] */
    const float* rope_sin;
    int kv_heads;           /* grouped-query attention (Mistral, InternLM2): K/V heads, a divisor of heads; 0 = heads */
    int resid_f32;          /* 1: the residual stream (x_in, acts.x_mid, acts.x_out) is fp32 [M][hidden] - never rounded to bf16; the
                             * norms read it in fp32, o_proj / down_proj add their fp32 accumulators to it (vlr_gemm_bf16_f32res).
                             * HF runs the bf16 checkpoint's residual adds in bf16 (transformers LlamaDecoderLayer.forward); this mode
                             * is what brings the 32-layer loss within north_star's tolerance of the fp32 reference.  The gradient
                             * stream (dx_out / dx_in) stays bf16.  0: bf16 stream (ABI v3 behaviour) */
} vlr_llama_cfg;
/* Shapes with Nq = heads*head_dim, Nkv = kv_heads*head_dim: wqkv [Nq + 2 Nkv][hidden] (q | k | v rows), wo [hidden][Nq],
 * qkv / dqkv activations [M][Nq + 2 Nkv], attn / dattn [M][Nq].  hidden == Nq for LLaMA / Mistral. */
typedef struct {  /* bf16 weights of one decoder layer; q|k|v and gate|up are stored fused */
    const void* ln1; const void* wqkv; const void* wo; const void* ln2; const void* wgu; const void* wdown;
    const void* bqkv;       /* [Nq + 2 Nkv] bias of the fused q|k|v projection (Qwen c_attn, QwenVL/modeling_qwen.py:104); NULL = none.
                             * Its gradient is the column sum of the dqkv scratch after vlr_decoder_layer_bwd (vlr_colsum). */
} vlr_layer_weights;
typedef struct {  /* bf16 gradient buffers, same shapes */
    void* ln1; void* wqkv; void* wo; void* ln2; void* wgu; void* wdown;
} vlr_layer_grads;
typedef struct {  /* activations one layer keeps for its backward (caller-allocated) */
    void* xn1; float* rstd1; void* qkv; void* attn; float* lse; void* x_mid; void* xn2; float* rstd2; void* gu;
    void* act; void* x_out;
} vlr_layer_acts;
typedef struct {  /* scratch shared by all layers in the backward */
    void* dact; void* dxn; void* dattn; void* dqkv; void* dx_mid; float* delta; void* norm_ws;
} vlr_layer_bwd_ws;

int vlr_decoder_layer_fwd(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_layer_acts* a,
                          const void* x_in, const int* pos, const int* key_mask, int batch, int S,
                          vlr_stream_t stream);
/* keep_for_backward = 0: a no-grad pass (reference model, evaluation) - tensors only the backward reads are not written */
int vlr_decoder_layer_fwd_ex(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_layer_acts* a,
                             const void* x_in, const int* pos, const int* key_mask, int batch, int S,
                             int keep_for_backward, vlr_stream_t stream);
int vlr_decoder_layer_bwd(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_layer_grads* g,
                          int accumulate, const vlr_layer_acts* a, const vlr_layer_bwd_ws* ws, const void* x_in,
                          const void* dx_out, void* dx_in, const int* pos, const int* key_mask, int batch, int S,
                          vlr_stream_t stream);

/* ---- LoRA (peft) adapters on the seven decoder linears (reference: LoraConfig built in utils/auto_load.py:559-571 from
 * LlavaForRL.default_lora_target, Llava/__init__.py:273-286; r/alpha/dropout defaults dpo.py:60-63, scripts/ddpo_llava.sh:24-31).
 * peft semantics: y = base(x) + scale * lora_B(lora_A(dropout(x))), scale = lora_alpha / r, base weights frozen.  The path is
 * un-merged: lora_dropout works and the frozen base weights serve as the reference model (adapter disabled, trl
 * null_ref_context).  Sub-targets of a fused group are stacked: A_*: [n*r][in], B_*: [n*out][r].
 * u [M][7r] (columns qkv | o | gate,up | down) = dropout_t(x) A_t^T, written by the forward and read by the backward.
 * ws_v: scratch [M][3r]; ws_xd: the forward ignores it (ABI v3 stored the seven dropped inputs there: 0.9 GB per layer at the 7B
 * shapes); the backward with dropout > 0 needs ONE scratch [M][max(hidden, inter)] shared by all layers.  Dropout target t of the layer
 * draws its mask from vlr_dropout(seed + t), t = 0..6 in q,k,v,o,gate,up,down order; the mask is applied to x while the skinny GEMMs
 * stage it (vlr_gemm_grouped) and regenerated in the backward - drop(x) is never written. */
typedef struct {
    int r;
    float scale;     /* lora_alpha / r */
    float dropout;   /* 0 in eval mode */
    const void* a_qkv; const void* b_qkv;   /* [3r][H], [3H][r] */
    const void* a_o;   const void* b_o;     /* [r][H],  [H][r]  */
    const void* a_gu;  const void* b_gu;    /* [2r][H], [2I][r] */
    const void* a_down; const void* b_down; /* [r][I],  [H][r]; both NULL: no adapter on the down projection */
    int qkv_targets;  /* 0 or 3: q_proj, k_proj, v_proj adapted separately (a_qkv [3r][H], u_q | u_k | u_v); 1: ONE adapter on the
                       * fused projection (Qwen c_attn: a_qkv [r][H], b_qkv [N][r]) */
    void* mask_bits;  /* ABI v5, dropout > 0 only: vlr_lora_mask_bytes(hidden, inter, M) bytes of THIS layer, or NULL.  The forward draws
                       * the seven packed keep masks into it in both forms of vlr_dropout_bits2 (row-major: target t < 6 at t * M * hidden
                       * / 8, down at 6 * M * hidden / 8; behind them, 64-byte aligned, the K-tile-blocked transposed ones) and the adapter
                       * GEMMs of the forward AND of the backward read them; NULL: every kernel hashes */
} vlr_lora_weights;
long vlr_lora_mask_bytes(int hidden, int inter, int M);
typedef struct {
    void* a_qkv; void* b_qkv; void* a_o; void* b_o; void* a_gu; void* b_gu; void* a_down; void* b_down;
} vlr_lora_grads;
int vlr_decoder_layer_fwd_lora(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                               const vlr_layer_acts* a, void* u, void* ws_xd, uint64_t seed, const void* x_in,
                               const int* pos, const int* key_mask, int batch, int S, vlr_stream_t stream);
int vlr_decoder_layer_bwd_lora(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                               const vlr_lora_grads* lg, int accumulate, const vlr_layer_acts* a, const void* u,
                               const vlr_layer_bwd_ws* ws, void* ws_v, void* ws_xd, uint64_t seed, const void* x_in,
                               const void* dx_out, void* dx_in, const int* pos, const int* key_mask, int batch, int S,
                               vlr_stream_t stream);
/* The same two passes for a decoder whose linears carry adapters as BASE-MODEL weights on a row subset - PLoRA of InternLM-XComposer2
 * (reference models/InternLMXC2/build_mlp.py:158-203: `res[im_mask] += Plora_B(Plora_A(dropout(x[im_mask])))`, r = 256): rowmask
 * [batch * S] bytes, non-zero = image row (NULL: all rows) - the text rows of u / v are zeroed, so the dense adapter kernels compute
 * exactly the row-restricted update; g != NULL: the base weights train too (full fine-tune: gradients as vlr_decoder_layer_bwd
 * writes them), NULL: frozen.  The dropout mask of target t is vlr_dropout(seed + t) indexed over the FULL [batch * S][in] input. */
int vlr_decoder_layer_fwd_lora_ex(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                                  const vlr_layer_acts* a, void* u, void* ws_xd, uint64_t seed, const unsigned char* rowmask,
                                  const void* x_in, const int* pos, const int* key_mask, int batch, int S, vlr_stream_t stream);
int vlr_decoder_layer_bwd_lora_ex(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_layer_grads* g,
                                  const vlr_lora_weights* lw, const vlr_lora_grads* lg, int accumulate, const vlr_layer_acts* a,
                                  const void* u, const vlr_layer_bwd_ws* ws, void* ws_v, void* ws_xd, uint64_t seed,
                                  const unsigned char* rowmask, const void* x_in, const void* dx_out, void* dx_in, const int* pos,
                                  const int* key_mask, int batch, int S, vlr_stream_t stream);
/* ---- ABI v7: TWO adapters per linear - peft LoRA (trainable, every row) stacked on PLoRA (frozen base-model weights, image rows only):
 * what reference scripts/dpo_internlmxc2vl7b.sh trains (--use_lora True over the PLoRA decoder of models/InternLMXC2/build_mlp.py:158-203,
 * LoraConfig from utils/auto_load.py:559-571).  y = W x + s_l B_l A_l drop_l(x) + [image rows] s_p B_p A_p drop_p(x) is ONE adapter segment
 * of rank R = lw->r + pw->r in the K loop of the fused projections: u [M][7R], per sub-target [u_lora (r_l) | u_plora (r_p)], and `bc` holds
 * the groups' row-wise concatenations [B_lora | B_plora] ([n*out][R], vlr_lora_concat_b; rebuilt by the caller when B_lora has moved).  The
 * two adapters draw independent dropout masks (seed_l + t / seed_p + t, their own mask_bits), both indexed over the full [batch*S][in]
 * operand; rowmask applies to the PLoRA half.  Backward: gradients for the LoRA pairs only (lg), input gradients through both; ws_v
 * [M][2R] at least (3R when q, k, v are adapted separately). */
typedef struct { const void* qkv; const void* o; const void* gu; const void* down; } vlr_lora_bcomb;
int vlr_lora_concat_b(const void* b1, int r1, const void* b2, int r2, void* out, long rows, vlr_stream_t stream);
int vlr_decoder_layer_fwd_lora2(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                                const vlr_lora_weights* pw, const vlr_lora_bcomb* bc, const vlr_layer_acts* a, void* u,
                                uint64_t seed_l, uint64_t seed_p, const unsigned char* rowmask, const void* x_in, const int* pos,
                                const int* key_mask, int batch, int S, vlr_stream_t stream);
int vlr_decoder_layer_bwd_lora2(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                                const vlr_lora_grads* lg, const vlr_lora_weights* pw, const vlr_lora_bcomb* bc, int accumulate,
                                const vlr_layer_acts* a, const void* u, const vlr_layer_bwd_ws* ws, void* ws_v, uint64_t seed_l,
                                uint64_t seed_p, const unsigned char* rowmask, const void* x_in, const void* dx_out, void* dx_in,
                                const int* pos, const int* key_mask, int batch, int S, vlr_stream_t stream);
/* rows of x [M][ld] (first `cols` columns) whose rowmask byte is 0 are zeroed */
int vlr_rows_mask(void* x, int ld, int cols, const unsigned char* rowmask, int M, vlr_stream_t stream);
/* counter-based dropout: out = mask * x * alpha / (1-p)  (add != 0: out += ...); the mask is a pure function of
 * (seed, element index) so the backward regenerates it.  vlr_dropout_mask writes the keep mask as bytes (tests). */
int vlr_dropout(const void* x, void* out, long n, float p, uint64_t seed, float alpha, int add, vlr_stream_t stream);
/* the keep mask of vlr_dropout(seed) over n elements (n % 32 == 0), packed: bit e of byte i = element 8 i + e */
int vlr_dropout_bits(void* bits_u8, long n, float p, uint64_t seed, vlr_stream_t stream);
/* the same mask over a [rows][cols] operand in BOTH packed forms (either pointer may be NULL): row-major as above, and K-tile-blocked
 * transposed - byte ((row / 64) * cols + col) * 8 + (row % 64) / 8, bit e = row + e; vlr_dropout_bits_kt_bytes(rows, cols) bytes, 8-byte
 * aligned - which vlr_gemm_grouped_bits(layout 2, mask_on 3) reads for dA = v^T (mask . x) */
int vlr_dropout_bits2(void* bits_u8, void* bits_kt_u8, int rows, int cols, float p, uint64_t seed, vlr_stream_t stream);
long vlr_dropout_bits_kt_bytes(int rows, int cols);
int vlr_dropout_mask(void* mask_u8, long n, float p, uint64_t seed, vlr_stream_t stream);

/* vlr_decoder_layer_bwd runs the weight-gradient GEMMs on a library-owned side stream (VLR_BWD_STREAMS=0 disables);
 * vlr_layers_join makes `stream` wait for them - call it before anything reads or reduces the weight gradients. */
int vlr_layers_join(vlr_stream_t stream);

typedef struct {
    int hidden, mlp, heads, head_dim;
    float ln_eps;
    /* the following default (0) to the CLIP tower of LLaVA; the Qwen-VL tower (QwenVL/visual.py:244-300) sets them */
    int act;            /* MLP activation: 0 / 1 quick_gelu (CLIP), 2 exact GELU (nn.GELU) */
    int head_dim_pad;   /* > head_dim: q|k|v rows and the attention output are laid out with head_dim_pad (128) features per head
                         * (weight rows / columns of the padding are zero) so that head_dim 104 runs on the 128-wide attention kernel;
                         * ws->qkv is [M][3*heads*head_dim_pad], ws->attn [M][heads*head_dim_pad] */
    float attn_scale;   /* softmax scale; 0 = 1/sqrt(head_dim) (of the TRUE head_dim when 0 and head_dim_pad is set) */
} vlr_vit_cfg;
typedef struct {  /* bf16; q|k|v fused [3D][D] + bias [3D] */
    const void* ln1_w; const void* ln1_b; const void* wqkv; const void* bqkv; const void* wo; const void* bo;
    const void* ln2_w; const void* ln2_b; const void* w1; const void* b1; const void* w2; const void* b2;
} vlr_vit_layer_weights;
typedef struct { void* xn; void* qkv; void* attn; void* h; } vlr_vit_ws;   /* [M][D], [M][3D], [M][D], [M][mlp] */
int vlr_vit_layer_fwd(const vlr_vit_cfg* cfg, const vlr_vit_layer_weights* w, const vlr_vit_ws* ws, void* x_inout,
                      int n_img, int T, vlr_stream_t stream);

/* ---- data-parallel gradient exchange on RCCL over xGMI (replaces accelerate MULTI_GPU / torch DDP's NCCL all-reduce:
 *      /root/reference accelerate_config/ddp.yaml:1-14; the reference itself never calls a collective).  One process per
 *      GPU, one communicator per process.  RCCL is dlopen'ed at run time (the copy already mapped into the process,
 *      else librccl.so.1; override with VLR_RCCL_LIB).  vlr_comm_unique_id fills `vlr_comm_unique_id_bytes()` HOST
 *      bytes on rank 0; the launcher carries them to every rank; vlr_comm_init blocks until all `world` ranks called
 *      it.  vlr_allreduce_bucket: in-place SUM over one contiguous slice of the flat gradient buffer, enqueued on
 *      `stream` (dtype 0 = bf16, 1 = fp32); 1/world is folded into the optimizer's gradient scale. */
/* vlr_set_comm_cus(k): the persistent GEMM / attention launches leave k CUs (rounded up to whole XCD octets) to the RCCL kernels
 * that run beside the backward (default: environment VLR_COMM_CUS, else 0); vlr_compute_cus() = what is left for them.
 * vlr_comm_probe: diagnostics only - a streaming copy src -> dst of n bytes on exactly `wgs` workgroups, the stand-in for one RCCL
 * ring kernel in the single-GPU interference bench (tools/comm_interference.py); no communicator involved. */
int vlr_set_comm_cus(int k);
int vlr_compute_cus(void);
int vlr_comm_probe(const void* src, void* dst, long n_bytes, int wgs, vlr_stream_t stream);
int vlr_comm_unique_id_bytes(void);
const char* vlr_comm_library(void);   /* path of the RCCL library in use ("" + vlr_last_error() when none loads) */
int vlr_comm_unique_id(void* id_host);
int vlr_comm_init(const void* id_host, int rank, int world, void** comm_out);
/* (ABI v8) ... with a PER-COMMUNICATOR bound on RCCL's channels (one channel = one workgroup of the ring kernel): ncclCommInitRankConfig
 * with minCTAs / maxCTAs, so that the ring kernels fit the CUs vlr_set_comm_cus reserved without the process-wide NCCL_*_NCHANNELS
 * environment (two communicators of one process may differ: bench.py measures one bucket bounded and unbounded).  max_ctas <= 0 = no
 * bound (vlr_comm_init).  VLR_ERR_HIP when the loaded RCCL has no ncclCommInitRankConfig or rejects the configuration - the caller
 * (vlrlhf/parallel.py NativeComm) then agrees with the other ranks and falls back to vlr_comm_init under the environment bound.
 * vlr_comm_rccl_version: ncclGetVersion of the library in use (major * 10000 + minor * 100 + patch; 0: unknown). */
int vlr_comm_init_cfg(const void* id_host, int rank, int world, int min_ctas, int max_ctas, void** comm_out);
int vlr_comm_has_config(void);      /* (ABI v9) 1: the loaded RCCL exports ncclCommInitRankConfig - agreed on by all ranks before vlr_comm_init_cfg is called */
int vlr_comm_rccl_version(void);
int vlr_comm_destroy(void* comm);
int vlr_allreduce_bucket(void* comm, void* buf, long n, int dtype, vlr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
