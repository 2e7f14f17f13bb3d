// Shared definitions of the bf16 GEMM kernels (gemm.hip: 128x128 tile, gemm256p.hip: 256x256 tile).
#pragma once
#include "common.h"

enum { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU = 2 };

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const bf16_t* bias;      // [N] or null
    const bf16_t* residual;  // [M,N] ld = ldr, or null (res_f32: an fp32 tensor behind the same pointer, ldr in floats)
    int M, N, K;
    int lda, ldb, ldc, ldr;
    int act;
    int accumulate;  // C += result (C read in its own dtype)
    int out_f32;
    int flags;       // tuning experiments (VLR_GEMM_FLAGS), 0 in production
    float alpha;     // v = act(alpha * acc + bias) + residual (+ C)
    int splitk;      // > 1: blockIdx.y = K-slice z of kchunk elements, raw alpha*acc -> part[z][M][N] fp32 (128x128 kernel only)
    int kchunk;
    float* part;
    // ---- fused epilogues of the 256x256 continuous-pipeline kernel (gemm256p.hip, NT only); fuse = 0 for a plain GEMM
    //  1 SwiGLU : B = [gate | up] weights [2I][K], N = 2I.  A workgroup computes gate columns [n0, n0+128) and the MATCHING up
    //             columns (its B-hi half tile is fetched from row I + n0), writes act = silu(gate) * up to C2 [M][ldc2] and,
    //             when store_c, gate | up to C [M][ldc] (the backward needs them; the no-grad reference pass does not).
    //  2 RoPE   : B = [q | k | v] weights, head_dim 128.  The B-lo / B-hi half tiles hold the first / second 64 features of the
    //             tile's two heads, so a lane owns both members of every rotate-half pair; columns < rope_cols are rotated by
    //             pos[row] in fp32 before the single rounding to bf16, the rest (v) pass through.
    //  3 SwiGLU backward (NN): acc = d act; C2 = gate | up [M][2I] is replaced in place by d gate | d up; C is not written.
    int fuse;
    int store_c;
    void* C2;
    int ldc2;
    const int* pos;
    const float* rope_cos;   // [max_pos][64]
    const float* rope_sin;
    int max_pos;
    int rope_cols;
    //  4 lm-head log-probs, forward (NT, A = response-row hidden states [R][H], B = lm_head [V][H]): no logits leave the kernel -
    //    every wave reduces its 64 columns of a row to (max, sum exp) -> part[row][tile_n * 4 + wc][2] and the lane that owns the
    //    row's target column stores that logit to f1[row]; `pos` = target ids.  A second kernel folds the partials (gemm.hip).
    //  5 lm-head log-probs, backward: C (bf16 [R][V]) = f1[row] * ([col == target] - exp(logit - f0[row])), f0 = lse, f1 = dlogps
    //    of the row's sequence (/ count when averaging)
    float* f0;
    float* f1;
    // ---- adapter segment (LoRA, NT only; template SEG of gemm256p_kernel): the K loop runs on past K into a second pair of
    // operands, C = A B^T + A2 B2^T with B2 block-diagonal over the output blocks [0,seg_b0) [seg_b0,seg_b1) [seg_b1,N):
    //   A2 [M][lda2] = s * dropout_t(x) A_t^T of the group's targets side by side (K2 columns each), B2 [N][ldb2] = lora_B rows.
    // An output tile of block t multiplies only A2[:, t*K2 .. (t+1)*K2) (the zero blocks are never touched); with fuse = 1 the
    // gate half uses A2[:, 0..K2) and the up half A2[:, K2..2*K2) (the other half's K range is fed from the zero page).
    // K % 64 == 0 required; K2 == 0: no segment.
    const bf16_t* A2;
    const bf16_t* B2;
    int lda2, ldb2, K2;
    int seg_b0, seg_b1;
    // ---- fuse = 6 (NN, LDS-image epilogue): C += keep(row * drop_ld + col) ? alpha * acc : 0 - the LoRA input-gradient term
    // dx += scaling/(1-p) * mask_t .* (v_t A_t) without materialising v_t A_t (counter-based mask of common.h)
    uint64_t drop_key;
    uint32_t drop_thr;
    int drop_ld;
    // ---- fp32 residual stream (vlr_llama_cfg::resid_f32): residual is read as fp32 [M][ldr]; with out_f32 the o_proj / down_proj
    // launches are C fp32 = acc + residual fp32 (no rounding of the stream at all)
    int res_f32;
    // ---- second problem of a GROUPED launch of the 256x256 TN continuous-pipeline kernel (gemm256p.hip, template GRP; same K, alpha 1, plain
    // bf16 epilogue): C1 [M1][ldc1] = A1^T B1.  Set by vlr_gemm256p_tn_pair_try_launch only.
    const bf16_t* A1;
    const bf16_t* B1;
    void* C1;
    int M1, N1, lda1, ldb1, ldc1;
    // ---- adapter segment, row-tile skip (round 5; two-adapter launches [u_keep | u_rest] per sub-target, u_rest ZERO on the rows of a tile
    // whose flag is set - InternLM-XComposer2's PLoRA block on all-text row tiles): seg_skip [ceil(M / 256)] bytes (device; 1 = the tile
    // runs only the first seg_keep K elements of every sub-target's K2-wide block) or null.  seg_keep % 64 == 0 and K2 % 64 == 0.
    const unsigned char* seg_skip;
    int seg_keep;
    // ---- A/B switches of the continuous-pipeline kernels (vlr_gemm_set_sched): bit 3 = adapter K tiles on the general staging path,
    // bit 4 = the two wave groups run their epilogues one after the other (the order before round 4), bit 5 (32) = the shared-panel tile
    // map of gemm_tilemap.h (round 5; off = the per-XCD contiguous tile ranges of rounds 1-4).  32 in production.
    int sched;
    // ---- 128x128 kernel only (gemm.hip): GROUPED launches and an on-the-fly lora_dropout mask.
    // groups > 1: blockIdx.z = group g runs the same problem shape on A + g * gA, B + g * gB, C + g * gC (elements of each type) -
    //   the skinny per-target GEMMs of a LoRA group (q, k, v / gate, up) in ONE launch that fills the chip instead of three at 39 %.
    // mask_on: 1 = zero the dropped elements of the k-contiguous A operand (NT: x rows [M][mask_ld]) while it is staged, 2 = of the
    //   k-strided B operand (TN: x stored [K = rows][mask_ld]); keep(row * mask_ld + col) of vlr_dropout(mask_seed + g); the 1 / (1 - p)
    //   factor travels in alpha.  x is never copied and drop(x) never stored: the backward regenerates the same mask.
    int groups;
    long gA, gB, gC;
    int mask_on;
    uint64_t mask_seed;
    uint32_t mask_thr;
    int mask_ld;
    // packed keep masks drawn beforehand (vlr_dropout_bits: bit e of byte i = element 8 i + e of the dense [rows][mask_ld] / [M][drop_ld]
    // operand); non-null: the kernels read them instead of hashing.  mask_bits + g * gMask bytes is the mask of group / term g
    // (mask_on, dropacc_multi_kernel); drop_bits the one of the fuse = 6 epilogue.
    const unsigned char* mask_bits;
    long gMask;
    const unsigned char* drop_bits;
    // [M] bytes or null (128x128 ring kernel, row-major A only): an output tile whose 128 rows all carry 0 is not computed - the caller
    // zeroes those rows afterwards (vlr_rows_mask: the text rows of InternLM-XComposer2's PLoRA, 46 % of the rows at 490 x 490 / 1024)
    const unsigned char* rowskip;
    // TN launches of the 128x128 ring kernel (K = token rows, K % 64 == 0): ktlist[0] = n, ktlist[1 .. n] = the K tiles (of 64 rows) to
    // contract over, ascending - the tiles that hold an image row (vlr_rows_tile_list); every other row of both operands' product is zero
    // by construction (PLoRA's u / v on the text rows).  The count is read on the device: split-K slices cut the LIST into equal runs.
    const int* ktlist;
#ifdef VLR_GEMM_TRACE
    uint32_t* trace;         // diagnostics build only: set by the launchers of gemm256p.hip (vlr_gemm_set_trace), never by callers
    int dephase_p, dephase_ticks, epi_abl, trace_clk;
#endif
};
#define VLR_SCHED_DEFAULT 32          // GemmParams::sched when VLR_GEMM_SCHED is not set: the shared-panel tile map (gemm_tilemap.h, round 5)
int vlr_gemm_sched_mode();

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_QUICK_GELU) return v / (1.f + __expf(-1.702f * v));
    if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}


// 256x256 tile, eight-phase schedule (gemm256p.hip); returns false when the problem does not qualify (fewer than 192 tiles,
// unaligned operands, VLR_GEMM_8PHASE=0): the caller falls back to the 128x128 kernel
bool vlr_gemm256p_try_launch(int layout, const GemmParams& p, hipStream_t stream);
// 128x128 tile on the LDS-DMA ring (gemm128p.hip), grid = (tiles, split-K slices, groups) of gemm_bf16_kernel; false: operands it does
// not take (alignment, masked / dropout-accumulate launches) or VLR_GEMM128P=0 - the caller launches gemm_bf16_kernel
bool vlr_gemm128p_try_launch(int layout, const GemmParams& p, dim3 grid, hipStream_t stream);
// fused-epilogue variants (p.fuse = 1 | 2, layout NT): false when the shape does not qualify for the persistent
// continuous-pipeline kernel - the caller then runs the plain GEMM followed by the elementwise kernel
bool vlr_gemm256p_dropacc_try_launch(const GemmParams& p, hipStream_t stream);
// streaming LoRA input-gradient kernel (lora_dx.hip); false: shape not taken
bool vlr_lora_dx_try_launch(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p_drop, uint64_t seed,
                            float scale, int accumulate, const void* bits, long bits_gstride, hipStream_t stream, const unsigned char* rowskip = nullptr);
// streaming row-slab adapter products (lora_rows.hip): mode 0 u = alpha (keep . x) A^T, mode 1 v = dy B; false: shape not taken
bool vlr_lora_rows_try_launch(int mode, int n, const void* X, int ldx, const int* Ks, const void* W, int ldw, void* out, int ldo, int ostride,
                              int M, int r, float alpha, const void* bits, long gbits, int bits_ld, const unsigned char* rowmask, hipStream_t st, bool force = false);
bool vlr_gemm256p_seg_try_launch(const GemmParams& p, hipStream_t stream);
bool vlr_gemm256p_fused_try_launch(const GemmParams& p, hipStream_t stream);
// two TN problems of equal K (the weight gradients of two linears of one layer) as ONE persistent launch; false: shapes it does not take
bool vlr_gemm256p_tn_pair_try_launch(const GemmParams& p0, const GemmParams& p1, hipStream_t stream);
// fuse = 3 (NN): d act = dy . Wdown with the SwiGLU backward applied in the epilogue to gate | up in p.C2 (in place)
bool vlr_gemm256p_swiglu_bwd_try_launch(const GemmParams& p, hipStream_t stream);
// fuse = 4 | 5: the lm-head GEMM with the log-softmax statistics / the logits gradient in its epilogue
bool vlr_gemm256p_lmhead_try_launch(const GemmParams& p, hipStream_t stream);
int vlr_gemm256p_lmhead_parts(int V);      // partial (max, sum) pairs per row the fuse = 4 epilogue writes
