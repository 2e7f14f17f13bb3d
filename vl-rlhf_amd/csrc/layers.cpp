// Composed passes: one LLaMA decoder layer forward / backward and one CLIP encoder layer forward, sequenced on a
// single HIP stream from the kernel-level entry points.  Pure host code: no allocation, no synchronisation.
//
// Decoder layer (transformers LlamaDecoderLayer; call site /root/reference src/vlrlhf/models/Llava/__init__.py:232):
//   xn1 = RMSNorm(x) ; qkv = xn1 Wqkv^T ; RoPE(q,k) ; attn = softmax(q k^T / sqrt(d) + causal/pad) v
//   x_mid = x + attn Wo^T ; xn2 = RMSNorm(x_mid) ; gu = xn2 Wgu^T ; act = silu(g) * u ; x_out = x_mid + act Wdown^T
// The backward is the exact adjoint, weight gradients written (or accumulated) straight into the flat bf16 gradient
// buffer views passed in `g` - no autograd graph, no temporaries beyond `ws`.
#include <math.h>

#include "../../include/vlr.h"
#include "common.h"

// ---- optional second stream for the weight-gradient GEMMs: dgrad (NN) and wgrad (TN) of a layer are independent, so the
// wgrad can run on a side stream and fill the dgrad's last, partly empty wave of workgroups.  Measured (profiles/): +1.5 %
// with the first 256-tile kernel, -2 % once the GEMM dispatcher peels the ragged tile rows itself (gemm.hip) - two
// 128 KiB-LDS kernels only time-share the CUs.  Off by default; VLR_BWD_STREAMS=1 enables it.
#include <stdlib.h>
static hipStream_t g_side = nullptr;
static hipEvent_t g_fork = nullptr, g_done[4] = {nullptr, nullptr, nullptr, nullptr};
static int g_two_streams = -1;
static bool g_side_used = false;

static bool two_streams() {
    if (g_two_streams < 0) {
        const char* e = getenv("VLR_BWD_STREAMS");
        g_two_streams = (e && e[0] == '1') ? 1 : 0;
        if (g_two_streams) {
            if (hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking) != hipSuccess) g_two_streams = 0;
            hipEventCreateWithFlags(&g_fork, hipEventDisableTiming);
            for (int i = 0; i < 4; ++i) hipEventCreateWithFlags(&g_done[i], hipEventDisableTiming);
        }
    }
    return g_two_streams == 1;
}
// side stream continues from the current point of `main`
static hipStream_t fork_side(hipStream_t main) {
    hipEventRecord(g_fork, main);
    hipStreamWaitEvent(g_side, g_fork, 0);
    return g_side;
}
static void side_done(int i) { hipEventRecord(g_done[i], g_side); g_side_used = true; }
static void wait_side(int i, hipStream_t main) { if (g_side_used) hipStreamWaitEvent(main, g_done[i], 0); }

// make `stream` wait for every wgrad GEMM issued on the side stream (call before anything reads the weight gradients)
extern "C" int vlr_layers_join(vlr_stream_t stream) {
    if (g_two_streams == 1 && g_side_used)
        for (int i = 0; i < 4; ++i) hipStreamWaitEvent(stream, g_done[i], 0);
    return VLR_OK;
}

// ---- side streams for the peeled rows of the decoder GEMMs (VlrGemmTail, common.h): one per main stream (the policy and the reference
// pass run on different streams), created on first use.  VLR_GEMM_TAIL=0 switches the overlap off (the peel then runs on the main
// stream as before round 4).
static int g_tail_on = -1;
static struct TailSlot { hipStream_t main, side; hipEvent_t fork, done; } g_tails[4];
static int g_ntails = 0;
static VlrGemmTail* tail_for(hipStream_t main, VlrGemmTail* t) {
    if (g_tail_on < 0) { const char* e = getenv("VLR_GEMM_TAIL"); g_tail_on = (e && e[0] == '1') ? 1 : 0; }
    if (!g_tail_on) return nullptr;
    TailSlot* sl = nullptr;
    for (int i = 0; i < g_ntails; ++i)
        if (g_tails[i].main == main) sl = &g_tails[i];
    if (!sl) {
        if (g_ntails == 4) return nullptr;
        sl = &g_tails[g_ntails];
        sl->main = main;
        if (hipStreamCreateWithFlags(&sl->side, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&sl->fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sl->done, hipEventDisableTiming) != hipSuccess) return nullptr;
        ++g_ntails;
    }
    t->side = sl->side; t->fork = sl->fork; t->done = sl->done; t->M1 = 0; t->used = 0;
    return t;
}

#define CHECK(call)                     \
    do {                                \
        int rc_ = (call);               \
        if (rc_ != VLR_OK) return rc_;  \
    } while (0)

// fp32 residual stream (cfg->resid_f32): x_in / x_mid / x_out are fp32 [M][H]; the norms read them in fp32 and the o_proj / down_proj
// GEMMs add their fp32 accumulators to them without any rounding.  The gradient stream stays bf16.
static int norm_fwd(int f32, const void* x, const void* w, void* y, float* rstd, int M, int H, float eps, hipStream_t st) {
    return f32 ? vlr_rmsnorm_fwd_f32((const float*)x, w, y, rstd, M, H, eps, st) : vlr_rmsnorm_fwd(x, w, y, rstd, M, H, eps, st);
}
static int norm_bwd(int f32, const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx, void* dw,
                    int acc, void* ws, int M, int H, hipStream_t st) {
    return f32 ? vlr_rmsnorm_bwd_f32(dy, (const float*)x, w, rstd, dres, dx, dw, acc, ws, M, H, st)
               : vlr_rmsnorm_bwd(dy, x, w, rstd, dres, dx, dw, acc, ws, M, H, st);
}
// y = a W^T + residual (NT), on the bf16 or the fp32 stream
static int proj_res(int f32, const void* a, const void* W, void* y, const void* res, int M, int N, int K, hipStream_t st) {
    return f32 ? vlr_gemm_bf16_f32res(0, a, W, (float*)y, (const float*)res, M, N, K, K, K, N, N, st)
               : vlr_gemm_bf16(0, a, W, y, nullptr, res, M, N, K, K, K, N, N, 0, 0, 0, st);
}

static inline const char* off(const void* p, size_t elems) { return (const char*)p + elems * 2; }
static inline char* off(void* p, size_t elems) { return (char*)p + elems * 2; }

// keep_for_backward = 0 (no-grad reference / evaluation pass): tensors only the backward reads (gate | up) are not written
extern "C" int vlr_decoder_layer_fwd_ex(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_layer_acts* a,
                                        const void* x_in, const int* pos, const int* key_mask, int batch, int S,
                                        int keep_for_backward, vlr_stream_t st) {
    VLR_REQUIRE(cfg && w && a && x_in && pos, "vlr_decoder_layer_fwd: null argument");
    const int H = cfg->hidden, I = cfg->inter, M = batch * S;
    const int kvh = cfg->kv_heads > 0 ? cfg->kv_heads : cfg->heads;
    VLR_REQUIRE(cfg->heads % kvh == 0, "vlr_decoder_layer_fwd: heads %d is not a multiple of kv_heads %d", cfg->heads, kvh);
    const int Nq = cfg->heads * cfg->head_dim, Nkv = kvh * cfg->head_dim, N = Nq + 2 * Nkv;
    const int rf = cfg->resid_f32;
    CHECK(norm_fwd(rf, x_in, w->ln1, a->xn1, a->rstd1, M, H, cfg->rms_eps, st));
    // q|k|v projection with RoPE applied to the fp32 accumulators in the GEMM epilogue (plain GEMM + rope kernel for the rows /
    // shapes the persistent kernel does not take)
    // (a bias of the fused projection - Qwen c_attn - is added to the accumulators before the rotation)
    CHECK(vlr_gemm_qkv_rope_bias(a->xn1, w->wqkv, w->bqkv, a->qkv, pos, cfg->rope_cos, cfg->rope_sin, M, N, Nq + Nkv, H, H, cfg->head_dim,
                                 cfg->max_pos, st));
    CHECK(vlr_attn_fwd_gqa(a->qkv, off(a->qkv, Nq), off(a->qkv, (size_t)Nq + Nkv), N, a->attn, Nq, a->lse, key_mask, batch, S,
                           cfg->heads, kvh, cfg->head_dim, 1, 1.0f / sqrtf((float)cfg->head_dim), st));
    // o_proj: its peeled last tile rows run on a side stream beside the RMSNorm of the rows that are already there (tail_for above)
    VlrGemmTail tl_, *tl = tail_for(st, &tl_);
    vlr_internal_set_gemm_tail(tl);
    CHECK(proj_res(rf, a->attn, w->wo, a->x_mid, x_in, M, H, Nq, st));
    vlr_internal_set_gemm_tail(nullptr);
    if (tl && tl->used) {
        const size_t xs = rf ? 4 : 2;
        CHECK(norm_fwd(rf, a->x_mid, w->ln2, a->xn2, a->rstd2, tl->M1, H, cfg->rms_eps, st));
        hipStreamWaitEvent(st, tl->done, 0);
        CHECK(norm_fwd(rf, (const char*)a->x_mid + (size_t)tl->M1 * H * xs, w->ln2, off(a->xn2, (size_t)tl->M1 * H), a->rstd2 + tl->M1, M - tl->M1, H,
                       cfg->rms_eps, st));
    } else {
        CHECK(norm_fwd(rf, a->x_mid, w->ln2, a->xn2, a->rstd2, M, H, cfg->rms_eps, st));
    }
    // gate|up projection with act = silu(gate) * up computed in the epilogue
    CHECK(vlr_gemm_swiglu(a->xn2, w->wgu, a->gu, a->act, M, I, H, H, keep_for_backward, st));
    CHECK(proj_res(rf, a->act, w->wdown, a->x_out, a->x_mid, M, H, I, st));
    return VLR_OK;
}
extern "C" int vlr_decoder_layer_fwd(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_layer_acts* a,
                                     const void* x_in, const int* pos, const int* key_mask, int batch, int S,
                                     vlr_stream_t st) {
    return vlr_decoder_layer_fwd_ex(cfg, w, a, x_in, pos, key_mask, batch, S, 1, st);
}

double vlr_internal_tn_pair_saves(int M0, int N0, int M1, int N1);      // gemm.hip: rounds vlr_gemm_bf16_tn_pair saves over two launches on the compute CUs of the moment
extern "C" int vlr_decoder_layer_bwd(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_layer_grads* g,
                                     int accumulate, const vlr_layer_acts* a, const vlr_layer_bwd_ws* ws,
                                     const void* x_in, const void* dx_out, void* dx_in, const int* pos,
                                     const int* key_mask, int batch, int S, vlr_stream_t st) {
    VLR_REQUIRE(cfg && w && g && a && ws && x_in && dx_out && dx_in && pos, "vlr_decoder_layer_bwd: null argument");
    const int H = cfg->hidden, I = cfg->inter, M = batch * S;
    const int kvh = cfg->kv_heads > 0 ? cfg->kv_heads : cfg->heads;
    const int Nq = cfg->heads * cfg->head_dim, Nkv = kvh * cfg->head_dim, N = Nq + 2 * Nkv;
    const bool two = two_streams();
    hipStream_t sd = st;
    // ---- MLP
    if (two) {
        sd = fork_side(st);
        CHECK(vlr_gemm_bf16(2, dx_out, a->act, g->wdown, nullptr, nullptr, H, I, M, H, I, I, 0, 0, accumulate, 0, sd));
        side_done(0);
    }
    CHECK(vlr_gemm_swiglu_bwd(dx_out, w->wdown, a->gu, ws->dact, M, I, H, st));   // gu now holds [dgate | dup]; d act is not materialised
    if (two) {
        sd = fork_side(st);
        CHECK(vlr_gemm_bf16(2, a->gu, a->xn2, g->wgu, nullptr, nullptr, 2 * I, H, M, 2 * I, H, H, 0, 0, accumulate, 0, sd));
        side_done(1);
    } else {
        // dW_gate|up and dW_down as ONE persistent launch: 1376 + 688 output tiles are 8.06 rounds of 256 CUs together, 6 + 3 apart
        CHECK(vlr_gemm_bf16_tn_pair(a->gu, a->xn2, g->wgu, 2 * I, H, 2 * I, H, H, dx_out, a->act, g->wdown, H, I, H, I, I, M, accumulate, st));
    }
    // the data-gradient GEMMs in front of the two RMSNorm backward passes: peeled rows on the side stream, the norm in two row ranges
    VlrGemmTail tl_, *tl = two ? nullptr : tail_for(st, &tl_);
    vlr_internal_set_gemm_tail(tl);
    CHECK(vlr_gemm_bf16(1, a->gu, w->wgu, ws->dxn, nullptr, nullptr, M, H, 2 * I, 2 * I, H, H, 0, 0, 0, 0, st));
    vlr_internal_set_gemm_tail(nullptr);
    if (two) wait_side(2, st);                           // previous layer's dWo GEMM still reads ws->dx_mid
    if (tl && tl->used)
        CHECK(vlr_internal_rmsnorm_bwd_split(ws->dxn, a->x_mid, cfg->resid_f32, w->ln2, a->rstd2, dx_out, ws->dx_mid, g->ln2, accumulate, ws->norm_ws, M, H,
                                             tl->M1, tl->done, st));
    else
        CHECK(norm_bwd(cfg->resid_f32, ws->dxn, a->x_mid, w->ln2, a->rstd2, dx_out, ws->dx_mid, g->ln2, accumulate, ws->norm_ws, M, H, st));
    // ---- attention
    // dW_o is one round of 256 tiles on its own; together with dW_qkv it would be 768 + 256 = 4 whole rounds of ONE persistent launch
    // (vlr_gemm_bf16_tn_pair below; both operands - dx_mid, attn - stay untouched until the end of this call).  Measured NEUTRAL in round 5
    // (565.9 vs 565.9 ms, same box, twice each) on the whole chip; with CUs left to RCCL (240-CU rounds: 4 + 2 apart, 5 together) it saves a
    // round - so: paired when the joint launch saves rounds on the CUs the launches have now (VLR_PAIR_QKVO=1 always, =0 never)
    static int pair_qkvo = -1;
    if (pair_qkvo < 0) { const char* e = getenv("VLR_PAIR_QKVO"); pair_qkvo = !e ? 2 : (e[0] == '1' ? 1 : 0); }
    const bool pair_o = !two && !accumulate && (pair_qkvo == 1 || (pair_qkvo == 2 && vlr_internal_tn_pair_saves(N, H, H, Nq) >= 0.25));
    if (two) { sd = fork_side(st); }
    if (!pair_o) CHECK(vlr_gemm_bf16(2, ws->dx_mid, a->attn, g->wo, nullptr, nullptr, H, Nq, M, H, Nq, Nq, 0, 0, accumulate, 0, sd));
    if (two) side_done(2);
    CHECK(vlr_gemm_bf16(1, ws->dx_mid, w->wo, ws->dattn, nullptr, nullptr, M, Nq, H, H, Nq, Nq, 0, 0, 0, 0, st));
    if (two) wait_side(3, st);                           // previous layer's dWqkv GEMM still reads ws->dqkv
    CHECK(vlr_attn_bwd_gqa(a->qkv, off(a->qkv, Nq), off(a->qkv, (size_t)Nq + Nkv), N, a->attn, ws->dattn, Nq, a->lse, ws->delta,
                           key_mask, ws->dqkv, off(ws->dqkv, Nq), off(ws->dqkv, (size_t)Nq + Nkv), N, batch, S, cfg->heads, kvh,
                           cfg->head_dim, 1, 1.0f / sqrtf((float)cfg->head_dim), st));
    // transpose of the rotation on the q and k column blocks ((Nq + Nkv) / head_dim consecutive heads)
    CHECK(vlr_rope_heads(ws->dqkv, pos, cfg->rope_cos, cfg->rope_sin, M, cfg->heads + kvh, cfg->head_dim, N, cfg->max_pos, 1, st));
    if (two) { sd = fork_side(st); }
    if (pair_o) CHECK(vlr_gemm_bf16_tn_pair(ws->dqkv, a->xn1, g->wqkv, N, H, N, H, H, ws->dx_mid, a->attn, g->wo, H, Nq, H, Nq, Nq, M, 0, st));
    else CHECK(vlr_gemm_bf16(2, ws->dqkv, a->xn1, g->wqkv, nullptr, nullptr, N, H, M, N, H, H, 0, 0, accumulate, 0, sd));
    if (two) side_done(3);
    tl = two ? nullptr : tail_for(st, &tl_);
    vlr_internal_set_gemm_tail(tl);
    CHECK(vlr_gemm_bf16(1, ws->dqkv, w->wqkv, ws->dxn, nullptr, nullptr, M, H, N, N, H, H, 0, 0, 0, 0, st));
    vlr_internal_set_gemm_tail(nullptr);
    if (two) { wait_side(0, st); wait_side(1, st); }     // this layer's dWdown / dWgu read dx_out / gu: done before dx_in (the
                                                         // buffer the NEXT layer overwrites dx_out with) is produced
    if (tl && tl->used)
        CHECK(vlr_internal_rmsnorm_bwd_split(ws->dxn, x_in, cfg->resid_f32, w->ln1, a->rstd1, ws->dx_mid, dx_in, g->ln1, accumulate, ws->norm_ws, M, H,
                                             tl->M1, tl->done, st));
    else
        CHECK(norm_bwd(cfg->resid_f32, ws->dxn, x_in, w->ln1, a->rstd1, ws->dx_mid, dx_in, g->ln1, accumulate, ws->norm_ws, M, H, st));
    return VLR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LoRA (peft lora.Linear.forward: result = base(x) + lora_B(lora_A(dropout(x))) * scaling) on the seven decoder linears,
// un-merged so that lora_dropout works and the frozen base weights double as the reference model (adapter disabled).
// One fused linear group = n sub-targets sharing the input x [M][in]; y/dy [M][n*out]; A [n*r][in]; B [n*out][r];
// u = dropout_t(x) A_t^T is kept for the backward ([M][7r] per layer: qkv | o | gate,up | down).
// Dropout target t of a layer uses seed + t (t = 0..6 in q,k,v,o,gate,up,down order).
// ---------------------------------------------------------------------------------------------------------------------
// outs[t]: output features of sub-target t (they differ under grouped-query attention: q has heads*hd, k and v kv_heads*hd)
// u_t = s * dropout_t(x) A_t^T for the n sub-targets of a group (the B half rides the K loop of the base GEMM: vlr_gemm_*_lora)
// bits != NULL (vlr_lora_weights::mask_bits): the packed keep masks of the n targets are DRAWN here (target t at bits + t * M * in / 8) and
// read by the staged-operand mask of the grouped launch - and again by the backward (lora_group_bwd)
// ustride (0 = r): elements between the u blocks of consecutive sub-targets - the two-adapter layout [u_lora | u_plora] per sub-target
// per-stream device buffer for vlr_rows_tile_list (count + tile indices), grown on demand; NULL: no memory (the dense reductions run)
static int* row_tiles_buf(hipStream_t st, int ints) {
    struct Slot { hipStream_t st; int* p; int cap; };
    static Slot slots[8];
    static int nslots = 0;
    Slot* s = nullptr;
    for (int i = 0; i < nslots; ++i) if (slots[i].st == st) s = &slots[i];
    if (!s) { if (nslots == 8) return nullptr; s = &slots[nslots++]; *s = Slot{st, nullptr, 0}; }
    if (s->cap < ints) {
        if (s->p) { hipStreamSynchronize(st); hipFree(s->p); s->p = nullptr; s->cap = 0; }
        const int cap = ints < 16384 ? 16384 : ints;
        if (hipMalloc((void**)&s->p, (size_t)cap * 4) != hipSuccess) { s->p = nullptr; return nullptr; }
        s->cap = cap;
    }
    return s->p;
}
// 256-row tile flags of a row-restricted adapter (1 = no marked row in the tile) for the adapter-segment GEMMs of a layer pass
// (vlr_gemm_seg_rowskip): per-stream device bytes behind the K-tile list of row_tiles_buf; NULL: VLR_SEG_SKIP=0, M % 256 != 0 never matters
// (the last tile is simply partial), or no memory
static const unsigned char* seg_skip_flags(hipStream_t st, const unsigned char* rowmask, int M) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("VLR_SEG_SKIP"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || !rowmask) return nullptr;
    const int tiles = (M + 255) / 256;
    int* buf = row_tiles_buf(st, M / 64 + 2 + (tiles + 3) / 4);
    if (!buf) return nullptr;
    unsigned char* flags = (unsigned char*)(buf + M / 64 + 2);
    if (vlr_rows_tile_flags(rowmask, M, 256, flags, st) != VLR_OK) return nullptr;
    return flags;
}
static bool row_tiles_on() {      // VLR_ROW_TILES=0: the K reductions of a row-restricted adapter read all token rows (A/B)
    static int on = -1;
    if (on < 0) { const char* e = getenv("VLR_ROW_TILES"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

// lora_rows.hip: the streaming row-slab adapter products (mode 0: u = alpha (keep . x) A^T, mode 1: v = dy B); false = shape not taken
bool vlr_lora_rows_try_launch(int mode, int n, const void* X, int ldx, const int* Ks, const void* W, int ldw, void* out, int ldo, int ostride,
                              int M, int r, float alpha, const void* bits, long gbits, int bits_ld, const unsigned char* rowmask, hipStream_t st, bool force = false);

static int lora_group_a(int n, int r, int in, const void* x, int ldx, const void* A, void* u, int ldu, float scale, float p,
                        uint64_t seed, void* ws_xd, int M, hipStream_t st, const unsigned char* rowmask = nullptr, unsigned char* bits = nullptr,
                        unsigned char* bits_kt = nullptr, int ustride = 0) {
    (void)ws_xd;
    if (ustride == 0) ustride = r;
    struct MaskAfter {       // PLoRA: the adapter acts on the image rows only - zero the other rows of u on the way out
        void* u; int ldu, n, r, us, M; const unsigned char* rm; hipStream_t st;
        int run() const {
            if (!rm) return VLR_OK;
            if (us == r) return vlr_rows_mask(u, ldu, n * r, rm, M, st);
            for (int t = 0; t < n; ++t) { const int e = vlr_rows_mask((char*)u + (size_t)t * us * 2, ldu, r, rm, M, st); if (e) return e; }
            return VLR_OK;
        }
    } after = {u, ldu, n, r, ustride, M, rowmask, st};
    if (p > 0.f && bits) {      // row-major masks (the streaming u kernel, the dx kernel) and the K-tile-blocked transposed ones (dA) in one draw per target
        VLR_REQUIRE(ldx == in, "lora: the dropout mask is indexed over [M][in]; x must be dense (ldx %d, in %d)", ldx, in);
        VLR_REQUIRE(((long)M * in) % 32 == 0, "lora: packed dropout masks need M * in %% 32 == 0 (M %d, in %d)", M, in);
        const long tstride = vlr_dropout_bits_kt_bytes(M, in);
        for (int t = 0; t < n; ++t)
            CHECK(vlr_dropout_bits2(bits + (size_t)t * ((long)M * in / 8), bits_kt ? bits_kt + (size_t)t * tstride : nullptr, M, in, p, seed + t, st));
    }
    // the streaming row-slab kernel (lora_rows.hip): x read once per target at HBM rate, the text rows of a row-restricted adapter zeroed on
    // the way out; shapes it does not take (and the hashing form without packed masks) run the grouped tile GEMMs below
    if (p == 0.f || bits) {
        const int Ks[4] = {in, in, in, in};
        if (n <= 4 && vlr_lora_rows_try_launch(0, n, x, ldx, Ks, A, in, u, ldu, ustride, M, r, p > 0.f ? scale / (1.f - p) : scale, p > 0.f ? bits : nullptr,
                                               (long)M * in / 8, in, rowmask, st))
            return vlr_check_launch("lora_rows(u)");
    }
    if (p > 0.f) {
        // ONE grouped launch for the n sub-targets: target t = group t reads the SAME x with its own keep mask (vlr_dropout(seed + t),
        // zeroed while the operand is staged - drop(x) is never written) against its own A_t; 1 / (1 - p) rides in alpha
        VLR_REQUIRE(ldx == in, "lora: the dropout mask is indexed over [M][in]; x must be dense (ldx %d, in %d)", ldx, in);
        const long gstride = (long)M * in / 8;
        if (rowmask) CHECK(vlr_gemm_grouped_bits_rows(0, x, A, u, M, r, in, ldx, in, ldu, n, 0L, (long)r * in, (long)ustride, scale / (1.f - p), 0, 1, seed, p,
                                                      in, bits, gstride, rowmask, st));      // all-text tiles are skipped; `after` zeroes the text rows
        else CHECK(vlr_gemm_grouped_bits(0, x, A, u, M, r, in, ldx, in, ldu, n, 0L, (long)r * in, (long)ustride, scale / (1.f - p), 0, 1, seed, p, in, bits,
                                         gstride, st));
    } else if (ustride == r) {
        CHECK(vlr_gemm_bf16_scaled(0, x, A, u, nullptr, nullptr, M, n * r, in, ldx, in, ldu, 0, 0, 0, 0, scale, st));
    } else {
        CHECK(vlr_gemm_grouped(0, x, A, u, M, r, in, ldx, in, ldu, n, 0L, (long)r * in, (long)ustride, scale, 0, 0, 0, 0.f, 0, st));
    }
    return after.run();
}

// dx [M][in] already holds dy W; adds the adapter path and writes the adapter gradients
// dx_fresh = 1: dx is WRITTEN (= the adapter term alone; the caller adds dy W afterwards - the fused SwiGLU-backward GEMM of down_proj)
static int lora_group_bwd(int n, int r, int in, const int* outs, const void* x, const void* dy, int lddy, const void* A, const void* B,
                          void* dA, void* dB, const void* u, int ldu, void* v, void* dx, float scale, float p, uint64_t seed,
                          void* ws_xd, int accumulate, int M, hipStream_t st, int dx_fresh = 0, const unsigned char* rowmask = nullptr,
                          const unsigned char* bits = nullptr, const unsigned char* bits_kt = nullptr, const int* ktl = nullptr) {
    const int nr = n * r;
    const long gstride = (long)M * in / 8;       // bytes between the packed keep masks of the group's targets (the forward drew them)
    size_t ofs[4] = {0, 0, 0, 0};
    bool same = true;
    for (int t = 0; t < n; ++t) { ofs[t + 1] = ofs[t] + (size_t)outs[t]; same = same && outs[t] == outs[0]; }
    // dB_t = dy_t^T (s u_t) and v_t = dy_t B_t: the n sub-targets as the groups of one launch each when their widths agree (multi-head
    // attention; grouped-query k / v are narrower than q: one launch per target then)
    const int ng = same ? 1 : n, gs = same ? n : 1;
    // v_t = dy_t B_t for all targets of the group in ONE pass over dy on the streaming row-slab kernel (lora_rows.hip; unmarked rows of a
    // row-restricted adapter come out zero); else per group on the tile GEMMs
    const bool v_done = n <= 4 && vlr_lora_rows_try_launch(1, n, dy, lddy, outs, B, r, v, nr, r, M, r, 1.f, nullptr, 0L, 0, rowmask, st);
    for (int g = 0; g < ng; ++g) {
        const int out = outs[g];
        // (ktl: PLoRA - u and v are zero on the text rows, the K reductions over the token rows read only the 64-row tiles with an image row)
        if (ktl) CHECK(vlr_gemm_grouped_bits_ktiles(2, off(dy, ofs[g]), off(u, (size_t)g * r), off(dB, ofs[g] * r), out, r, M, lddy, ldu, r, gs, (long)out,
                                                    (long)r, (long)out * r, 1.f, accumulate, 0, 0, 0.f, 0, nullptr, 0L, ktl, st));
        else CHECK(vlr_gemm_grouped(2, off(dy, ofs[g]), off(u, (size_t)g * r), off(dB, ofs[g] * r), out, r, M, lddy, ldu, r, gs, (long)out, (long)r,
                                    (long)out * r, 1.f, accumulate, 0, 0, 0.f, 0, st));                  // u is stored scaled
        if (v_done) continue;
        if (rowmask) CHECK(vlr_gemm_grouped_bits_rows(1, off(dy, ofs[g]), off(B, ofs[g] * r), off(v, (size_t)g * r), M, r, out, lddy, r, nr, gs, (long)out,
                                                      (long)out * r, (long)r, 1.f, 0, 0, 0, 0.f, 0, nullptr, 0L, rowmask, st));
        else CHECK(vlr_gemm_grouped(1, off(dy, ofs[g]), off(B, ofs[g] * r), off(v, (size_t)g * r), M, r, out, lddy, r, nr, gs, (long)out,
                                    (long)out * r, (long)r, 1.f, 0, 0, 0, 0.f, 0, st));
    }
    if (rowmask && !v_done) CHECK(vlr_rows_mask(v, nr, nr, rowmask, M, st));      // PLoRA: no gradient flows through the adapter on the text rows
    if (p > 0.f) {
        // dA_t = s / (1 - p) v_t^T (mask_t . x): the n targets as groups, x masked while it is staged (the mask of the forward, regenerated)
        if (bits_kt && ktl) CHECK(vlr_gemm_grouped_bits_ktiles(2, v, x, dA, r, in, M, nr, in, in, n, (long)r, 0L, (long)r * in, scale / (1.f - p), accumulate, 3,
                                                               seed, p, in, bits_kt, vlr_dropout_bits_kt_bytes(M, in), ktl, st));
        else if (bits_kt) CHECK(vlr_gemm_grouped_bits(2, v, x, dA, r, in, M, nr, in, in, n, (long)r, 0L, (long)r * in, scale / (1.f - p), accumulate, 3, seed, p,
                                                      in, bits_kt, vlr_dropout_bits_kt_bytes(M, in), st));
        else CHECK(vlr_gemm_grouped_bits(2, v, x, dA, r, in, M, nr, in, in, n, (long)r, 0L, (long)r * in, scale / (1.f - p), accumulate, 2, seed, p, in,
                                         bits, gstride, st));
        // dx (+)= s / (1 - p) sum_t mask_t . (v_t A_t): ONE pass over dx for the n targets (vlr_gemm_dropout_acc_multi: the streaming kernel
        // of lora_dx.hip, 222 us for q, k, v at [12792 x 4096], r = 128, against 3 x 90 us one target at a time; tools/lora_gemm_bench.py).
        // VLR_LORA_MULTI=0: one pass per target on the 128x128 GEMM kernel (mask in its epilogue)
        static int multi = -1;
        if (multi < 0) { const char* e = getenv("VLR_LORA_MULTI"); multi = (e && e[0] == '0') ? 0 : 1; }
        if (multi || dx_fresh) {
            CHECK(vlr_gemm_dropout_acc_multi_rows(n, v, nr, A, dx, M, in, r, p, seed, scale, dx_fresh ? 0 : 1, bits, gstride, rowmask, st));
        } else {
            VLR_REQUIRE(ws_xd, "lora backward: lora_dropout > 0 needs a scratch buffer [M][in]");
            for (int t = 0; t < n; ++t)
                CHECK(vlr_gemm_dropout_acc_bits(off(v, (size_t)t * r), nr, off(A, (size_t)t * r * in), dx, ws_xd, M, in, r, p, seed + t, scale,
                                                bits ? bits + (size_t)t * gstride : nullptr, st));
        }
    } else {
        CHECK(vlr_gemm_bf16_scaled(2, v, x, dA, nullptr, nullptr, nr, in, M, nr, in, in, 0, 0, accumulate, 0, scale, st));  // dA = s v^T x
        CHECK(vlr_gemm_bf16_scaled(1, v, A, dx, nullptr, nullptr, M, in, nr, nr, in, in, 0, 0, dx_fresh ? 0 : 1, 0, scale, st));     // dx (+)= s v A
    }
    return VLR_OK;
}

static long lora_rowmajor_bytes(int hidden, int inter, int M) { return (((long)M * hidden / 8) * 6 + (long)M * inter / 8 + 63) / 64 * 64; }
extern "C" long vlr_lora_mask_bytes(int hidden, int inter, int M) {
    return lora_rowmajor_bytes(hidden, inter, M) + 6 * vlr_dropout_bits_kt_bytes(M, hidden) + vlr_dropout_bits_kt_bytes(M, inter);
}
static int lora_check(const char* who, const vlr_lora_weights* lw, const void* ws_xd) {
    VLR_REQUIRE(lw->r > 0 && lw->r % 8 == 0, "%s: LoRA rank must be a positive multiple of 8, got %d", who, lw->r);
    VLR_REQUIRE(lw->dropout >= 0.f && lw->dropout < 1.f, "%s: lora_dropout must be in [0,1), got %g", who, (double)lw->dropout);
    (void)ws_xd;      // (ABI v3 kept drop(x) of the seven targets there; since v4 the mask is applied while the operand is staged)
    VLR_REQUIRE(lw->a_qkv && lw->b_qkv && lw->a_o && lw->b_o && lw->a_gu && lw->b_gu && (!lw->a_down == !lw->b_down), "%s: null adapter pointer", who);
    VLR_REQUIRE(lw->qkv_targets == 0 || lw->qkv_targets == 1 || lw->qkv_targets == 3, "%s: qkv_targets must be 1 or 3, got %d", who, lw->qkv_targets);
    return VLR_OK;
}

extern "C" int vlr_decoder_layer_fwd_lora(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                                          const vlr_layer_acts* a, void* u, void* ws_xd, uint64_t seed, const void* x_in,
                                          const int* pos, const int* key_mask, int batch, int S, vlr_stream_t st) {
    return vlr_decoder_layer_fwd_lora_ex(cfg, w, lw, a, u, ws_xd, seed, nullptr, x_in, pos, key_mask, batch, S, st);
}
// rowmask [batch * S] bytes (NULL: every row): the adapters act on the rows whose byte is non-zero only - PLoRA of InternLM-XComposer2
// (reference models/InternLMXC2/build_mlp.py:158-203: im_mask = the image rows)
extern "C" int vlr_decoder_layer_fwd_lora_ex(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                                             const vlr_layer_acts* a, void* u, void* ws_xd, uint64_t seed, const unsigned char* rowmask,
                                             const void* x_in, const int* pos, const int* key_mask, int batch, int S, vlr_stream_t st) {
    VLR_REQUIRE(cfg && w && lw && a && u && x_in && pos, "vlr_decoder_layer_fwd_lora: null argument");
    CHECK(lora_check("vlr_decoder_layer_fwd_lora", lw, ws_xd));
    const int H = cfg->hidden, I = cfg->inter, M = batch * S, r = lw->r, ldu = 7 * r;
    const int kvh = cfg->kv_heads > 0 ? cfg->kv_heads : cfg->heads;
    const int Nq = cfg->heads * cfg->head_dim, Nkv = kvh * cfg->head_dim, N = Nq + 2 * Nkv;
    VLR_REQUIRE(Nq == H, "vlr_decoder_layer_fwd_lora: heads*head_dim != hidden");
    const float sc = lw->scale, p = lw->dropout;
#define XD(seg) (ws_xd)      // ABI v4: one scratch [M][max(hidden, inter)] (the fallback path of vlr_gemm_dropout_acc), not per-target copies
#define MB(t) (lw->mask_bits && p > 0.f ? (unsigned char*)lw->mask_bits + (size_t)(t) * ((size_t)M * H / 8) : nullptr)      // packed keep mask of target t
#define MT(t) (lw->mask_bits && p > 0.f ? (unsigned char*)lw->mask_bits + lora_rowmajor_bytes(H, I, M) + (size_t)(t) * (size_t)vlr_dropout_bits_kt_bytes(M, H) : nullptr)   // ... K-tile-blocked transposed
    const int rf = cfg->resid_f32;
    // a row-restricted adapter (PLoRA alone): row tiles without a marked row skip the WHOLE segment (u is zero there)
    const unsigned char* skipf = (r % 64 == 0) ? seg_skip_flags(st, rowmask, M) : nullptr;
    CHECK(norm_fwd(rf, x_in, w->ln1, a->xn1, a->rstd1, M, H, cfg->rms_eps, st));
    const int nq = lw->qkv_targets == 1 ? 1 : 3;             // one adapter over the fused projection (Qwen c_attn) or q, k, v separately
    CHECK(lora_group_a(nq, r, H, a->xn1, H, lw->a_qkv, u, ldu, sc, p, seed + 0, ws_xd, M, st, rowmask, MB(0), MT(0)));
    CHECK(vlr_gemm_seg_rowskip(skipf, 0));
    CHECK(vlr_gemm_qkv_rope_lora(a->xn1, w->wqkv, w->bqkv, a->qkv, pos, cfg->rope_cos, cfg->rope_sin, M, N, Nq + Nkv, H, H,
                                 cfg->head_dim, cfg->max_pos, u, ldu, lw->b_qkv, r, nq == 1 ? N : Nq, nq == 1 ? 0 : Nkv, st));
    CHECK(vlr_attn_fwd_gqa(a->qkv, off(a->qkv, Nq), off(a->qkv, (size_t)Nq + Nkv), N, a->attn, Nq, a->lse, key_mask, batch, S,
                           cfg->heads, kvh, cfg->head_dim, 1, 1.0f / sqrtf((float)cfg->head_dim), st));
    CHECK(lora_group_a(1, r, H, a->attn, H, lw->a_o, off(u, 3 * (size_t)r), ldu, sc, p, seed + 3, XD(3), M, st, rowmask, MB(3), MT(3)));
    CHECK(vlr_gemm_seg_rowskip(skipf, 0));
    if (rf) CHECK(vlr_gemm_lora_f32res(a->attn, H, w->wo, (float*)a->x_mid, H, (const float*)x_in, H, M, H, H, off(u, 3 * (size_t)r), ldu, lw->b_o, r, st));
    else CHECK(vlr_gemm_lora(a->attn, H, w->wo, a->x_mid, H, x_in, H, M, H, H, off(u, 3 * (size_t)r), ldu, lw->b_o, r, st));
    CHECK(norm_fwd(rf, a->x_mid, w->ln2, a->xn2, a->rstd2, M, H, cfg->rms_eps, st));
    CHECK(lora_group_a(2, r, H, a->xn2, H, lw->a_gu, off(u, 4 * (size_t)r), ldu, sc, p, seed + 4, XD(4), M, st, rowmask, MB(4), MT(4)));
    CHECK(vlr_gemm_seg_rowskip(skipf, 0));
    CHECK(vlr_gemm_swiglu_lora(a->xn2, w->wgu, a->gu, a->act, M, I, H, H, off(u, 4 * (size_t)r), ldu, lw->b_gu, r, st));
    if (lw->a_down) {
        CHECK(lora_group_a(1, r, I, a->act, I, lw->a_down, off(u, 6 * (size_t)r), ldu, sc, p, seed + 6, XD(6), M, st, rowmask, MB(6), MT(6)));
        CHECK(vlr_gemm_seg_rowskip(skipf, 0));
        if (rf) CHECK(vlr_gemm_lora_f32res(a->act, I, w->wdown, (float*)a->x_out, H, (const float*)a->x_mid, H, M, H, I, off(u, 6 * (size_t)r), ldu, lw->b_down, r, st));
        else CHECK(vlr_gemm_lora(a->act, I, w->wdown, a->x_out, H, a->x_mid, H, M, H, I, off(u, 6 * (size_t)r), ldu, lw->b_down, r, st));
    } else {
        CHECK(proj_res(rf, a->act, w->wdown, a->x_out, a->x_mid, M, H, I, st));
    }
    return VLR_OK;
}

extern "C" int vlr_decoder_layer_bwd_lora(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                                          const vlr_lora_grads* lg, int accumulate, const vlr_layer_acts* a, const void* u,
                                          const vlr_layer_bwd_ws* ws, void* ws_v, void* ws_xd, uint64_t seed, const void* x_in,
                                          const void* dx_out, void* dx_in, const int* pos, const int* key_mask, int batch,
                                          int S, vlr_stream_t st) {
    return vlr_decoder_layer_bwd_lora_ex(cfg, w, nullptr, lw, lg, accumulate, a, u, ws, ws_v, ws_xd, seed, nullptr, x_in, dx_out, dx_in, pos,
                                         key_mask, batch, S, st);
}
// g != NULL: the base weights are trainable too (their gradients as in vlr_decoder_layer_bwd) - the FULL fine-tune of a decoder whose
// linears carry adapters as base-model weights (InternLM-XComposer2's PLoRA); rowmask as in vlr_decoder_layer_fwd_lora_ex
extern "C" int vlr_decoder_layer_bwd_lora_ex(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_layer_grads* g,
                                             const vlr_lora_weights* lw, const vlr_lora_grads* lg, int accumulate, const vlr_layer_acts* a,
                                             const void* u, const vlr_layer_bwd_ws* ws, void* ws_v, void* ws_xd, uint64_t seed,
                                             const unsigned char* rowmask, const void* x_in, const void* dx_out, void* dx_in, const int* pos,
                                             const int* key_mask, int batch, int S, vlr_stream_t st) {
    VLR_REQUIRE(cfg && w && lw && lg && a && u && ws && ws_v && x_in && dx_out && dx_in && pos, "vlr_decoder_layer_bwd_lora: null argument");
    CHECK(lora_check("vlr_decoder_layer_bwd_lora", lw, ws_xd));
    const int H = cfg->hidden, I = cfg->inter, M = batch * S, r = lw->r, ldu = 7 * r;
    const int kvh = cfg->kv_heads > 0 ? cfg->kv_heads : cfg->heads;
    const int Nq = cfg->heads * cfg->head_dim, Nkv = kvh * cfg->head_dim, N = Nq + 2 * Nkv;
    VLR_REQUIRE(Nq == H, "vlr_decoder_layer_bwd_lora: heads*head_dim != hidden");
    const int o_qkv[3] = {Nq, Nkv, Nkv}, o_h[1] = {H}, o_gu[2] = {I, I};
    const float sc = lw->scale, p = lw->dropout;
    const int* ktl = nullptr;         // PLoRA: the 64-row K tiles with an image row, for the adapter-gradient reductions over the token rows
    if (rowmask && M % 64 == 0 && lg->a_qkv && row_tiles_on()) {
        int* buf = row_tiles_buf(st, M / 64 + 1);
        if (buf) { CHECK(vlr_rows_tile_list(rowmask, M, buf, st)); ktl = buf; }
    }
#define XD(seg) (ws_xd)      // ABI v4: one scratch [M][max(hidden, inter)] (the fallback path of vlr_gemm_dropout_acc), not per-target copies
#define MB(t) (lw->mask_bits && p > 0.f ? (unsigned char*)lw->mask_bits + (size_t)(t) * ((size_t)M * H / 8) : nullptr)      // packed keep mask of target t
#define MT(t) (lw->mask_bits && p > 0.f ? (unsigned char*)lw->mask_bits + lora_rowmajor_bytes(H, I, M) + (size_t)(t) * (size_t)vlr_dropout_bits_kt_bytes(M, H) : nullptr)   // ... K-tile-blocked transposed
    // ---- MLP
    static int fuse_down = -1;     // adapter term of down_proj first, then the dgrad GEMM with the SwiGLU backward in its epilogue (VLR_LORA_FUSE_DOWN=0: three separate kernels)
    if (fuse_down < 0) { const char* e = getenv("VLR_LORA_FUSE_DOWN"); fuse_down = (e && e[0] == '0') ? 0 : 1; }
    if (lw->a_down && fuse_down) {
        // history: SLOWER than the three separate kernels in round 3 (38.8 ms against 25.5 + 7.7 per step: the addend is a third 16-byte
        // load stream in an epilogue that already reads gate | up), equal in round 4 after the epilogue staging, 1.7 ms FASTER in round 5
        // (482.3 against 484.1 ms, same box, twice each): the default now - no swiglu_bwd_kernel launch under LoRA
        CHECK(lora_group_bwd(1, r, I, o_h, a->act, dx_out, H, lw->a_down, lw->b_down, lg->a_down, lg->b_down, off(u, 6 * (size_t)r), ldu, ws_v,
                             ws->dact, sc, p, seed + 6, XD(6), accumulate, M, st, 1, rowmask, MB(6), MT(6), ktl));
        CHECK(vlr_gemm_swiglu_bwd_add(dx_out, w->wdown, a->gu, ws->dact, ws->dact, M, I, H, st));   // gu now holds [dgate | dup]
    } else if (lw->a_down) {
        CHECK(vlr_gemm_bf16(1, dx_out, w->wdown, ws->dact, nullptr, nullptr, M, I, H, H, I, I, 0, 0, 0, 0, st));
        CHECK(lora_group_bwd(1, r, I, o_h, a->act, dx_out, H, lw->a_down, lw->b_down, lg->a_down, lg->b_down, off(u, 6 * (size_t)r), ldu, ws_v,
                             ws->dact, sc, p, seed + 6, XD(6), accumulate, M, st, 0, rowmask, MB(6), MT(6), ktl));
        CHECK(vlr_swiglu_bwd(a->gu, ws->dact, M, I, st));   // gu now holds [dgate | dup]
    } else {
        CHECK(vlr_gemm_swiglu_bwd(dx_out, w->wdown, a->gu, ws->dact, M, I, H, st));
    }
    if (g) CHECK(vlr_gemm_bf16_tn_pair(a->gu, a->xn2, g->wgu, 2 * I, H, 2 * I, H, H, dx_out, a->act, g->wdown, H, I, H, I, I, M, accumulate, st));
    CHECK(vlr_gemm_bf16(1, a->gu, w->wgu, ws->dxn, nullptr, nullptr, M, H, 2 * I, 2 * I, H, H, 0, 0, 0, 0, st));
    CHECK(lora_group_bwd(2, r, H, o_gu, a->xn2, a->gu, 2 * I, lw->a_gu, lw->b_gu, lg->a_gu, lg->b_gu, off(u, 4 * (size_t)r), ldu, ws_v,
                         ws->dxn, sc, p, seed + 4, XD(4), accumulate, M, st, 0, rowmask, MB(4), MT(4), ktl));
    CHECK(norm_bwd(cfg->resid_f32, ws->dxn, a->x_mid, w->ln2, a->rstd2, dx_out, ws->dx_mid, g ? g->ln2 : nullptr, g ? accumulate : 0, ws->norm_ws, M, H, st));
    // ---- attention
    if (g) CHECK(vlr_gemm_bf16(2, ws->dx_mid, a->attn, g->wo, nullptr, nullptr, H, Nq, M, H, Nq, Nq, 0, 0, accumulate, 0, st));
    CHECK(vlr_gemm_bf16(1, ws->dx_mid, w->wo, ws->dattn, nullptr, nullptr, M, H, H, H, H, H, 0, 0, 0, 0, st));
    CHECK(lora_group_bwd(1, r, H, o_h, a->attn, ws->dx_mid, H, lw->a_o, lw->b_o, lg->a_o, lg->b_o, off(u, 3 * (size_t)r), ldu, ws_v,
                         ws->dattn, sc, p, seed + 3, XD(3), accumulate, M, st, 0, rowmask, MB(3), MT(3), ktl));
    CHECK(vlr_attn_bwd_gqa(a->qkv, off(a->qkv, Nq), off(a->qkv, (size_t)Nq + Nkv), N, a->attn, ws->dattn, Nq, a->lse, ws->delta,
                           key_mask, ws->dqkv, off(ws->dqkv, Nq), off(ws->dqkv, (size_t)Nq + Nkv), N, batch, S, cfg->heads, kvh,
                           cfg->head_dim, 1, 1.0f / sqrtf((float)cfg->head_dim), st));
    CHECK(vlr_rope_heads(ws->dqkv, pos, cfg->rope_cos, cfg->rope_sin, M, cfg->heads + kvh, cfg->head_dim, N, cfg->max_pos, 1, st));
    if (g) CHECK(vlr_gemm_bf16(2, ws->dqkv, a->xn1, g->wqkv, nullptr, nullptr, N, H, M, N, H, H, 0, 0, accumulate, 0, st));
    CHECK(vlr_gemm_bf16(1, ws->dqkv, w->wqkv, ws->dxn, nullptr, nullptr, M, H, N, N, H, H, 0, 0, 0, 0, st));
    const int o_all[1] = {N};
    const int nq = lw->qkv_targets == 1 ? 1 : 3;
    CHECK(lora_group_bwd(nq, r, H, nq == 1 ? o_all : o_qkv, a->xn1, ws->dqkv, N, lw->a_qkv, lw->b_qkv, lg->a_qkv, lg->b_qkv, u, ldu, ws_v, ws->dxn,
                         sc, p, seed + 0, ws_xd, accumulate, M, st, 0, rowmask, MB(0), MT(0), ktl));
    CHECK(norm_bwd(cfg->resid_f32, ws->dxn, x_in, w->ln1, a->rstd1, ws->dx_mid, dx_in, g ? g->ln1 : nullptr, g ? accumulate : 0, ws->norm_ws, M, H, st));
    return VLR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// TWO adapters per linear: peft LoRA (trainable, every row) stacked on PLoRA (base-model weights, frozen here, image rows only) - the
// configuration reference scripts/dpo_internlmxc2vl7b.sh ships (--use_lora True over the PLoRA decoder of
// models/InternLMXC2/build_mlp.py:158-203; LoraConfig from utils/auto_load.py:559-571).
//   y = W x + s_l B_l A_l drop_l(x) + [image rows] s_p B_p A_p drop_p(x)  =  [x | u_l | u_p] . [W | B_l | B_p]^T
// i.e. ONE adapter segment of rank R = r_l + r_p in the K loop of the fused projections: per sub-target the u block is [u_lora | u_plora]
// (u [M][7R]) and the B operand is the row-wise concatenation [B_lora | B_plora] (vlr_lora_concat_b, built by the caller once per
// optimizer step: B_plora is frozen, B_lora moves).  Dropout: the two adapters draw independent masks (their own seed and packed-mask
// buffer), both indexed over the full [M][in] operand.  Backward: v = dy . [B_l | B_p] in one grouped product; the LoRA half feeds
// dA_l / dB_l and its input-gradient term, the PLoRA half (text rows zeroed) only its input-gradient term.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lora_concat_b_kernel(const bf16_t* __restrict__ b1, int r1, const bf16_t* __restrict__ b2, int r2,
                                                            bf16_t* __restrict__ out, long rows) {
    const int R = r1 + r2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * R; i += (long)gridDim.x * 256) {
        const long row = i / R;
        const int c = (int)(i % R);
        out[i] = c < r1 ? b1[row * r1 + c] : b2[row * r2 + (c - r1)];
    }
}
extern "C" int vlr_lora_concat_b(const void* b1, int r1, const void* b2, int r2, void* out, long rows, vlr_stream_t st) {
    VLR_REQUIRE(b1 && b2 && out && r1 > 0 && r2 > 0 && rows > 0, "vlr_lora_concat_b: bad arguments");
    long blocks = (rows * (r1 + r2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(lora_concat_b_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, (const bf16_t*)b1, r1, (const bf16_t*)b2, r2,
                       (bf16_t*)out, rows);
    return vlr_check_launch("vlr_lora_concat_b");
}

struct Lora2Side {          // one adapter's view of a group
    int r; float scale, p; uint64_t seed; const void* A; void* dA; void* dB; const unsigned char* bits; const unsigned char* bits_kt;
};
// dx [M][in] already holds dy W.  L: the trainable adapter (dA / dB written), P: the frozen row-masked one (dA = dB = NULL).
// dx_fresh = 1: dx is WRITTEN (the adapter terms alone - the first one, LoRA's, covers every row; the caller adds dy W afterwards)
static int lora2_group_bwd(int n, int in, const int* outs, const void* x, const void* dy, int lddy, const Lora2Side& L, const Lora2Side& P,
                           const void* Bcomb, const void* u, int ldu, void* v, void* dx, int accumulate, int M, const unsigned char* rowmask,
                           hipStream_t st, int dx_fresh = 0) {
    int acc_dx = dx_fresh ? 0 : 1;
    const int R = L.r + P.r, nR = n * R;
    const long gstride = (long)M * in / 8;
    size_t ofs[4] = {0, 0, 0, 0};
    bool same = true;
    for (int t = 0; t < n; ++t) { ofs[t + 1] = ofs[t] + (size_t)outs[t]; same = same && outs[t] == outs[0]; }
    const int ng = same ? 1 : n, gs = same ? n : 1;
    // (v_t = dy_t [B_l | B_p]_t as two launches of the streaming row-slab kernel - LoRA columns on every row, PLoRA columns on the image rows -
    // measured 4.5 ms per step SLOWER than the one grouped tile GEMM over all R columns: dy is read twice; gpurun_out r06_rows_auto)
    for (int g = 0; g < ng; ++g) {
        const int out = outs[g];
        // v_t = dy_t [B_l | B_p]_t  ([M][R] per sub-target, sub-targets R apart)
        CHECK(vlr_gemm_grouped(1, off(dy, ofs[g]), off(Bcomb, ofs[g] * R), off(v, (size_t)g * R), M, R, out, lddy, R, nR, gs, (long)out,
                               (long)out * R, (long)R, 1.f, 0, 0, 0, 0.f, 0, st));
        // dB_l,t = dy_t^T u_l,t   (u is stored scaled; the LoRA block is the first r_l columns of the sub-target's u block)
        CHECK(vlr_gemm_grouped(2, off(dy, ofs[g]), off(u, (size_t)g * R), off(L.dB, ofs[g] * L.r), out, L.r, M, lddy, ldu, L.r, gs, (long)out,
                               (long)R, (long)out * L.r, 1.f, accumulate, 0, 0, 0.f, 0, st));
    }
    if (rowmask)
        for (int t = 0; t < n; ++t) CHECK(vlr_rows_mask(off(v, (size_t)t * R + L.r), nR, P.r, rowmask, M, st));      // no gradient through PLoRA on the text rows
    for (int side = 0; side < 2; ++side) {
        const Lora2Side& a = side == 0 ? L : P;
        const size_t c0 = side == 0 ? 0 : (size_t)L.r;                 // column of this adapter's block inside a sub-target's [R]
        for (int t = 0; t < n; ++t) {
            const void* vt = off(v, (size_t)t * R + c0);
            const void* At = off(a.A, (size_t)t * a.r * in);
            if (a.p > 0.f) {
                const unsigned char* bt = a.bits ? a.bits + (size_t)t * gstride : nullptr;
                if (a.dA) {
                    if (a.bits_kt) CHECK(vlr_gemm_grouped_bits(2, vt, x, off(a.dA, (size_t)t * a.r * in), a.r, in, M, nR, in, in, 1, 0L, 0L, 0L,
                                                               a.scale / (1.f - a.p), accumulate, 3, a.seed + t, a.p, in,
                                                               a.bits_kt + (size_t)t * vlr_dropout_bits_kt_bytes(M, in), 0L, st));
                    else CHECK(vlr_gemm_grouped_bits(2, vt, x, off(a.dA, (size_t)t * a.r * in), a.r, in, M, nR, in, in, 1, 0L, 0L, 0L,
                                                     a.scale / (1.f - a.p), accumulate, 2, a.seed + t, a.p, in, bt, 0L, st));
                }
                CHECK(vlr_gemm_dropout_acc_multi_rows(1, vt, nR, At, dx, M, in, a.r, a.p, a.seed + t, a.scale, acc_dx, bt, gstride, side == 1 ? rowmask : nullptr, st));
            } else {
                if (a.dA) CHECK(vlr_gemm_bf16_scaled(2, vt, x, off(a.dA, (size_t)t * a.r * in), nullptr, nullptr, a.r, in, M, nR, in, in, 0, 0,
                                                     accumulate, 0, a.scale, st));
                CHECK(vlr_gemm_bf16_scaled(1, vt, At, dx, nullptr, nullptr, M, in, a.r, nR, in, in, 0, 0, acc_dx, 0, a.scale, st));
            }
            acc_dx = 1;
        }
    }
    return VLR_OK;
}

static int lora2_check(const char* who, const vlr_lora_weights* lw, const vlr_lora_weights* pw, const vlr_lora_bcomb* bc) {
    CHECK(lora_check(who, lw, nullptr));
    CHECK(lora_check(who, pw, nullptr));
    VLR_REQUIRE(bc && bc->qkv && bc->o && bc->gu, "%s: null concatenated B", who);
    VLR_REQUIRE((lw->qkv_targets == 1) == (pw->qkv_targets == 1), "%s: the two adapters must split the qkv projection the same way", who);
    VLR_REQUIRE(!lw->a_down == !pw->a_down && (!lw->a_down || bc->down), "%s: down-projection adapters must be both present or both absent", who);
    return VLR_OK;
}
#define MB2(w_, t) ((w_)->mask_bits && (w_)->dropout > 0.f ? (unsigned char*)(w_)->mask_bits + (size_t)(t) * ((size_t)M * H / 8) : nullptr)
#define MT2(w_, t) ((w_)->mask_bits && (w_)->dropout > 0.f ? (unsigned char*)(w_)->mask_bits + lora_rowmajor_bytes(H, I, M) + (size_t)(t) * (size_t)vlr_dropout_bits_kt_bytes(M, H) : nullptr)

extern "C" int vlr_decoder_layer_fwd_lora2(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                                           const vlr_lora_weights* pw, const vlr_lora_bcomb* bc, const vlr_layer_acts* a, void* u,
                                           uint64_t seed_l, uint64_t seed_p, const unsigned char* rowmask, const void* x_in, const int* pos,
                                           const int* key_mask, int batch, int S, vlr_stream_t st) {
    VLR_REQUIRE(cfg && w && lw && pw && a && u && x_in && pos, "vlr_decoder_layer_fwd_lora2: null argument");
    CHECK(lora2_check("vlr_decoder_layer_fwd_lora2", lw, pw, bc));
    const int H = cfg->hidden, I = cfg->inter, M = batch * S, rl = lw->r, rp = pw->r, R = rl + rp, ldu = 7 * R;
    const int kvh = cfg->kv_heads > 0 ? cfg->kv_heads : cfg->heads;
    const int Nq = cfg->heads * cfg->head_dim, Nkv = kvh * cfg->head_dim, N = Nq + 2 * Nkv;
    VLR_REQUIRE(Nq == H, "vlr_decoder_layer_fwd_lora2: heads*head_dim != hidden");
    const int rf = cfg->resid_f32;
    const int nq = lw->qkv_targets == 1 ? 1 : 3;
    // group g of the layer: u columns [c, c + n R); LoRA block first, PLoRA block behind it (row-masked)
    auto both = [&](int n, int in, const void* x, const void* Al, const void* Ap, size_t c, int t0) -> int {
        CHECK(lora_group_a(n, rl, in, x, in, Al, off(u, c), ldu, lw->scale, lw->dropout, seed_l + t0, nullptr, M, st, nullptr, MB2(lw, t0), MT2(lw, t0), R));
        CHECK(lora_group_a(n, rp, in, x, in, Ap, off(u, c + rl), ldu, pw->scale, pw->dropout, seed_p + t0, nullptr, M, st, rowmask, MB2(pw, t0), MT2(pw, t0), R));
        return VLR_OK;
    };
    // row tiles without an image row run only the LoRA part (the first rl of every sub-target's R K elements) of the [B_lora | B_plora]
    // segment: the PLoRA block of u is zero there (45 % of the row tiles at 490 x 490 / max_length 1024)
    const unsigned char* skipf = (rl % 64 == 0 && R % 64 == 0) ? seg_skip_flags(st, rowmask, M) : nullptr;
    CHECK(norm_fwd(rf, x_in, w->ln1, a->xn1, a->rstd1, M, H, cfg->rms_eps, st));
    CHECK(both(nq, H, a->xn1, lw->a_qkv, pw->a_qkv, 0, 0));
    CHECK(vlr_gemm_seg_rowskip(skipf, rl));
    CHECK(vlr_gemm_qkv_rope_lora(a->xn1, w->wqkv, w->bqkv, a->qkv, pos, cfg->rope_cos, cfg->rope_sin, M, N, Nq + Nkv, H, H,
                                 cfg->head_dim, cfg->max_pos, u, ldu, bc->qkv, R, nq == 1 ? N : Nq, nq == 1 ? 0 : Nkv, st));
    CHECK(vlr_attn_fwd_gqa(a->qkv, off(a->qkv, Nq), off(a->qkv, (size_t)Nq + Nkv), N, a->attn, Nq, a->lse, key_mask, batch, S,
                           cfg->heads, kvh, cfg->head_dim, 1, 1.0f / sqrtf((float)cfg->head_dim), st));
    CHECK(both(1, H, a->attn, lw->a_o, pw->a_o, 3 * (size_t)R, 3));
    CHECK(vlr_gemm_seg_rowskip(skipf, rl));
    if (rf) CHECK(vlr_gemm_lora_f32res(a->attn, H, w->wo, (float*)a->x_mid, H, (const float*)x_in, H, M, H, H, off(u, 3 * (size_t)R), ldu, bc->o, R, st));
    else CHECK(vlr_gemm_lora(a->attn, H, w->wo, a->x_mid, H, x_in, H, M, H, H, off(u, 3 * (size_t)R), ldu, bc->o, R, st));
    CHECK(norm_fwd(rf, a->x_mid, w->ln2, a->xn2, a->rstd2, M, H, cfg->rms_eps, st));
    CHECK(both(2, H, a->xn2, lw->a_gu, pw->a_gu, 4 * (size_t)R, 4));
    CHECK(vlr_gemm_seg_rowskip(skipf, rl));
    CHECK(vlr_gemm_swiglu_lora(a->xn2, w->wgu, a->gu, a->act, M, I, H, H, off(u, 4 * (size_t)R), ldu, bc->gu, R, st));
    if (lw->a_down) {
        CHECK(both(1, I, a->act, lw->a_down, pw->a_down, 6 * (size_t)R, 6));
        CHECK(vlr_gemm_seg_rowskip(skipf, rl));
        if (rf) CHECK(vlr_gemm_lora_f32res(a->act, I, w->wdown, (float*)a->x_out, H, (const float*)a->x_mid, H, M, H, I, off(u, 6 * (size_t)R), ldu, bc->down, R, st));
        else CHECK(vlr_gemm_lora(a->act, I, w->wdown, a->x_out, H, a->x_mid, H, M, H, I, off(u, 6 * (size_t)R), ldu, bc->down, R, st));
    } else {
        CHECK(proj_res(rf, a->act, w->wdown, a->x_out, a->x_mid, M, H, I, st));
    }
    return VLR_OK;
}

extern "C" int vlr_decoder_layer_bwd_lora2(const vlr_llama_cfg* cfg, const vlr_layer_weights* w, const vlr_lora_weights* lw,
                                           const vlr_lora_grads* lg, const vlr_lora_weights* pw, const vlr_lora_bcomb* bc, int accumulate,
                                           const vlr_layer_acts* a, const void* u, const vlr_layer_bwd_ws* ws, void* ws_v, uint64_t seed_l,
                                           uint64_t seed_p, const unsigned char* rowmask, const void* x_in, const void* dx_out, void* dx_in,
                                           const int* pos, const int* key_mask, int batch, int S, vlr_stream_t st) {
    VLR_REQUIRE(cfg && w && lw && lg && pw && a && u && ws && ws_v && x_in && dx_out && dx_in && pos, "vlr_decoder_layer_bwd_lora2: null argument");
    CHECK(lora2_check("vlr_decoder_layer_bwd_lora2", lw, pw, bc));
    const int H = cfg->hidden, I = cfg->inter, M = batch * S, rl = lw->r, rp = pw->r, R = rl + rp, ldu = 7 * R;
    const int kvh = cfg->kv_heads > 0 ? cfg->kv_heads : cfg->heads;
    const int Nq = cfg->heads * cfg->head_dim, Nkv = kvh * cfg->head_dim, N = Nq + 2 * Nkv;
    VLR_REQUIRE(Nq == H, "vlr_decoder_layer_bwd_lora2: heads*head_dim != hidden");
    const int o_qkv[3] = {Nq, Nkv, Nkv}, o_h[1] = {H}, o_gu[2] = {I, I}, o_all[1] = {N};
    const int nq = lw->qkv_targets == 1 ? 1 : 3;
    auto side = [&](const vlr_lora_weights* x_, const void* A, void* dA, void* dB, uint64_t seed, int t0) {
        return Lora2Side{x_->r, x_->scale, x_->dropout, seed + (uint64_t)t0, A, dA, dB, MB2(x_, t0), MT2(x_, t0)};
    };
    // ---- MLP
    // (as vlr_decoder_layer_bwd_lora_ex: the adapter terms of down_proj first, then the dgrad GEMM with the SwiGLU backward in its epilogue and
    // those terms as its addend - no separate pass over [M][2I]; VLR_LORA_FUSE_DOWN=0: three kernels)
    static int fuse_down = -1;
    if (fuse_down < 0) { const char* e = getenv("VLR_LORA_FUSE_DOWN"); fuse_down = (e && e[0] == '0') ? 0 : 1; }
    if (lw->a_down && fuse_down) {
        CHECK(lora2_group_bwd(1, I, o_h, a->act, dx_out, H, side(lw, lw->a_down, lg->a_down, lg->b_down, seed_l, 6),
                              side(pw, pw->a_down, nullptr, nullptr, seed_p, 6), bc->down, off(u, 6 * (size_t)R), ldu, ws_v, ws->dact, accumulate, M,
                              rowmask, st, 1));
        CHECK(vlr_gemm_swiglu_bwd_add(dx_out, w->wdown, a->gu, ws->dact, ws->dact, M, I, H, st));   // gu now holds [dgate | dup]
    } else if (lw->a_down) {
        CHECK(vlr_gemm_bf16(1, dx_out, w->wdown, ws->dact, nullptr, nullptr, M, I, H, H, I, I, 0, 0, 0, 0, st));
        CHECK(lora2_group_bwd(1, I, o_h, a->act, dx_out, H, side(lw, lw->a_down, lg->a_down, lg->b_down, seed_l, 6),
                              side(pw, pw->a_down, nullptr, nullptr, seed_p, 6), bc->down, off(u, 6 * (size_t)R), ldu, ws_v, ws->dact, accumulate, M,
                              rowmask, st));
        CHECK(vlr_swiglu_bwd(a->gu, ws->dact, M, I, st));   // gu now holds [dgate | dup]
    } else {
        CHECK(vlr_gemm_swiglu_bwd(dx_out, w->wdown, a->gu, ws->dact, M, I, H, st));
    }
    CHECK(vlr_gemm_bf16(1, a->gu, w->wgu, ws->dxn, nullptr, nullptr, M, H, 2 * I, 2 * I, H, H, 0, 0, 0, 0, st));
    CHECK(lora2_group_bwd(2, H, o_gu, a->xn2, a->gu, 2 * I, side(lw, lw->a_gu, lg->a_gu, lg->b_gu, seed_l, 4), side(pw, pw->a_gu, nullptr, nullptr, seed_p, 4),
                          bc->gu, off(u, 4 * (size_t)R), ldu, ws_v, ws->dxn, accumulate, M, rowmask, st));
    CHECK(norm_bwd(cfg->resid_f32, ws->dxn, a->x_mid, w->ln2, a->rstd2, dx_out, ws->dx_mid, nullptr, 0, ws->norm_ws, M, H, st));
    // ---- attention
    CHECK(vlr_gemm_bf16(1, ws->dx_mid, w->wo, ws->dattn, nullptr, nullptr, M, H, H, H, H, H, 0, 0, 0, 0, st));
    CHECK(lora2_group_bwd(1, H, o_h, a->attn, ws->dx_mid, H, side(lw, lw->a_o, lg->a_o, lg->b_o, seed_l, 3), side(pw, pw->a_o, nullptr, nullptr, seed_p, 3),
                          bc->o, off(u, 3 * (size_t)R), ldu, ws_v, ws->dattn, accumulate, M, rowmask, st));
    CHECK(vlr_attn_bwd_gqa(a->qkv, off(a->qkv, Nq), off(a->qkv, (size_t)Nq + Nkv), N, a->attn, ws->dattn, Nq, a->lse, ws->delta,
                           key_mask, ws->dqkv, off(ws->dqkv, Nq), off(ws->dqkv, (size_t)Nq + Nkv), N, batch, S, cfg->heads, kvh,
                           cfg->head_dim, 1, 1.0f / sqrtf((float)cfg->head_dim), st));
    CHECK(vlr_rope_heads(ws->dqkv, pos, cfg->rope_cos, cfg->rope_sin, M, cfg->heads + kvh, cfg->head_dim, N, cfg->max_pos, 1, st));
    CHECK(vlr_gemm_bf16(1, ws->dqkv, w->wqkv, ws->dxn, nullptr, nullptr, M, H, N, N, H, H, 0, 0, 0, 0, st));
    CHECK(lora2_group_bwd(nq, H, nq == 1 ? o_all : o_qkv, a->xn1, ws->dqkv, N, side(lw, lw->a_qkv, lg->a_qkv, lg->b_qkv, seed_l, 0),
                          side(pw, pw->a_qkv, nullptr, nullptr, seed_p, 0), bc->qkv, u, ldu, ws_v, ws->dxn, accumulate, M, rowmask, st));
    CHECK(norm_bwd(cfg->resid_f32, ws->dxn, x_in, w->ln1, a->rstd1, ws->dx_mid, dx_in, nullptr, 0, ws->norm_ws, M, H, st));
    return VLR_OK;
}

// CLIP encoder layer (transformers CLIPEncoderLayer, pre-LN, quick_gelu; call site Llava/__init__.py:178), in place on x
extern "C" int vlr_vit_layer_fwd(const vlr_vit_cfg* cfg, const vlr_vit_layer_weights* w, const vlr_vit_ws* ws,
                                 void* x, int n_img, int T, vlr_stream_t st) {
    VLR_REQUIRE(cfg && w && ws && x, "vlr_vit_layer_fwd: null argument");
    const int D = cfg->hidden, F = cfg->mlp, M = n_img * T;
    VLR_REQUIRE(cfg->heads * cfg->head_dim == D, "vlr_vit_layer_fwd: heads*head_dim != hidden");
    const int hdp = cfg->head_dim_pad > cfg->head_dim ? cfg->head_dim_pad : cfg->head_dim;   // per-head width of the q|k|v / attention buffers
    const int A = cfg->heads * hdp;
    const float scale = cfg->attn_scale > 0.f ? cfg->attn_scale : 1.0f / sqrtf((float)cfg->head_dim);
    const int act = cfg->act == 2 ? 2 /*gelu*/ : 1 /*quick_gelu*/;
    CHECK(vlr_layernorm_fwd(x, w->ln1_w, w->ln1_b, ws->xn, M, D, cfg->ln_eps, st));
    CHECK(vlr_gemm_bf16(0, ws->xn, w->wqkv, ws->qkv, w->bqkv, nullptr, M, 3 * A, D, D, D, 3 * A, 0, 0, 0, 0, st));
    CHECK(vlr_attn_fwd(ws->qkv, off(ws->qkv, A), off(ws->qkv, 2 * (size_t)A), 3 * A, ws->attn, A, nullptr, nullptr, n_img, T,
                       cfg->heads, hdp, 0, scale, st));
    CHECK(vlr_gemm_bf16(0, ws->attn, w->wo, x, w->bo, x, M, D, A, A, A, D, D, 0, 0, 0, st));
    CHECK(vlr_layernorm_fwd(x, w->ln2_w, w->ln2_b, ws->xn, M, D, cfg->ln_eps, st));
    CHECK(vlr_gemm_bf16(0, ws->xn, w->w1, ws->h, w->b1, nullptr, M, F, D, D, D, F, 0, act, 0, 0, st));
    CHECK(vlr_gemm_bf16(0, ws->h, w->w2, x, w->b2, x, M, D, F, F, F, D, D, 0, 0, 0, st));
    return VLR_OK;
}
