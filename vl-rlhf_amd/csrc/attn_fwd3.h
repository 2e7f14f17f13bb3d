// Forward attention, D = 128, 64 queries per wave (included by attention.hip after the fwd2 kernel: shares its tile layout, DMA and mask
// helpers).  Replaces flash-attn 2.5.8 at /root/reference/src/vlrlhf/utils/auto_load.py:49-56,534 for the decoder's forward passes.
//
// Why: with 32 queries per wave every v_mfma_f32_32x32x16_bf16 needs one fresh 1-KiB fragment from LDS, and the matrix pipe, the LDS array
// and the softmax VALU of attn_fwd2_kernel each sit well below half (round 6 counters: 42 / 23 / 55 % busy - no unit is the bound, the
// phases of the two waves of a SIMD do not overlap; the "128 B/clk = LDS-bound" this header used to quote was wrong by a factor of two).
// Here a wave owns TWO 32-query halves: every K / V fragment feeds two MFMAs (half the LDS bytes per FLOP).  That costs ~420 registers,
// i.e. ONE wave per SIMD, so nothing but the wave's own instruction order can put the softmax beside the matrix work.  A first version
// that left the order to hipcc ran at 0.7x of fwd2 (DESIGN.md "tried and reverted"); this one spells the issue order out:
//   * software pipeline ACROSS KV tiles: iteration t runs S(t+1) = K(t+1) Q^T (32 MFMAs) beside the first half of softmax(t), then
//     O += V(t) P(t) (32 MFMAs) beside the second half of softmax(t) and the row maxima of S(t+1);
//   * the tile body is 32 fenced SLOTS (__builtin_amdgcn_sched_barrier): one LDS fragment -> two MFMAs (64 matrix-pipe cycles), the
//     fragment of three slots ahead is requested, and one "pair" of softmax work rides along: v_pk_fma (scale, -max), 2 x v_exp,
//     v_pk_add (row sum), v_cvt_pk_bf16 - 44 VALU cycles against the 56 a slot leaves free.  Pairs are ordered by the deadline of the
//     P fragment they belong to;
//   * lazy rescale: the running maximum a query's exponentials are taken against only moves when the new maximum exceeds it by more
//     than 2^F3_TAU (P <= 256, exact in bf16 as a power-of-two scaling; lse = m + log2 l is invariant), so the 128-register rescale of O
//     - which has to pass through VGPRs - leaves the steady state;
//   * 128-thread workgroups (2 waves), two per CU: the barrier per tile couples two waves instead of four, the AttnGrid block map
//     (128 queries per block) and the persistent ticket counters of fwd2 apply unchanged.
// LDS: [K stage 0 | V stage 0 | K stage 1 | V stage 1 | tile masks]; at iteration t K(t+1) and V(t) are read, K(t+2) and V(t+1) are
// fetched by LDS-DMA into the stages iteration t-1 read.
#pragma once

#define F3_TAU 8.0f
#ifndef F3_NW
#define F3_NW 4          // waves per workgroup: 4 = 256 queries per workgroup, one per CU; 2 = 128 queries, two per CU
#endif
#ifndef F3_ABLATE
#define F3_ABLATE 0      // timing experiments (wrong results): 1 no softmax pairs, 2 no MFMAs, 4 no fragment reads, 8 no per-tile barrier
#endif
#define F3_PF 3            // fragments requested this many slots ahead of their MFMAs

#ifdef F3_TRACE
// timing probe (diagnostics build only): wave 0 of workgroup 0 stamps s_memtime at fixed points of its first tiles
__device__ unsigned long long f3_trace_buf[4096];
#define F3_STAMP(slot) do { if (f3_tr_on && f3_tr_n < 4000) { f3_trace_buf[f3_tr_n++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); } } while (0)
#else
#define F3_STAMP(slot) do {} while (0)
#endif

struct AttnFwd3 {
    static constexpr int TILE_BYTES = KV_TILE * 128 * 2;
    static constexpr int LDS_BYTES = 4 * TILE_BYTES + ATTN_MAX_TILES * 8 + 16;      // K / V stages, tile masks, one any-mask word per wave
};

// key-validity words of the first nkv tiles (bit = key is padded or past S), NW waves per workgroup: wave w takes tiles w, w + NW, ...,
// four tiles' loads in flight per ballot (attn_tile_masks of attention.hip for any NW).  Returns whether ANY of this wave's words is
// non-zero: the workgroup ORs that (one LDS word per wave) and the tile loop only reads mask words when there is a masked key at all -
// the per-tile ds_read_b64 + wait cost 120 cycles in front of every tile (tools/attn_fwd3_trace.py).
template <int NW>
__device__ __forceinline__ bool f3_tile_masks(unsigned long long* tilemask, const int* __restrict__ kmask, size_t tok0, int S, int nkv,
                                              int wave, int lane) {
    unsigned long long any = 0ull;
    for (int base = wave; base < nkv; base += 4 * NW) {
        int ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = (base + NW * u) * KV_TILE + lane;
            ok[u] = key < S ? (kmask ? kmask[tok0 + key] : 1) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long bad = __builtin_amdgcn_ballot_w64(ok[u] == 0);
            if (base + NW * u < nkv) {
                any |= bad;
                if (lane == 0) tilemask[base + NW * u] = bad;
            }
        }
    }
    return any != 0ull;
}

// LDS fragment reads from explicit per-lane byte addresses: the lane part (row, swizzled chunk) is computed ONCE per workgroup - 8
// addresses for the K row fragments (one per 16-wide d step) and 8 for the transposing V reads (4 d blocks x {rows 0-7, rows 8-15}) -
// and the key block / 16-key step / stage of a fragment is an immediate offset (the swizzle does not depend on them).  Left to
// tile_off() inside the four inlined tile bodies hipcc hoisted a separate address set per body and spilled.
typedef __attribute__((address_space(3))) const bf16x8 f3_lds_bf16x8_t;
struct F3Addr {
    uint32_t k[8];        // K stage 0, key block 0: row lane & 31, chunk 2 st + (lane >> 5)
    uint32_t v[4][2];     // V stage 0, 16-key step 0: d block db, low / high row group
};
__device__ __forceinline__ bf16x8 f3_kfrag(const F3Addr& a, int st, int kb, int stage) {
    return *(f3_lds_bf16x8_t*)(uintptr_t)(a.k[st] + (uint32_t)(kb * 32 * 256 + stage * 2 * AttnFwd3::TILE_BYTES));
}
__device__ __forceinline__ bf16x8 f3_vfrag(const F3Addr& a, int db, int ks, int stage) {
    const uint32_t off = (uint32_t)(ks * 16 * 256 + stage * 2 * AttnFwd3::TILE_BYTES);
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(uintptr_t)(a.v[db][0] + off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(uintptr_t)(a.v[db][1] + off));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ float f3_pair_max(float x) {
    // max over the lane pair (l, l ^ 32) without the LDS crossbar (ds_bpermute + lgkmcnt(0): 140 cycles per query half, measured):
    // v_permlane32_swap exchanges lanes 32..63 of its first register with lanes 0..31 of its second; with the row's value in both,
    // one register then holds the lower lane's value in every lane and the other the upper lane's.  By inline asm: through
    // __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) folds fmaxf(r[0], r[1]) of a swap whose operands carry the same value to r[0]
    // - the swap is emitted, the max is not (tools/probe_permlane32_swap.hip) - and every lane is left with the LOWER lane's value.
    // s_nop 1: the two wait states between a VALU write of an operand and the swap; hipcc pads the reader behind the statement.
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// scores of one 64-key tile for the wave's two query halves: keys that are padded / past S (mk) or in the query's future -> -inf.
// Lane group g holds keys kb*32 + e(r) + 4g, e(r) = (r&3) + 8 (r>>2): the 64-bit word is cut to this lane's 32 keys of a key block and
// shifted by 4g once, the bit of register r is then an immediate; the causal test compares e(r) with a per-lane threshold.
template <bool CAUSAL>
__device__ __forceinline__ void f3_mask(f32x16 (&S)[2][2], unsigned long long mk, int k0, int qi0, int g) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const uint32_t m32 = (uint32_t)(mk >> (32 * kb)) >> (4 * g);
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
            const int thr = qi0 + 32 * qh - k0 - 32 * kb - 4 * g;       // key e is in the future iff e > thr
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = (r & 3) + 8 * (r >> 2);
                if (((m32 >> e) & 1u) || (CAUSAL && e > thr)) S[qh][kb][r] = -INFINITY;
            }
        }
    }
}

// write_rows_staged (attention.hip) with a LOWER row bound as well: rows [rlo, rhi) of the wave's 32 x 128 tile are stored (the ragged
// first block of a sequence starts in front of row 0)
__device__ __forceinline__ void f3_write_rows(const f32x16* acc, float mul, char* stage, bf16_t* __restrict__ dst0, size_t ld, int rlo,
                                              int rhi) {
    constexpr int CPR = 16, RB = 256, RPI = 4;
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int row = lane & 31, g = lane >> 5;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            u32x2 w;
            w[0] = pack_bf16(acc[db][4 * rq] * mul, acc[db][4 * rq + 1] * mul);
            w[1] = pack_bf16(acc[db][4 * rq + 2] * mul, acc[db][4 * rq + 3] * mul);
            *reinterpret_cast<u32x2*>(stage + row * RB + (((db * 4 + rq) ^ (row & (CPR - 1))) << 4) + g * 8) = w;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes (in order per wave) before its reads
    const int c = lane % CPR, r0 = lane / CPR;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int r = i * RPI + r0;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * RB + ((c ^ (r & (CPR - 1))) << 4));
        if (r >= rlo && r < rhi) *reinterpret_cast<u32x4*>(dst0 + (ptrdiff_t)r * (ptrdiff_t)ld + c * 8) = v;
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void f3_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        f3_for<I + 1, N>(f);
    }
}

#include "attn_fwd3_regs.h"

// hipcc pads no hazard of an asm statement (cdna_hip_programming.md 5.7).  Every reader of an MFMA result in this file is either the next
// MFMA of the same accumulate chain (0 states) or at least one slot (>= 2 MFMAs) later; where that does not hold by construction a
// settle statement carries the 12 wait states of an 8-pass MFMA.
__device__ __forceinline__ void f3_settle_s(f32x16 (&S)[2][2]) {
    asm volatile("s_nop 11" : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[1][0]), "+v"(S[1][1]));
}
// v_max3_f32 by hand: fmaxf on values hipcc did not produce itself (the S tiles are asm outputs) draws a canonicalising v_max in front.
// hipcc pads one wait state between an asm statement and an instruction that reads its result straight away, so the row maxima run as
// TWO interleaved chains per query half.
__device__ __forceinline__ float f3_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ void f3_rowmax16(const f32x16& a, float (&m)[2]) {
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
        m[0] = f3_max3(a[r], a[r + 1], m[0]);
        m[1] = f3_max3(a[r + 2], a[r + 3], m[1]);
    }
}
__device__ __forceinline__ float f3_rowmax32(const f32x16& a, const f32x16& b) {
    float m[2] = {-INFINITY, -INFINITY};
    f3_rowmax16(a, m);
    f3_rowmax16(b, m);
    return fmaxf(m[0], m[1]);
}

// one pair of softmax work: two scores of (query half qh, 16-key step ks) -> two probabilities -> row sum, one packed bf16 word.
// Pair j: ks = j / 8 (the P fragment it belongs to, i.e. its deadline), qh = (j / 4) & 1, word i = j & 3.
template <int J>
__device__ __forceinline__ void f3_pair(const f32x16 (&Sc)[2][2], float c, const float (&negm)[2], float (&l)[2][2], u32x4 (&pw)[2][4]) {
    constexpr int ks = J >> 3, qh = (J >> 2) & 1, i = J & 3, kb = ks >> 1, r = 8 * (ks & 1) + 2 * i;
    if constexpr (F3_ABLATE & 1) { pw[qh][ks][i] = 0x3f803f80u; return; }
    // ONE asm statement, seven VALU instructions in this order:  fma fma exp exp add add cvt_pk.
    //  * scalar on purpose: v_pk_fma_f32 / v_pk_add_f32 beside MFMAs cost +22..26 cycles per gap (MI355X_MICROARCH.md), and left to
    //    itself hipcc packs the two query halves' row sums into v_pk_add_f32 chains that it sinks behind the last MFMA of the tile
    //    (64 live exponentials);
    //  * gfx950 needs one wait state between a transcendental (v_exp_f32) and a VALU instruction that reads its result; hipcc pads
    //    that for its own instructions (an s_nop per pair in front of v_cvt_pk) but not for an asm consumer - an asm v_add_f32 placed
    //    right behind the v_exp_f32 of its operand read a stale register (rows summed to -inf).  In this order every consumer has
    //    another instruction between itself and the v_exp_f32 it depends on, so no pad is needed at all;
    //  * two row-sum chains per query half (even / odd element).
    float x0, x1;
    uint32_t w;
    asm("v_fma_f32 %0, %5, %7, %8\n\tv_fma_f32 %1, %6, %7, %8\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %0\n\t"
        "v_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %4, %0, %1"
        : "=&v"(x0), "=&v"(x1), "+v"(l[qh][0]), "+v"(l[qh][1]), "=v"(w)
        : "v"(Sc[qh][kb][r]), "v"(Sc[qh][kb][r + 1]), "s"(c), "v"(negm[qh]));
    pw[qh][ks][i] = w;
}
// The same pair in two statements, for the steady-state slots: A (fma fma exp) goes behind the slot's FIRST MFMA, B (exp add add cvt_pk)
// behind the second.  One wave per SIMD issues in order: two MFMAs back to back leave the wave stalled at the second one until the
// matrix pipe is free (~28 cycles in which nothing issues), and the whole pair (~48 cycles) then runs behind a 32-cycle MFMA - 101-112
// cycles per slot measured (tools/attn_fwd3_trace.py) against 64 of matrix work.  With half a pair behind each MFMA the next MFMA reaches
// the head of the queue about when the pipe frees.  volatile: hipcc must not regroup them around the (volatile) MFMA statements.
template <int J>
__device__ __forceinline__ void f3_pair_a(const f32x16 (&Sc)[2][2], float c, const float (&negm)[2], float& x0, float& x1) {
    constexpr int ks = J >> 3, qh = (J >> 2) & 1, i = J & 3, kb = ks >> 1, r = 8 * (ks & 1) + 2 * i;
    if constexpr (F3_ABLATE & 1) return;
    asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %5\n\tv_exp_f32 %0, %0"
                 : "=&v"(x0), "=&v"(x1)
                 : "v"(Sc[qh][kb][r]), "v"(Sc[qh][kb][r + 1]), "s"(c), "v"(negm[qh]));
}
template <int J>
__device__ __forceinline__ void f3_pair_b(float (&l)[2][2], u32x4 (&pw)[2][4], float& x0, float& x1) {
    constexpr int ks = J >> 3, qh = (J >> 2) & 1, i = J & 3;
    if constexpr (F3_ABLATE & 1) { pw[qh][ks][i] = 0x3f803f80u; return; }
    uint32_t w;
    asm volatile("v_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %4, %0, %1"
                 : "+v"(x0), "+v"(x1), "+v"(l[qh][0]), "+v"(l[qh][1]), "=v"(w));
    pw[qh][ks][i] = w;
}

// The running maxima: a query's exponentials are taken against m_used, which only moves when the tile's maximum is more than 2^TAU above
// it (lazy rescale).  DECIDED beside the last MFMAs of the previous tile (f3_decide: pure VALU, slots 14 / 15 of phase 2) and APPLIED in
// front of the next tile's first pair (f3_rescale: a wave-uniform branch that is not taken in the steady state) - computed at the top
// of the tile these ~25 instructions ran with the matrix pipe idle (236 cycles per tile, tools/attn_fwd3_trace.py).
struct F3Next {
    float negm[2];      // -m for the exponent of the tile to come
    float alpha[2];     // factor for O and l where the maximum moved, 1 elsewhere
    bool any;           // some lane of this wave moved
};
__device__ __forceinline__ void f3_decide(float (&m_used)[2], const float (&mx)[2], float c, F3Next& nx) {
    nx.any = false;
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        const float ms = mx[qh] * c;
        const bool grow = ms > m_used[qh] + F3_TAU;
        const float m_new = grow ? ms : m_used[qh];
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        nx.alpha[qh] = grow ? __builtin_amdgcn_exp2f(m_used[qh] - m_use) : 1.f;
        m_used[qh] = m_new;
        nx.negm[qh] = -m_use;
        nx.any = nx.any || grow;
    }
}
__device__ __forceinline__ void f3_rescale(const F3Next& nx, float (&l)[2][2]) {
    if (__builtin_amdgcn_ballot_w64(nx.any)) {       // rare after the first tiles: O passes through two VGPRs, in place
        l[0][0] *= nx.alpha[0];
        l[0][1] *= nx.alpha[0];
        l[1][0] *= nx.alpha[1];
        l[1][1] *= nx.alpha[1];
        f3_o_scale<0>(nx.alpha[0]);
        f3_o_scale<1>(nx.alpha[1]);
    }
}

// the NP pieces a wave contributes to one K or V tile (see `issue` in attn_fwd3_block)
template <int STAGE_OFF, int I, int NP, int NW>
__device__ __forceinline__ void f3_dma_tile(const bf16_t* tb, const uint32_t (&voff)[NP], uint32_t ldsw) {
    if constexpr (I < NP) {
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[I]), "s"(tb), "s"(ldsw),
                     "i"(STAGE_OFF + NW * I * 1024)
                     : "memory", "scc");
        f3_dma_tile<STAGE_OFF, I + 1, NP, NW>(tb, voff, ldsw);
    }
}

// One KV tile of the pipeline.  Sc: raw scores of tile t (masked), mx: their row maxima (in), those of tile t+1 (out).  S(t+1) -> Sn
// from K(t+1) (stage kst) beside the first half of the softmax; O += V(t) P(t) (stage vst) beside the second half and the row maxima
// of Sn.  diag_next / mk_next (wave-uniform, rare): tile t+1 is the diagonal tile or holds padded keys - masked after the slots,
// maxima taken again.
template <bool CAUSAL, int KST, int VST>
__device__ __forceinline__ void f3_tile(f32x16 (&Sc)[2][2], f32x16 (&Sn)[2][2], float (&m_used)[2], float (&l)[2][2], F3Next& nx,
                                        const F3Addr& fa, float c, int lane, bool diag_next, unsigned long long mk_next, int k0_next,
                                        int qi0, bool f3_tr_on, int& f3_tr_n) {
    u32x4 pw[2][4];
    bf16x8 kf[16], vf[16];
    F3_STAMP(1);
    f3_for<0, F3_PF>([&](auto ic) { constexpr int i = ic; kf[i] = f3_kfrag(fa, i & 7, i >> 3, KST); });
    f3_rescale(nx, l);
    const float negm[2] = {nx.negm[0], nx.negm[1]};
    __builtin_amdgcn_sched_barrier(0);
    F3_STAMP(2);
    // ---- phase 1: S(t+1) = K(t+1) Q^T beside the pairs of the first two P fragments (keys 0..31)
    f3_for<0, 16>([&](auto ic) {
        constexpr int i = ic, kb = i >> 3, st = i & 7, n = i + F3_PF;
        if constexpr (F3_ABLATE & 4) {
            if constexpr (n < 16) kf[n] = kf[0];
            else vf[n - 16] = kf[0];
        } else {
            if constexpr (n < 16) kf[n] = f3_kfrag(fa, n & 7, n >> 3, KST);
            else vf[n - 16] = f3_vfrag(fa, (n - 16) & 3, (n - 16) >> 2, VST);
        }
        float x0, x1;
        if constexpr (!(F3_ABLATE & 2)) f3_mma_s<st, st == 0>(Sn[0][kb], kf[i]);
        f3_pair_a<i>(Sc, c, negm, x0, x1);
        if constexpr (!(F3_ABLATE & 2)) f3_mma_s<8 + st, st == 0>(Sn[1][kb], kf[i]);
        else if constexpr (st == 0) {
            Sn[0][kb] = Sc[0][kb];
            Sn[1][kb] = Sc[1][kb];
        }
        f3_pair_b<i>(l, pw, x0, x1);
        __builtin_amdgcn_sched_barrier(0);
    });
    F3_STAMP(3);
    // ---- phase 2: O += V(t) P(t) beside the pairs of keys 32..63 (slots 0..11: 16 pairs, slots 0 / 3 / 6 / 9 take two) and the row
    // maxima of S(t+1) (slots 12..15)
    // slots 0..11: the 16 pairs of keys 32..63 (slots 0 / 3 / 6 / 9 take two); then Sn - complete since the end of phase 1 - is masked
    // if tile t+1 needs it (rare branch), slots 12 / 13 take its row maxima (two interleaved v_max3 chains per query half), slot 14
    // the exchange with the partner lane, slot 15 the decision about the running maxima
    float mxa[2][2] = {{-INFINITY, -INFINITY}, {-INFINITY, -INFINITY}};
    float mxp[2];
    auto slot2 = [&](auto jc) {
        constexpr int j = jc, ks = j >> 2, db = j & 3, n = j + F3_PF;
        if constexpr (n < 16) {
            if constexpr (F3_ABLATE & 4) vf[n] = vf[0];
            else vf[n] = f3_vfrag(fa, n & 3, n >> 2, VST);
        }
        constexpr int first = 16 + j + (j + 2) / 3;            // slots 0,3,6,9 carry two pairs: 16,17 | 18 | 19 | 20,21 | ...
        float x0, x1;
        if constexpr (!(F3_ABLATE & 2)) f3_mma_o<db, db == 0>(vf[j], pw[0][ks]);
        if constexpr (j < 12) f3_pair_a<first>(Sc, c, negm, x0, x1);
        if constexpr (j == 12) f3_rowmax16(Sn[0][0], mxa[0]);
        if constexpr (j == 13) f3_rowmax16(Sn[1][0], mxa[1]);
        if constexpr (j == 14) mxp[0] = f3_pair_max(fmaxf(mxa[0][0], mxa[0][1]));
        if constexpr (!(F3_ABLATE & 2)) f3_mma_o<4 + db, db == 0>(vf[j], pw[1][ks]);
        else asm volatile("" ::"v"(vf[j]), "v"(pw[0][ks]), "v"(pw[1][ks]));
        if constexpr (j < 12) {
            f3_pair_b<first>(l, pw, x0, x1);
            if constexpr (j % 3 == 0) f3_pair<first + 1>(Sc, c, negm, l, pw);
        }
        if constexpr (j == 12) f3_rowmax16(Sn[0][1], mxa[0]);
        if constexpr (j == 13) f3_rowmax16(Sn[1][1], mxa[1]);
        if constexpr (j == 14) mxp[1] = f3_pair_max(fmaxf(mxa[1][0], mxa[1][1]));
        if constexpr (j == 15) f3_decide(m_used, mxp, c, nx);
        __builtin_amdgcn_sched_barrier(0);
    };
    f3_for<0, 12>(slot2);
    if (diag_next || mk_next != 0ull) {
        f3_mask<CAUSAL>(Sn, mk_next, k0_next, qi0, lane >> 5);
        __builtin_amdgcn_sched_barrier(0);
    }
    f3_for<12, 16>(slot2);
    F3_STAMP(4);
    F3_STAMP(5);
}

// The wave's LAST tile: no next tile to overlap with - softmax, then O += V P, in program order (once per 128-query block and wave).
template <int VST>
__device__ __forceinline__ void f3_tail(f32x16 (&Sc)[2][2], float (&l)[2][2], const F3Next& nx, const F3Addr& fa, float c) {
    u32x4 pw[2][4];
    f3_rescale(nx, l);
    const float negm[2] = {nx.negm[0], nx.negm[1]};
    f3_for<0, 4>([&](auto kc) {
        constexpr int ks = kc;
        f3_for<0, 8>([&](auto jc) { f3_pair<8 * ks + decltype(jc)::value>(Sc, c, negm, l, pw); });
        f3_for<0, 4>([&](auto dc) {
            constexpr int db = dc;
            const bf16x8 vfr = f3_vfrag(fa, db, ks, VST);
            f3_mma_o<db, true>(vfr, pw[0][ks]);
            f3_mma_o<4 + db, true>(vfr, pw[1][ks]);
        });
    });
}

template <bool CAUSAL>
__device__ __forceinline__ unsigned attn_fwd3_block(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                    int ld, bf16_t* __restrict__ o, int ldo, float* __restrict__ lse,
                                                    const int* __restrict__ kmask, int S, int Sp, float scale_log2, const AttnGrid& ag,
                                                    int L, char* smem, unsigned* ctr) {
    constexpr int D = 128, TB = AttnFwd3::TILE_BYTES;
    unsigned long long* tilemask = reinterpret_cast<unsigned long long*>(smem + 4 * TB);
    const int t = threadIdx.x, lane = t & 63, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int head, kvhead, b, qslot;
    unsigned nxt = 0;
    if (!ag.decode(L, head, kvhead, b, qslot)) return (ctr && t == 0) ? atomicAdd(ctr, 1u) : 0u;
    const int nh = ag.heads;
    const size_t tok0 = (size_t)b * S;
    const bf16_t* qh_ = q + tok0 * ld + head * D;
    const bf16_t* kh = k + tok0 * ld + kvhead * D;
    const bf16_t* vh = v + tok0 * ld + kvhead * D;
    // Query blocks of BR = 64 NW rows are aligned to the END of the sequence: block j covers rows [S - BR (j+1), S - BR j), so the ragged
    // block is the FIRST one (one or two KV tiles) instead of the last (all of them) - with 256-row blocks a ragged last block would keep
    // three idle waves waiting at 25 barriers.  Slot 0 is the heaviest block.  Rows below 0 are clamped on load, masked, never stored.
    constexpr int NW = F3_NW, BR = 64 * NW, NP = 16 / NW;
    const int row0 = S - BR * (qslot + 1);
    const int qw0 = row0 + wave * 64;
    const int qi0 = qw0 + (lane & 31);                       // query of half 0; half 1 is qi0 + 32
    const int nkv = CAUSAL ? (row0 + BR + KV_TILE - 1) / KV_TILE : (S + KV_TILE - 1) / KV_TILE;
    // tiles this wave computes: keys up to its last query (causal); none when all its queries are in front of the sequence
    const int nt_w = qw0 + 63 < 0 ? 0 : (CAUSAL ? min(nkv, (qw0 + 63) / KV_TILE + 1) : nkv);

    // DMA geometry: piece pc = wave + NW*i covers tile rows [4 pc, 4 pc + 4); lane -> (row, LDS chunk position).  The per-lane byte
    // offsets (row inside the tile, swizzled chunk) are computed once; the tile's first row goes into the SCALAR base: a tile's DMA is
    // one 64-bit scalar multiply-add + NP x {s_mov m0, global_load_lds}, no vector arithmetic.  Only the sequence's last tile (rows past
    // S - 1 are clamped) takes the per-piece path.
    const int prow = lane >> 4, ppos = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lvoid_t*)smem;
    uint32_t voff[NP];               // per piece: (4 pc + prow) rows below the tile's first row, the lane's swizzled 16-byte chunk
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int r = (wave + NW * i) * 4 + prow;
        const int swz = (tile_off<D>(r, 0) - r * (D * 2)) >> 4;
        voff[i] = (uint32_t)(((size_t)r * ld + ((ppos ^ swz) * 8)) * 2);
    }
    const uint32_t ldsw = lds0 + (uint32_t)wave * 1024u;       // scalar: this wave's first piece of stage 0 / K
    // STAGE_OFF: byte offset of the destination tile in LDS (a constant at every call site).  One piece = {s_add_u32 m0, ldsw, constant;
    // s_nop 0; global_load_lds_dwordx4}: the per-piece destination is an immediate added straight into m0 (92 cycles per piece went
    // into seven scalar instructions and spilled-SGPR reloads before, tools/attn_fwd3_trace.py)
    auto issue = [&](const bf16_t* base, int it, auto stage_c) {
        constexpr int STAGE_OFF = decltype(stage_c)::value;
        const int k0 = it * KV_TILE;
#ifdef F3_NODMA          // timing experiment (wrong results): only the first two tiles are fetched, the others reuse what is in LDS
        if (it >= 2) return;
#endif
        if (k0 + KV_TILE <= S) {
            const bf16_t* tb = base + (size_t)k0 * ld;          // scalar: the tile's first row
            f3_dma_tile<STAGE_OFF, 0, NP, NW>(tb, voff, ldsw);
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int pc = wave + NW * i;
                const int r = pc * 4 + prow;
                const int swz = (tile_off<D>(r, 0) - r * (D * 2)) >> 4;
                const int cc = ppos ^ swz;
                int row = k0 + r;
                row = row < S ? row : S - 1;
                attn_dma16(base, (uint32_t)(((size_t)row * ld + cc * 8) * 2), lds0 + STAGE_OFF + pc * 1024);
            }
        }
    };
    using KS0 = std::integral_constant<int, 0>;
    using VS0 = std::integral_constant<int, AttnFwd3::TILE_BYTES>;
    using KS1 = std::integral_constant<int, 2 * AttnFwd3::TILE_BYTES>;
    using VS1 = std::integral_constant<int, 3 * AttnFwd3::TILE_BYTES>;
    F3Addr fa;
    {
        const int l31 = lane & 31;
#pragma unroll
        for (int st = 0; st < 8; ++st) fa.k[st] = lds0 + (uint32_t)tile_off<D>(l31, 2 * st + g);
        const int q4 = lane >> 4, pq = lane & 15;
        const int row = 4 * (q4 >> 1) + (pq >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int col = db * 32 + 16 * (q4 & 1) + (pq & 3) * 4;
            const int sub = ((col >> 2) & 1) * 8;
            fa.v[db][0] = lds0 + (uint32_t)(TB + tile_off<D>(row, col >> 3) + sub);
            fa.v[db][1] = lds0 + (uint32_t)(TB + tile_off<D>(row + 8, col >> 3) + sub);
        }
    }
    issue(kh, 0, KS0{});

    uint32_t* wflag = reinterpret_cast<uint32_t*>(smem + 4 * TB + ATTN_MAX_TILES * 8);
    {
        const bool wany = f3_tile_masks<NW>(tilemask, kmask, tok0, S, nkv, wave, lane);
        if (lane == 0) wflag[wave] = wany ? 1u : 0u;
    }

    // the wave's Q fragments -> a[128:191] (B operands of the S MFMAs), O = a[0:127] = 0
    f3_for<0, 16>([&](auto ic) {
        constexpr int idx = ic, h = idx >> 3, st = idx & 7;
        int qrow = qi0 + 32 * h;
        qrow = qrow < 0 ? 0 : qrow;            // (rows in front of the sequence: finite data, fully masked, never stored)
        f3_q_put<idx>(*reinterpret_cast<const u32x4*>(qh_ + (size_t)qrow * ld + 16 * st + 8 * g));
    });
    f3_o_zero();
    float m_used[2] = {-INFINITY, -INFINITY};
    float l[2][2] = {{0.f, 0.f}, {0.f, 0.f}};       // this lane's half of the row sums, two chains each (its 32 of every 64 keys); the lane pair is added at the end
    F3Next nx;                      // the running-maximum decision for the tile whose softmax comes next
    nx.any = false;
    f32x16 Sx[2][2][2];             // [tile parity][query half][key block]: raw scores, double buffered across the pipeline
    // K(0) has landed (vmcnt also retires the Q fragments and the mask words' loads); V(0) and K(1) go out under S(0) = K(0) Q^T
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (lgkmcnt: the tile-mask words written above)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue(vh, 0, VS0{});
    if (nkv > 1) issue(kh, 1, KS1{});
    bool has_masks = false;
#pragma unroll
    for (int w = 0; w < NW; ++w) has_masks = has_masks || wflag[w] != 0u;
    has_masks = __builtin_amdgcn_readfirstlane((int)has_masks) != 0;
    auto mask_word = [&](int it) { return has_masks ? tilemask[it] : 0ull; };
    if (nt_w > 0) {
        f3_for<0, 16>([&](auto ic) {
            constexpr int i = ic, kb = i >> 3, st = i & 7;
            const bf16x8 kf = f3_kfrag(fa, st, kb, 0);
            f3_mma_s<st, st == 0>(Sx[0][0][kb], kf);
            f3_mma_s<8 + st, st == 0>(Sx[0][1][kb], kf);
        });
        f3_settle_s(Sx[0]);
        {
            const unsigned long long mk0 = mask_word(0);
            if (mk0 != 0ull || (CAUSAL && KV_TILE - 1 > qw0)) f3_mask<CAUSAL>(Sx[0], mk0, 0, qi0, g);
        }
        float mx[2];
        mx[0] = f3_pair_max(f3_rowmax32(Sx[0][0][0], Sx[0][0][1]));
        mx[1] = f3_pair_max(f3_rowmax32(Sx[0][1][0], Sx[0][1][1]));
        f3_decide(m_used, mx, scale_log2, nx);
    }

#define F3_ITER(P_)                                                                                                              \
    {                                                                                                                            \
        const int it_ = it + (P_);                                                                                               \
        F3_STAMP(6);                                                                                                             \
        /* K(it+1) and V(it) have landed; nobody still reads K(it) / V(it-1) */                                                  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        if (!(F3_ABLATE & 8)) __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("" ::: "memory");                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        F3_STAMP(7);                                                                                                             \
        if (it_ + 2 < nkv) issue(kh, it_ + 2, std::integral_constant<int, (P_)*2 * AttnFwd3::TILE_BYTES>{});                     \
        if (it_ + 1 < nkv) issue(vh, it_ + 1, std::integral_constant<int, (((P_) ^ 1) * 2 + 1) * AttnFwd3::TILE_BYTES>{});       \
        else if (ctr && t == 0) nxt = atomicAdd(ctr, 1u);                                                                        \
        if (it_ + 1 < nt_w) {                                                                                                    \
            const bool dn = CAUSAL && (it_ + 1) * KV_TILE + KV_TILE - 1 > qw0;                                                   \
            F3_STAMP(8);                                                                                                         \
            f3_tile<CAUSAL, (P_) ^ 1, P_>(Sx[P_], Sx[(P_) ^ 1], m_used, l, nx, fa, scale_log2, lane, dn, mask_word(it_ + 1),      \
                                          (it_ + 1) * KV_TILE, qi0, f3_tr_on, f3_tr_n);                                          \
        } else if (it_ < nt_w) {                                                                                                 \
            f3_tail<P_>(Sx[P_], l, nx, fa, scale_log2);                                                                          \
        }                                                                                                                        \
    }
    const bool f3_tr_on = (L == 0) && wave == 0;          // (diagnostics builds: the first workgroup's wave 0; item 0 = the heaviest block)
    int f3_tr_n = 0;
    F3_STAMP(9);
    for (int it = 0; it < nkv; it += 2) {
        F3_ITER(0)
        if (it + 1 < nkv) F3_ITER(1)
    }
#undef F3_ITER
    // epilogue: O = acc / l through a wave-private LDS patch (whole rows out), lse in the log2 domain
    __builtin_amdgcn_s_barrier();            // every wave is done with the K / V stages
    asm volatile("" ::: "memory");
    f3_o_settle();
    // lse rows are padded to Sp (multiple of 64) and the tail holds +inf so that the backward's P is exactly 0 there: written by the
    // block that ends at row S
    if (lse && qslot == 0 && wave == 0 && S + lane < Sp) lse[((size_t)b * nh + head) * Sp + S + lane] = INFINITY;
    float inv[2], lsum[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        lsum[h] = l[h][0] + l[h][1];
        lsum[h] += __shfl_xor(lsum[h], 32);
        inv[h] = lsum[h] > 0.f ? 1.f / lsum[h] : 0.f;
        const int qi = qi0 + 32 * h;
        if (lse && g == 0 && qi >= 0) lse[((size_t)b * nh + head) * Sp + qi] = lsum[h] > 0.f ? m_used[h] + log2f(lsum[h]) : INFINITY;
    }
    if (nt_w > 0) {
        {
            const f32x16 a4[4] = {f3_o_get<0>(), f3_o_get<1>(), f3_o_get<2>(), f3_o_get<3>()};
            f3_write_rows(a4, inv[0], smem + wave * TB, o + ((ptrdiff_t)tok0 + qw0) * (ptrdiff_t)ldo + head * D, ldo, -qw0, 32);
        }
        {
            const f32x16 a4[4] = {f3_o_get<4>(), f3_o_get<5>(), f3_o_get<6>(), f3_o_get<7>()};
            f3_write_rows(a4, inv[1], smem + wave * TB + 32 * D * 2, o + ((ptrdiff_t)tok0 + qw0 + 32) * (ptrdiff_t)ldo + head * D, ldo, -(qw0 + 32), 32);
        }
    }
    return nxt;
}

template <bool CAUSAL>
__global__ __launch_bounds__(64 * F3_NW, 1) void attn_fwd3_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, int ld, bf16_t* __restrict__ o, int ldo,
                                                           float* __restrict__ lse, const int* __restrict__ kmask, int S, int Sp,
                                                           float scale_log2, AttnGrid ag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!ag.ctr) {
        attn_fwd3_block<CAUSAL>(q, k, v, ld, o, ldo, lse, kmask, S, Sp, scale_log2, ag, blockIdx.x, smem, nullptr);
        return;
    }
    __shared__ int s_item;
    const int xcd = blockIdx.x & 7;
    unsigned* ctr = ag.ctr + xcd * 32;
    if (threadIdx.x == 0) s_item = (int)atomicAdd(ctr, 1u);
    __syncthreads();
    int item = __builtin_amdgcn_readfirstlane(s_item);
    while (item < ag.items) {
        const unsigned nxt = attn_fwd3_block<CAUSAL>(q, k, v, ld, o, ldo, lse, kmask, S, Sp, scale_log2, ag, item * 8 + xcd, smem, ctr);
        __syncthreads();                 // every wave is done with the stages, the masks and s_item
        if (threadIdx.x == 0) s_item = (int)nxt;
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_item);
    }
    if (threadIdx.x == 0 && atomicAdd(ctr + 1, 1u) == (unsigned)(gridDim.x / 8 - 1)) {
        ctr[0] = 0;
        ctr[1] = 0;
    }
}
