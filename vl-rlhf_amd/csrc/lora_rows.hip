// Row-slab adapter products of LoRA / PLoRA without split-K partials (round 6; opt-in for the layer passes, see below):
//   mode U (forward) : u [M][t * ostride .. + r] = alpha * (keep_t . x [M][in]) . A_t [r][in]^T      (peft lora_A(dropout(x)) * scaling)
//   mode V (backward): v [M][t * r .. + r]       = dy [M][ofs_t .. + out_t] . B_t [out_t][r]         (the input of the dA / dx terms)
// Both contract a LONG dimension (in / out = 4096 .. 14336) of a [M][K] activation that is read once against a SMALL matrix (r = 64 ..
// 256 columns, L2 resident): 2 M K r flops on 2 M K bytes.  The 128x128 ring kernel (gemm128p.hip) runs them as split-K tile GEMMs with
// fp32 partials and a reduction launch at 2.4 - 6 x the HBM floor.  Here a workgroup owns 64 ROWS (times one target) and walks the whole
// K range in steps of 64: no partials, no reduction launch, a fixed summation order (bit-reproducible), PLoRA's row set handled in the
// kernel (slabs without a marked row write zeros and leave; unmarked rows are zeroed on the way out - no vlr_rows_mask launch).
//   * a ring of LR_NS LDS stages, each {activation tile [64 rows][64 k] | packed keep masks [64 rows][8 B] | the small operand's 64 k
//     slice ([r][64 k] of A_t, K-contiguous; or [64 k][r] of B_t, K-strided, read back by ds_read_b64_tr_b16)}, filled by
//     global_load_lds_dwordx4 LR_NS - 1 steps ahead with ONE counted vmcnt per step; the 24 one-KiB pieces of a step are dealt to EIGHT
//     waves, four of which (2 row halves x 2 k halves) also do the MFMAs;
//   * the fragment reads of step tau + 1 are issued in front of the MFMAs of step tau (an LDS read stream of 40 KiB per step is as long
//     as the step's MFMAs); lora_dropout is applied to the activation fragments from the stage's keep bytes; the two k halves are summed
//     through LDS at the end, scaled and written once as bf16.  r = 256 runs as two column halves of 128.
// WHAT WAS LEARNT (profiles/r06_lora_rows_ablation.txt; five designs measured on the same shapes):
//   1. fragment-shaped global loads of the activation straight into MFMA registers (16 rows x 16 B per quarter wave) cost the vector L1 a
//      128-byte line access per 16 useful bytes: no faster than the tile kernel although nothing but the small operand touched LDS;
//   2. LDS-DMA and VGPR loads of ONE wave do not retire in order relative to each other: a counted vmcnt over a mixed queue hands over
//      garbage (every parity test failed) - streams of different kinds need different waves, or one kind;
//   3. LDS-DMA is bound per CU at ~40 B/clk whoever issues it (1, 4 or 8 waves: 24 KiB per ~600 cycles with every source line in L2 and no
//      MFMA at all): a 64-row slab moves 3 bytes through it per activation byte (its own + 2 of the small operand), the 128-row split-K
//      tile kernel 2 + partials - which is why both land at 3.3 - 3.5 TB/s of activation and why this kernel measures 0.3 - 0.4 % SLOWER
//      in the step (484.5 - 485.1 against 482.8 - 483.0 ms, LLaVA-1.5-7B LoRA r 128, same box).  The lever is bytes of the small operand per
//      activation byte (taller slabs + a K split across CUs), not pipeline depth.
// The layer passes therefore take it where it wins (a row-restricted adapter, rank <= 64: lr_mode below) and keep the tile GEMMs elsewhere.
#include <stdlib.h>

#include <type_traits>

#include "../../include/vlr.h"
#include "gemm.h"

#define LR_NS 6                 // LDS stages: LR_NS - 1 steps in flight
#define LR_MAXT 8
// a stage = activation tile [ROWS rows][64 k] | packed keep masks [ROWS rows][8 B] | the small operand's 64 k slice

typedef __attribute__((address_space(3))) void lr_lvoid_t;
typedef __attribute__((ext_vector_type(4))) short lr_s16x4_t;
typedef __attribute__((address_space(3))) lr_s16x4_t lr_lds_s16x4_t;

struct LoraRowsParams {
    const bf16_t* X; int ldx;            // streamed activation [M][ldx]
    int M, n, ct;                        // n (pseudo-)targets, ct = columns / 16 per target (4 or 8)
    int xofs[LR_MAXT];                   // first column of target t's K range in X
    int K[LR_MAXT];                      // contraction length, multiple of 64
    const bf16_t* W[LR_MAXT];            // mode U: A_t [16 ct][ldw]; mode V: B_t [K][ldw] (its first 16 ct columns)
    int ldw;
    bf16_t* out; int ldo;
    int ocol[LR_MAXT];                   // first output column of target t
    float alpha;
    const unsigned char* bits[LR_MAXT];  // packed keep masks over the dense [M][bits_ld] activation, or null
    int bits_ld;
    const unsigned char* rowmask;        // [M] or null
    int dbg;                             // VLR_LORA_ROWS_DBG (timing experiments only: 1 = activation staged once, 2 = small operand staged once, 4 = no MFMA)
};

__device__ __forceinline__ void lr_dma16_s(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void lr_dma4_s(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
#define LR_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LR_BARRIER()                        \
    do {                                    \
        __builtin_amdgcn_sched_barrier(0);  \
        __builtin_amdgcn_s_barrier();       \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)

// K-contiguous image [rows][128 B], 16-byte chunk c of row r at c ^ ((r >> 1) & 7): fragment of 16 rows x 32 k (k half s)
__device__ __forceinline__ bf16x8 lr_frag_kc(const char* tile, int rbase, int s, int lane) {
    const int row = rbase + (lane & 15);
    const int chunk = s * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
// K-strided image [64 k][256 B], chunk c of k-row r at c ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1)) (gemm128p.hip / gemm256p.hip)
__device__ __forceinline__ bf16x8 lr_frag_ks(const char* tile, int cbase, int s, int lane) {
    const int g = lane >> 4, pq = lane & 15;
    const int krow = s * 32 + g * 8 + (pq >> 2);
    const int col = cbase + (pq & 3) * 4;
    const int swz = ((krow & 3) << 2) | (((krow >> 3) & 1) << 1);
    const int off = krow * 256 + (((col >> 3) ^ swz) << 4) + ((col >> 2) & 1) * 8;
    const lr_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lr_lds_s16x4_t*)(tile + off));
    const lr_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lr_lds_s16x4_t*)(tile + off + 4 * 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// zero the dropped elements of a fragment (bit e of keep = element e)
__device__ __forceinline__ bf16x8 lr_mask8(bf16x8 f, uint32_t keep) {
    u32x4 w = __builtin_bit_cast(u32x4, f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t two = (keep >> (2 * e)) & 3u;
        const uint32_t m = ((two | (two << 15)) & 0x10001u) * 0xffffu;
        w[e] &= m;
    }
    return __builtin_bit_cast(bf16x8, w);
}

// RW = MFMA wave rows: 2 = a workgroup owns 64 rows (waves 0-3 do the MFMAs, waves 4-7 only feed the ring, six stages); 4 = 128 rows (all
// eight waves do MFMAs: the small operand's slice is shared by twice the rows - 2 bytes through LDS-DMA per activation byte instead of 3 -
// four stages)
template <int MODE, int CT, bool MASK, int RW = 2>
__global__ __launch_bounds__(512) void lora_rows_kernel(LoraRowsParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // NS stages of (activation tile | keep bytes | small-operand slice)
    constexpr int ROWS = 32 * RW, XT = ROWS * 128, XS = XT + ROWS * 8;
    constexpr int NS = RW == 2 ? LR_NS : 4;
    constexpr int WS = MODE == 0 ? CT * 16 * 128 : 64 * 256;         // the small operand's slice of a stage (64 k)
    constexpr int NW = WS / 4096;                                     // its 1 KiB pieces per wave of four (4, or 2 at r = 64 in mode U)
    constexpr int SS = XS + WS;                                       // bytes per stage
    constexpr int NMW = 2 * RW;                                       // waves that do MFMAs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (slab, target): the targets of a slab are neighbours in one XCD's dispatch order - the activation's other reads hit that L2
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int t = idx % p.n, slab = (idx / p.n) * 8 + xcd;
    const int m0 = slab * ROWS;
    if (m0 >= p.M) return;
    bf16_t* out = p.out + p.ocol[t];
    if (p.rowmask) {
        bool any = false;
#pragma unroll
        for (int h = 0; h < ROWS / 64; ++h) {
            const int row = m0 + h * 64 + lane;
            any = any || (row < p.M && p.rowmask[row] != 0);
        }
        if (__ballot(any) == 0) {       // no marked row: the slab's block of the output is zero (wave-uniform, the same in every wave)
            for (int i = tid; i < ROWS * CT * 4; i += 512) {
                const int r_ = i / (CT * 4), c4 = i % (CT * 4);
                if (m0 + r_ < p.M) *reinterpret_cast<u32x2*>(out + (size_t)(m0 + r_) * p.ldo + c4 * 4) = u32x2{0u, 0u};
            }
            return;
        }
    }
    const int nt = p.K[t] >> 6;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lr_lvoid_t*)smem;
    const int wr = wave >> 1, kp = wave & 1, lm = lane & 15, q = lane >> 4;

    // ---- the ROWS / 8 + 4 NW one-KiB pieces of a step are dealt to the EIGHT waves (piece g: g < ROWS / 8 = the activation tile's 8-row
    // piece g, else piece g - ROWS / 8 of the small operand's slice; wave w takes g = w, w + 8, ...).  Keep bytes: waves 0-3.
    constexpr int NXP = ROWS / 8;
    constexpr int NP = (NXP + 4 * NW) / 8;                           // pieces per wave
    uint32_t offP[NP], offB;
    uint32_t ldsP[NP];                                                // LDS offset of the piece inside a stage
    bool isx[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int g = wave + 8 * i;
        isx[i] = g < NXP;
        if (g < NXP) {
            const int r_ = g * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r_ >> 1) & 7);
            int row = m0 + r_;
            row = row < p.M ? row : p.M - 1;
            offP[i] = (uint32_t)(((size_t)row * p.ldx + p.xofs[t] + c * 8) * 2);
            ldsP[i] = g * 1024;
        } else {
            const int jj = g - NXP;
            if constexpr (MODE == 0) {
                const int r_ = jj * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r_ >> 1) & 7);
                offP[i] = (uint32_t)(((size_t)r_ * p.ldw + c * 8) * 2);
            } else {
                const int kr = jj * 4 + (lane >> 4);
                int c = (lane & 15) ^ (((kr & 3) << 2) | (((kr >> 3) & 1) << 1));
                if (c >= 2 * CT) c &= 2 * CT - 1;          // r = 64: the image's unused chunk slots re-read a used chunk of the row
                offP[i] = (uint32_t)(((size_t)kr * p.ldw + c * 8) * 2);
            }
            ldsP[i] = XS + jj * 1024;
        }
    }
    {   // keep bytes [ROWS rows][8 B]: waves 0-3, one dword per lane; 64 rows: lanes 0-31 of each (16 rows x 2 halves), 128 rows: all 64 lanes
        constexpr int RPWV = ROWS / 4;
        int row = m0 + (wave & 3) * RPWV + ((lane & (2 * RPWV - 1)) >> 1);
        row = row < p.M ? row : p.M - 1;
        offB = (uint32_t)((size_t)row * (p.bits_ld >> 3) + (lane & 1) * 4);
    }
    const char* xbase = reinterpret_cast<const char*>(p.X);
    const char* bbase = reinterpret_cast<const char*>(p.bits[t]);
    const char* wbase = reinterpret_cast<const char*>(p.W[t]);
    const size_t wstep = MODE == 0 ? (size_t)128 : (size_t)64 * p.ldw * 2;
    // every step issues a later step, clamped to the last one (the tail re-reads it - L2 hits - into a stage nobody reads): the counted
    // wait is one constant
    auto issue = [&](int step) {
        const int src = step < nt ? step : nt - 1;
        const char* xb = xbase + (size_t)((p.dbg & 1) ? 0 : src) * 128;
        const char* wb = wbase + (size_t)((p.dbg & 2) ? 0 : src) * wstep;
        const uint32_t l = lds_base + (step % NS) * SS;
#pragma unroll
        for (int i = 0; i < NP; ++i) lr_dma16_s((8 * i + 7 < NXP || (8 * i < NXP && isx[i])) ? xb : wb, offP[i], l + ldsP[i]);
        if constexpr (MASK) {
            if (wave < 4) {
                const char* kbp = bbase + (size_t)src * 8;
                if (lane < ROWS / 2) lr_dma4_s(kbp, offB, l + XT + wave * (ROWS * 2));
            }
        }
    };
    f32x4 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xa[2][2], wf[2][CT];
    uint32_t kb[2][2];
    auto load = [&](auto sc, int step) {
        constexpr int S = decltype(sc)::value;
        const char* xt = smem + (step % NS) * SS;
        const char* wt = xt + XS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xa[S][i] = lr_frag_kc(xt, wr * 32 + i * 16, kp, lane);
            if constexpr (MASK) kb[S][i] = reinterpret_cast<const unsigned char*>(xt)[XT + (wr * 32 + i * 16 + lm) * 8 + kp * 4 + q];
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if constexpr (MODE == 0) wf[S][j] = lr_frag_kc(wt, j * 16, kp, lane);
            else wf[S][j] = lr_frag_ks(wt, j * 16, kp, lane);
        }
    };
    auto mma = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if ((p.dbg & 4) || wave >= NMW) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 x = xa[S][i];
            if constexpr (MASK) x = lr_mask8(x, kb[S][i]);
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[S][j], x, acc[i][j], 0, 0, 0);
        }
    };
    // hand step `step` to the workgroup: this wave's pieces of it have landed (LR_NS - 2 younger steps stay in flight), so have its LDS
    // reads of the step before; one barrier; the stage of the step before is refilled LR_NS - 1 steps ahead; the step's fragments are
    // read into set F - in FRONT of the MFMAs of the step before (an LDS read stream of 40 KiB per step is as long as the step's MFMAs)
    auto hand = [&](auto fc, int step) {
        if (MASK && wave < 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * (NP + 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * NP) : "memory");
        LR_BARRIER();
        issue(step + NS - 1);
        if (wave < NMW) load(fc, step);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    for (int s_ = 0; s_ < NS - 1; ++s_) issue(s_);
    hand(S0{}, 0);
    int tau = 0;
    for (; tau + 2 < nt; tau += 2) {      // branch-free body: two steps
        hand(S1{}, tau + 1); mma(S0{});
        hand(S0{}, tau + 2); mma(S1{});
    }
    if (tau + 1 < nt) { hand(S1{}, tau + 1); mma(S0{}); mma(S1{}); }
    else mma(S0{});
    LR_WAIT(0);      // the tail's re-reads are still in flight: they must have landed before the stages are reused for the exchange below
    // ---- sum the two k halves through LDS: wave (wr, kp) keeps row tile kp and hands the other one to its partner
    __builtin_amdgcn_s_waitcnt(0xc07f);
    LR_BARRIER();
    if (wave >= NMW) {      // the feeding waves (64-row form): one more barrier (the exchange below) and out
        LR_BARRIER();
        return;
    }
    f32x4* ex = reinterpret_cast<f32x4*>(smem);
    f32x4 mine[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        ex[((wr * 2 + kp) * CT + j) * 64 + lane] = kp ? acc[0][j] : acc[1][j];
        mine[j] = kp ? acc[1][j] : acc[0][j];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the LDS writes are done before the barrier
    LR_BARRIER();
    const int row = m0 + wr * 32 + kp * 16 + lm;
    const bool live = row < p.M;
    const bool keeprow = !p.rowmask || (live && p.rowmask[row] != 0);
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const f32x4 o = ex[((wr * 2 + (1 - kp)) * CT + j) * 64 + lane];
        f32x4 s_;
#pragma unroll
        for (int e = 0; e < 4; ++e) s_[e] = keeprow ? p.alpha * (mine[j][e] + o[e]) : 0.f;
        if (live) *reinterpret_cast<u32x2*>(out + (size_t)row * p.ldo + j * 16 + 4 * q) = u32x2{pack_bf16(s_[0], s_[1]), pack_bf16(s_[2], s_[3])};
    }
}

template <int MODE, int CT, bool MASK, int RW>
static void lr_launch(const LoraRowsParams& p, int wgs, hipStream_t st) {
    constexpr int WS = MODE == 0 ? CT * 16 * 128 : 64 * 256;
    constexpr int LDS = (RW == 2 ? LR_NS : 4) * (32 * RW * 136 + WS);      // (>= the RW * 2 * CT * 64 * 16 bytes of the final exchange)
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)lora_rows_kernel<MODE, CT, MASK, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
    hipLaunchKernelGGL((lora_rows_kernel<MODE, CT, MASK, RW>), dim3(wgs), dim3(512), LDS, st, p);
}

// Which adapter products of the layer passes (layers.cpp) take this kernel - VLR_LORA_ROWS: unset = where it measured faster in the step
// (profiles/r06_lora_rows_ablation.txt): a row-restricted adapter (PLoRA: all-text slabs are not read, no vlr_rows_mask launch - InternLM-
// XComposer2 LoRA 845.6 / 840.3 -> 837.2 / 834.5 ms, full fine-tune 1040.9 / 1040.3 -> 1029.7 / 1030.1 ms, same box) and rank <= 64 (the
// small operand's slice is then no larger than the activation tile: 2 bytes through LDS-DMA per activation byte instead of 3); 1 = every
// product it can take (LLaVA r 128: 0.3 - 0.4 % slower than the split-K tile GEMMs); 0 = none.  The C-ABI entry points always run it.
static int lr_mode() {
    static int m = -1;
    if (m < 0) { const char* e = getenv("VLR_LORA_ROWS"); m = !e ? 2 : (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : 2)); }
    return m;
}

// mode 0: out[:, ocol_t ..] = alpha * (keep_t . X) A_t^T with A_t = W + t * r * ldw rows (K = the common `in`), bits + t * gbits (or null)
// mode 1: out[:, t * r ..]  = X[:, ofs_t .. + Ks[t]] . B_t with B_t = W + ofs_t * r (row pitch r)
// false: a shape the kernel does not take (the caller runs the tile GEMMs)
bool vlr_lora_rows_try_launch(int mode, int n, const void* X, int ldx, const int* Ks, const void* W, int ldw, void* out, int ldo, int ostride,
                              int M, int r, float alpha, const void* bits, long gbits, int bits_ld, const unsigned char* rowmask, hipStream_t st, bool force) {
    if (n < 1 || M < 1) return false;
    if (!force) {
        const int m = lr_mode();
        if (m == 0 || (m == 2 && !(rowmask || r <= 64))) return false;
    }
    if (r != 64 && r != 128 && r != 256) return false;
    const int parts = r == 256 ? 2 : 1, rc = r == 256 ? 128 : r;
    if (n * parts > LR_MAXT) return false;
    if (ldx % 8 != 0 || ldw % 8 != 0 || ldo % 4 != 0 || ostride % 4 != 0) return false;
    if (((uintptr_t)X | (uintptr_t)W) & 15 || ((uintptr_t)out & 7)) return false;
    if ((size_t)M * ldx * 2 >= ((size_t)1 << 32)) return false;
    if (bits && (mode != 0 || bits_ld % 64 != 0 || gbits % 4 != 0 || ((uintptr_t)bits & 3))) return false;
    LoraRowsParams p;
    p.X = (const bf16_t*)X; p.ldx = ldx; p.M = M; p.n = n * parts; p.ct = rc / 16; p.ldw = ldw;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("VLR_LORA_ROWS_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    p.out = (bf16_t*)out; p.ldo = ldo; p.alpha = alpha; p.bits_ld = bits ? bits_ld : 0; p.rowmask = rowmask;
    int ofs = 0;
    for (int t = 0; t < n; ++t) {
        if (Ks[t] < 64 || Ks[t] % 64 != 0) return false;
        for (int h = 0; h < parts; ++h) {
            const int i = t * parts + h;
            p.K[i] = Ks[t];
            p.ocol[i] = t * ostride + h * 128;
            if (mode == 0) {
                p.xofs[i] = 0;
                p.W[i] = (const bf16_t*)W + ((size_t)t * r + h * 128) * ldw;
                p.bits[i] = bits ? (const unsigned char*)bits + (size_t)t * gbits : nullptr;
            } else {
                p.xofs[i] = ofs;
                p.W[i] = (const bf16_t*)W + (size_t)ofs * ldw + h * 128;
                p.bits[i] = nullptr;
            }
        }
        if (mode == 1 && ofs % 8 != 0) return false;
        ofs += Ks[t];
    }
    // 128-row workgroups where they measured faster (rank 256 = two column halves of 128 that share the activation's lines; VLR_LORA_ROWS_R=64 | 128 forces one form)
    static int rows_env = -1;
    if (rows_env < 0) { const char* e = getenv("VLR_LORA_ROWS_R"); rows_env = e ? atoi(e) : 0; }
    const bool big = rc == 128 && (rows_env == 128 || (rows_env != 64 && r == 256));
    const int rows = big ? 128 : 64;
    const int slabs = (M + rows - 1) / rows;
    const int wgs = ((slabs + 7) / 8) * 8 * p.n;
    const bool mask = bits != nullptr;
    if (mode == 0) {
        if (rc == 128) {
            if (big) { if (mask) lr_launch<0, 8, true, 4>(p, wgs, st); else lr_launch<0, 8, false, 4>(p, wgs, st); }
            else { if (mask) lr_launch<0, 8, true, 2>(p, wgs, st); else lr_launch<0, 8, false, 2>(p, wgs, st); }
        } else { if (mask) lr_launch<0, 4, true, 2>(p, wgs, st); else lr_launch<0, 4, false, 2>(p, wgs, st); }
    } else {
        if (rc == 128) { if (big) lr_launch<1, 8, false, 4>(p, wgs, st); else lr_launch<1, 8, false, 2>(p, wgs, st); }
        else lr_launch<1, 4, false, 2>(p, wgs, st);
    }
    return true;
}

// ---- C ABI (include/vlr.h, ABI v9)
extern "C" int vlr_lora_rows_u(int n, const void* x, int ldx, const void* A, void* u, int ldu, int ustride, int M, int in, int r, float alpha,
                               const void* bits, long bits_gstride, const unsigned char* rowmask, vlr_stream_t stream) {
    VLR_REQUIRE(x && A && u && n >= 1 && n <= 4 && M > 0 && in > 0, "vlr_lora_rows_u: bad arguments");
    VLR_REQUIRE(!bits || ldx == in, "vlr_lora_rows_u: the packed keep masks are indexed over the dense [M][in] activation (ldx %d, in %d)", ldx, in);
    int Ks[4] = {in, in, in, in};
    VLR_REQUIRE(vlr_lora_rows_try_launch(0, n, x, ldx, Ks, A, in, u, ldu, ustride ? ustride : r, M, r, alpha, bits, bits_gstride, in, rowmask, (hipStream_t)stream, true),
                "vlr_lora_rows_u: shape not taken (r in {64,128,256}, in %% 64 == 0, 16-byte aligned rows; M %d in %d r %d ldx %d ldu %d)", M, in, r, ldx, ldu);
    return vlr_check_launch("vlr_lora_rows_u");
}
extern "C" int vlr_lora_rows_v(int n, const void* dy, int lddy, const int* outs, const void* B, void* v, int ldv, int M, int r,
                               const unsigned char* rowmask, vlr_stream_t stream) {
    VLR_REQUIRE(dy && outs && B && v && n >= 1 && n <= 4 && M > 0, "vlr_lora_rows_v: bad arguments");
    VLR_REQUIRE(vlr_lora_rows_try_launch(1, n, dy, lddy, outs, B, r, v, ldv, r, M, r, 1.f, nullptr, 0, 0, rowmask, (hipStream_t)stream, true),
                "vlr_lora_rows_v: shape not taken (r in {64,128,256}, outs %% 64 == 0, 16-byte aligned rows; M %d r %d lddy %d ldv %d)", M, r, lddy, ldv);
    return vlr_check_launch("vlr_lora_rows_v");
}
