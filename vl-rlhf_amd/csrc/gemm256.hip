// 256x256x64-tile bf16 MFMA GEMM for gfx950: the large-shape path of vlr_gemm_bf16 (same layouts / epilogue as gemm.hip).
//
// Why a second tile: a 128x128 tile needs 2 B of L2->CU traffic per 64 FLOP, i.e. ~39 TB/s at the 2.5 PF MFMA peak - more
// than the ~34 TB/s the eight L2s deliver; 256x256 halves it.  512 threads = 8 waves (2 along M x 4 along N), each wave
// 128x64 = 4x2 v_mfma_f32_32x32x16_bf16 tiles (128 accumulator registers); 6 ds_read_b128 feed 8 MFMAs.
// K-contiguous operands are staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write): the LDS image
// is lane-linear per wave instruction (8 rows x 128 B), so the XOR swizzle is applied to the per-lane SOURCE address and
// undone on the fragment read (same involution on both sides).  K-strided operands (dgrad B, wgrad A and B) keep the
// register path with 4x4 v_perm transposes.  Two LDS stages (128 KiB): the DMA for tile t+1 is issued right after the
// barrier that retires tile t-1 and lands during tile t's 64 MFMAs per wave; the only wait is a vmcnt(0) in front of
// that barrier, ~2000 cycles after issue.  One barrier per K tile.
#include <stdlib.h>

#include <type_traits>

#include "gemm.h"

#define TM 256
#define TN 256
#define TK 64
#define NOSG(a,b,c) __builtin_amdgcn_sched_group_barrier(a,b,c)
#define STAGE_BYTES ((TM + TN) * TK * 2)   // 64 KiB

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

__device__ __forceinline__ int swz256(int row, int chunk) { return row * (TK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4); }

// LDS-DMA of a 256-row x 64-k tile: wave w issues 4 instructions ("pieces"), each 8 rows x 128 B; rows clamped to nrows-1
__device__ __forceinline__ void dma_piece(const bf16_t* __restrict__ P, int ld, int row0, int nrows, int k0, char* lds,
                                          int wave, int lane, int i) {
    const int R = (wave * 4 + i) * 8;
    const int r = R + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int grow = row0 + r;
    grow = grow < nrows ? grow : nrows - 1;
    const bf16_t* g = P + (size_t)grow * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(lds + R * (TK * 2)), 16, 0, 0);
}
__device__ __forceinline__ void dma_kc(const bf16_t* __restrict__ P, int ld, int row0, int nrows, int k0, char* lds,
                                       int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(P, ld, row0, nrows, k0, lds, wave, lane, i);
}
// register path for k-contiguous operands (16-byte loads, ds_write_b128): 256 rows x 64 k, thread -> (chunk t%8, row t/8 + 64 i)
__device__ __forceinline__ void load_kc256(const bf16_t* __restrict__ P, int ld, int row0, int nrows, int k0, int K, int t,
                                           u32x4 (&r)[4]) {
    const int c = t & 7;
    const int k = k0 + c * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + (t >> 3) + 64 * i;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < nrows && k + 8 <= K) v = *reinterpret_cast<const u32x4*>(P + (size_t)row * ld + k);
        r[i] = v;
    }
}
__device__ __forceinline__ void store_kc256(char* lds, int t, const u32x4 (&r)[4]) {
    const int c = t & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(lds + swz256((t >> 3) + 64 * i, c)) = r[i];
}
// register path for k-strided operands: stored [K][ncols]; tile 64 k x 256 cols; thread -> (k block = t/64, col quad = t%64)
__device__ __forceinline__ void load_ks256(const bf16_t* __restrict__ P, int ld, int col0, int ncols, int k0, int K, int t,
                                           u32x2 (&r)[8]) {
    const int kb = t >> 6, nq = t & 63;
    const int col = col0 + nq * 4;
    const bool cok = col + 4 <= ncols;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + kb * 8 + j;
        u32x2 v = {0u, 0u};
        if (cok && k < K) v = *reinterpret_cast<const u32x2*>(P + (size_t)k * ld + col);
        r[j] = v;
    }
}
__device__ __forceinline__ void store_ks256(char* lds, int t, const u32x2 (&r)[8]) {
    const int kb = t >> 6, nq = t & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = nq * 4 + i;
        u32x4 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t lo = r[2 * p][i >> 1], hi = r[2 * p + 1][i >> 1];
            o[p] = (i & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        *reinterpret_cast<u32x4*>(lds + swz256(row, kb)) = o;
    }
}

// ABL (debug, timing only - results are wrong when set): 1 skip the LDS-DMA, 2 skip the fragment ds_reads after the first
// k-step, 4 skip the per-tile vmcnt + barrier.  Selected with VLR_GEMM_ABLATE for the NT layout.
template <bool A_KS, bool B_KS, int ABL = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * STAGE_BYTES = 128 KiB
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        const int q = nwg >> 3, rem = nwg & 7;
        pid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int gid = pid / per_group;
    const int first_m = gid * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int tm = first_m + (pid % per_group) % gsz;
    const int tn = (pid % per_group) / gsz;
    const int m0 = tm * TM, n0 = tn * TN;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x2 ra[8], rb[8];
    u32x4 rka[4], rkb[4];
    constexpr bool REGKC = (ABL & 8) != 0;
    const int nt = (p.K + TK - 1) / TK;

    auto issue = [&](int kt, int buf) {
        char* a = smem + buf * STAGE_BYTES;
        char* b = a + TM * TK * 2;
        const int k0 = kt * TK;
        if constexpr (A_KS) load_ks256(p.A, p.lda, m0, p.M, k0, p.K, t, ra);
        else if constexpr (REGKC) load_kc256(p.A, p.lda, m0, p.M, k0, p.K, t, rka);
        else dma_kc(p.A, p.lda, m0, p.M, k0, a, wave, lane);
        if constexpr (B_KS) load_ks256(p.B, p.ldb, n0, p.N, k0, p.K, t, rb);
        else if constexpr (REGKC) load_kc256(p.B, p.ldb, n0, p.N, k0, p.K, t, rkb);
        else dma_kc(p.B, p.ldb, n0, p.N, k0, b, wave, lane);
    };
    auto commit = [&](int buf) {   // register-staged operands: transposed write into LDS
        char* a = smem + buf * STAGE_BYTES;
        char* b = a + TM * TK * 2;
        if constexpr (A_KS) store_ks256(a, t, ra);
        else if constexpr (REGKC) store_kc256(a, t, rka);
        if constexpr (B_KS) store_ks256(b, t, rb);
        else if constexpr (REGKC) store_kc256(b, t, rkb);
    };

    issue(0, 0);
    commit(0);
    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        if constexpr (!(ABL & 4)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA for tile kt has landed
            __syncthreads();                                    // everyone's has; everyone is done reading stage cur^1
        }
        const char* a = smem + cur * STAGE_BYTES;
        const char* b = a + TM * TK * 2;
        char* na = smem + (cur ^ 1) * STAGE_BYTES;
        char* nb = na + TM * TK * 2;
        // branch-free prefetch: on the last tile the "next" tile is the last one again (lands in the idle stage, unused)
        const int nk0 = (kt + 1 < nt ? kt + 1 : kt) * TK;
        // register-staged (k-strided) operands: their global loads go out first and land under the MFMAs
        if constexpr (A_KS) load_ks256(p.A, p.lda, m0, p.M, nk0, p.K, t, ra);
        else if constexpr (REGKC) load_kc256(p.A, p.lda, m0, p.M, nk0, p.K, t, rka);
        if constexpr (B_KS) load_ks256(p.B, p.ldb, n0, p.N, nk0, p.K, t, rb);
        else if constexpr (REGKC) load_kc256(p.B, p.ldb, n0, p.N, nk0, p.K, t, rkb);
        // fragments are double-buffered in registers: the ds_reads of k-step kk+1 are in flight under the MFMAs of kk;
        // the next tile's LDS-DMA pieces are issued in between the MFMA groups (their issue cost hides in the MFMA shadow)
        bf16x8 fa[2][4], fb[2][2];
        auto ldfrag = [&](int kk, int s) {
            const int c = kk * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(a + swz256(wm * 128 + i * 32 + (lane & 31), c));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(b + swz256(wn * 64 + j * 32 + (lane & 31), c));
        };
        constexpr bool PINNED = A_KS || B_KS;   // measured: NT is fastest with the whole DMA burst right behind the barrier
        if constexpr (!PINNED && !(ABL & 1) && !REGKC) {
            dma_kc(p.A, p.lda, m0, p.M, nk0, na, wave, lane);
            dma_kc(p.B, p.ldb, n0, p.N, nk0, nb, wave, lane);
        }
        auto dma = [&](int q) {          // q in 0..7: A pieces then B pieces of tile kt+1
            if constexpr (!PINNED || REGKC) return;
            if (q < 4) { if constexpr (!A_KS) dma_piece(p.A, p.lda, m0, p.M, nk0, na, wave, lane, q); }
            else { if constexpr (!B_KS) dma_piece(p.B, p.ldb, n0, p.N, nk0, nb, wave, lane, q - 4); }
        };
        ldfrag(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if constexpr (!(ABL & 2)) { if (kk + 1 < 4) ldfrag(kk + 1, (kk + 1) & 1); }
#define MM(i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(ABL & 2) ? 0 : (kk & 1)][i], fb[(ABL & 2) ? 0 : (kk & 1)][j], acc[i][j], 0, 0, 0)
            MM(0, 0); MM(0, 1); MM(1, 0); MM(1, 1);
            dma(kk * 2);
            MM(2, 0); MM(2, 1); MM(3, 0); MM(3, 1);
#undef MM
            dma(kk * 2 + 1);
            // pin the software pipeline (hipcc otherwise re-serialises ds_read -> wait -> MFMA): one LDS read of the NEXT
            // k-step (or one DMA piece) behind every MFMA of this one
            constexpr int NV = ((!A_KS) || (!B_KS)) ? 1 : 0;
            if constexpr (!PINNED) {
            } else if (kk + 1 < 4) {
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    NOSG(0x008, 1, 0);
                    NOSG(0x100, 1, 0);
                }
                NOSG(0x008, 1, 0);
                if (NV) NOSG(0x020, 1, 0);
                NOSG(0x008, 1, 0);
                if (NV) NOSG(0x020, 1, 0);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) NOSG(0x008, 1, 0);
                if (NV) NOSG(0x020, 1, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) NOSG(0x008, 1, 0);
                if (NV) NOSG(0x020, 1, 0);
            }
        }
        commit(cur ^ 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (unused) last prefetch before LDS is reused
    __syncthreads();

    // ---- epilogue through LDS in two passes (rows [0,64) then [64,128) of each wave's 128x64 block): 16 KiB per wave
    float* stage = reinterpret_cast<float*>(smem) + wave * 64 * 64;
    const int gn0 = n0 + wn * 64;
    auto epilogue_pass = [&](auto pass_c) {
        constexpr int pass = decltype(pass_c)::value;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    stage[row * 64 + j * 32 + (lane & 31)] = acc[pass * 2 + ii][j][r];
                }
        __syncthreads();
        const int gm0 = m0 + wm * 128 + pass * 64;
        if (!p.out_f32) {
            bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
            const int cq = (lane & 7) * 8;
            const int gn = gn0 + cq;
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = 0.f;
            if (p.bias && gn + 8 <= p.N) unpack8(*reinterpret_cast<const u32x4*>(p.bias + gn), bv);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3);
                const int gm = gm0 + row;
                if (gm < p.M && gn + 8 <= p.N) {
                    float v[8];
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + cq);
                    const f32x4 s1 = *reinterpret_cast<const f32x4*>(stage + row * 64 + cq + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = s0[e]; v[4 + e] = s1[e]; }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = apply_act(p.alpha * v[e] + bv[e], p.act);
                    if (p.residual) {
                        float rv[8];
                        unpack8(*reinterpret_cast<const u32x4*>(p.residual + (size_t)gm * p.ldr + gn), rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rv[e];
                    }
                    bf16_t* dst = C + (size_t)gm * p.ldc + gn;
                    if (p.accumulate) {
                        float ov[8];
                        unpack8(*reinterpret_cast<const u32x4*>(dst), ov);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += ov[e];
                    }
                    *reinterpret_cast<u32x4*>(dst) = pack8(v);
                }
            }
        } else {
            float* C = reinterpret_cast<float*>(p.C);
            const int cq = (lane & 15) * 4;
            const int gn = gn0 + cq;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && gn + 4 <= p.N) {
                const u32x2 w = *reinterpret_cast<const u32x2*>(p.bias + gn);
                bv[0] = bf16lo(w[0]); bv[1] = bf16hi(w[0]); bv[2] = bf16lo(w[1]); bv[3] = bf16hi(w[1]);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4);
                const int gm = gm0 + row;
                if (gm < p.M && gn + 4 <= p.N) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 64 + cq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(p.alpha * v[e] + bv[e], p.act);
                    if (p.residual) {
                        const u32x2 w = *reinterpret_cast<const u32x2*>(p.residual + (size_t)gm * p.ldr + gn);
                        v[0] += bf16lo(w[0]); v[1] += bf16hi(w[0]); v[2] += bf16lo(w[1]); v[3] += bf16hi(w[1]);
                    }
                    float* dst = C + (size_t)gm * p.ldc + gn;
                    if (p.accumulate) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += o[e];
                    }
                    *reinterpret_cast<f32x4*>(dst) = v;
                }
            }
        }
        __syncthreads();
    };
    epilogue_pass(std::integral_constant<int, 0>{});
    epilogue_pass(std::integral_constant<int, 1>{});
}



// =====================================================================================================================
// Staggered two-group schedule (NT layout).  Ablation of the kernel above (VLR_GEMM_ABLATE) shows the LDS-DMA *issue* is the
// bottleneck: with the DMA removed the same loop runs 1.57 PF instead of 0.96, with the ds_reads removed only 8 % faster,
// with the barrier removed 4 %.  A global_load_lds blocks its (in-order) wave for ~100 cycles, and with all eight waves in
// lock-step both waves of every SIMD are in their DMA phase at the same time, so the matrix pipe idles.
// Here the K tile is 32 deep, four LDS stages (4 x 32 KiB), and every K tile is split in a LOAD phase (4 DMA pieces of
// tile t+2, first fragment reads of tile t) and a MATH phase (16 MFMAs), separated by barriers.  Waves 4-7 (the SIMD
// partners of waves 0-3) run one barrier behind waves 0-3, so on every SIMD one wave is always in MATH while its partner
// is in LOAD.  Only counted waits: vmcnt(4) keeps the newest tile in flight across barriers; raw s_barrier (a
// __syncthreads would drain the DMA queue with vmcnt(0)).
#define SK 32
#define SSTAGE ((TM + TN) * SK * 2)   // 32 KiB
#define NSTAGE 4

__device__ __forceinline__ int swz32(int row, int chunk) { return row * (SK * 2) + ((chunk ^ ((row >> 2) & 3)) << 4); }

// k-contiguous operand: LDS image [256 rows][32 k] (64-byte rows), 16 rows per wave instruction; K tail -> zeros
__device__ __forceinline__ void dma_piece32(const bf16_t* __restrict__ P, int ld, int row0, int nrows, int k0, int K,
                                            const bf16_t* __restrict__ zero16, char* lds, int wave, int lane, int i) {
    const int R = (wave * 2 + i) * 16;
    const int r = R + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    int grow = row0 + r;
    grow = grow < nrows ? grow : nrows - 1;
    const int k = k0 + c * 8;
    const bf16_t* g = (k + 8 <= K) ? P + (size_t)grow * ld + k : zero16;
    __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(lds + R * (SK * 2)), 16, 0, 0);
}
// k-contiguous operand, FULL-LINE variant: the vector-memory pipeline is the bottleneck of this kernel and it pays per
// 128-byte line, so a 64-byte row (K tile of 32) costs as much as a whole line.  The k-contiguous operand is therefore kept
// in two DOUBLE stages of [256 rows][64 k] (128-byte rows = two consecutive K tiles side by side); one DMA instruction
// brings 8 rows x 128 B, and a double tile is fetched every second K tile.  Same LDS footprint as 4 single stages.
__device__ __forceinline__ void dma_piece64(const bf16_t* __restrict__ P, int ld, int row0, int nrows, int k0, int K,
                                            const bf16_t* __restrict__ zero16, char* lds, int wave, int lane, int i) {
    const int R = (wave * 4 + i) * 8;
    const int r = R + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int grow = row0 + r;
    grow = grow < nrows ? grow : nrows - 1;
    const int k = k0 + c * 8;
    const bf16_t* g = (k + 8 <= K) ? P + (size_t)grow * ld + k : zero16;
    __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(lds + R * 128), 16, 0, 0);
}
// k-strided operand (stored [K][ncols]): LDS image [32 k][256 cols] (512-byte rows), 2 k-rows per wave instruction.  The
// 16-byte chunks of row r are XOR-permuted by (r&3)<<2 on the SOURCE side so that the ds_read_b64_tr_b16 of a 32-lane
// half (4 k-rows x 64 B) touches 16 distinct chunks of a 256-byte bank row.  k >= K -> zeros; columns clamped in range.
__device__ __forceinline__ void dma_piece32_ks(const bf16_t* __restrict__ P, int ld, int col0, int ncols, int k0, int K,
                                               const bf16_t* __restrict__ zero16, char* lds, int wave, int lane, int i) {
    const int R = (wave * 2 + i) * 2;
    const int r = R + (lane >> 5);
    const int chunk = (lane & 31) ^ ((r & 3) << 2);
    int col = col0 + chunk * 8;
    col = col + 8 <= ncols ? col : ncols - 8;
    const int k = k0 + r;
    const bf16_t* g = (k < K) ? P + (size_t)k * ld + col : zero16;
    __builtin_amdgcn_global_load_lds((gvoid_t*)g, (lvoid_t*)(lds + R * 512), 16, 0, 0);
}
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
// MFMA operand fragment (32 rows x 16 k) of a k-strided tile: two hardware-transposing LDS reads.  Within a 16-lane group
// lane p supplies the address of (k-row p/4, 4 columns at (p%4)*4) and receives column p of that 4 x 16 block.
__device__ __forceinline__ bf16x8 frag_ks(const char* region, int nbase, int kk, int lane) {
    const int q = lane >> 4, pq = lane & 15;
    const int krow = kk * 16 + 8 * (q >> 1) + (pq >> 2);
    const int col = nbase + 16 * (q & 1) + (pq & 3) * 4;
    const int off = krow * 512 + (((col >> 3) ^ ((krow & 3) << 2)) << 4) + ((col >> 2) & 1) * 8;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(region + off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(region + off + 4 * 512));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

#define RAW_BARRIER()                               \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();               \
        asm volatile("" ::: "memory");              \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)

template <bool A_KS, bool B_KS>
__global__ __launch_bounds__(512) void gemm256s_kernel(GemmParams p, const bf16_t* __restrict__ zero16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NSTAGE * SSTAGE = 128 KiB
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);

    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        const int q = nwg >> 3, rem = nwg & 7;
        pid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int gid = pid / per_group;
    const int first_m = gid * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int tm = first_m + (pid % per_group) % gsz;
    const int tn = (pid % per_group) / gsz;
    const int m0 = tm * TM, n0 = tn * TN;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (p.K + SK - 1) / SK;
    char* const a_base = smem;                       // 64 KiB per operand
    char* const b_base = smem + NSTAGE * TM * SK * 2;
    // byte offset of the LDS region holding K tile kt of an operand
    auto a_region = [&](int kt) { return a_base + (A_KS ? (kt & 3) * (TM * SK * 2) : ((kt >> 1) & 1) * (TM * SK * 4)); };
    auto b_region = [&](int kt) { return b_base + (B_KS ? (kt & 3) * (TN * SK * 2) : ((kt >> 1) & 1) * (TN * SK * 4)); };
    // issue the DMA owed in the LOAD phase of tile kt: k-strided operands fetch tile kt+2 every phase, k-contiguous
    // operands fetch the double tile (kt+2, kt+3) on even kt
    auto dma_for = [&](int kt) {
        const int k0 = (kt + 2) * SK;
        if constexpr (A_KS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) dma_piece32_ks(p.A, p.lda, m0, p.M, k0, p.K, zero16, a_region(kt + 2), wave, lane, i);
        } else if ((kt & 1) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_piece64(p.A, p.lda, m0, p.M, k0, p.K, zero16, a_region(kt + 2), wave, lane, i);
        }
        if constexpr (B_KS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) dma_piece32_ks(p.B, p.ldb, n0, p.N, k0, p.K, zero16, b_region(kt + 2), wave, lane, i);
        } else if ((kt & 1) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_piece64(p.B, p.ldb, n0, p.N, k0, p.K, zero16, b_region(kt + 2), wave, lane, i);
        }
    };
    // prologue: tiles 0 and 1 resident before anyone reads
    dma_for(-2);                                     // K tiles 0 (and 1 for k-contiguous operands)
    dma_for(-1);                                     // K tile 1 of k-strided operands
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER();
    if (grp == 1) RAW_BARRIER();          // waves 4-7 run one phase behind waves 0-3

    bf16x8 fa[2][4], fb[2][2];
    for (int kt = 0; kt < nt; ++kt) {
        const char* a = a_region(kt);
        const char* b = b_region(kt);
        // ---------------- LOAD phase: all 12 fragment reads of tile kt (the MATH phase is register-only), then the DMA
        if (!(p.flags & 16) || kt == 0)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = kk * 2 + (lane >> 5);
            const int c64 = (kt & 1) * 4 + c;        // chunk inside a 128-byte double-stage row
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (A_KS) fa[kk][i] = frag_ks(a, wm * 128 + i * 32, kk, lane);
                else fa[kk][i] = *reinterpret_cast<const bf16x8*>(a + swz256(wm * 128 + i * 32 + (lane & 31), c64));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (B_KS) fb[kk][j] = frag_ks(b, wn * 64 + j * 32, kk, lane);
                else fb[kk][j] = *reinterpret_cast<const bf16x8*>(b + swz256(wn * 64 + j * 32 + (lane & 31), c64));
            }
        }
        // everything issued BEFORE this phase must have landed by the barrier below (it is first read two phases later);
        // what this phase issues stays in flight: counted vmcnt = number of pieces issued here
        if (kt + 2 < nt) {
            if (!(p.flags & 8)) dma_for(kt);
            constexpr int N_KS = (A_KS ? 2 : 0) + (B_KS ? 2 : 0);
            constexpr int N_KC = (A_KS ? 0 : 4) + (B_KS ? 0 : 4);
            if ((kt & 1) == 0) {
                if constexpr (N_KS + N_KC == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr (N_KS + N_KC == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                if constexpr (N_KS == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if constexpr (N_KS == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        RAW_BARRIER();
        // ---------------- MATH phase
        if (!(p.flags & 1)) __builtin_amdgcn_s_setprio(1);
#define MMS(s_, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i], fb[s_][j], acc[i][j], 0, 0, 0)
        MMS(0, 0, 0); MMS(0, 0, 1); MMS(0, 1, 0); MMS(0, 1, 1); MMS(0, 2, 0); MMS(0, 2, 1); MMS(0, 3, 0); MMS(0, 3, 1);
        MMS(1, 0, 0); MMS(1, 0, 1); MMS(1, 1, 0); MMS(1, 1, 1); MMS(1, 2, 0); MMS(1, 2, 1); MMS(1, 3, 0); MMS(1, 3, 1);
#undef MMS
        __builtin_amdgcn_s_setprio(0);
        RAW_BARRIER();
    }
    if (grp == 0) RAW_BARRIER();          // balance the extra barrier of waves 4-7
    __syncthreads();

    float* stage = reinterpret_cast<float*>(smem) + wave * 64 * 64;
    const int gn0 = n0 + wn * 64;
    auto epilogue_pass = [&](auto pass_c) {
        constexpr int pass = decltype(pass_c)::value;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    stage[row * 64 + j * 32 + (lane & 31)] = acc[pass * 2 + ii][j][r];
                }
        __syncthreads();
        const int gm0 = m0 + wm * 128 + pass * 64;
        if (!p.out_f32) {
            bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
            const int cq = (lane & 7) * 8;
            const int gn = gn0 + cq;
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = 0.f;
            if (p.bias && gn + 8 <= p.N) unpack8(*reinterpret_cast<const u32x4*>(p.bias + gn), bv);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3);
                const int gm = gm0 + row;
                if (gm < p.M && gn + 8 <= p.N) {
                    float v[8];
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + cq);
                    const f32x4 s1 = *reinterpret_cast<const f32x4*>(stage + row * 64 + cq + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = s0[e]; v[4 + e] = s1[e]; }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = apply_act(p.alpha * v[e] + bv[e], p.act);
                    if (p.residual) {
                        float rv[8];
                        unpack8(*reinterpret_cast<const u32x4*>(p.residual + (size_t)gm * p.ldr + gn), rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rv[e];
                    }
                    bf16_t* dst = C + (size_t)gm * p.ldc + gn;
                    if (p.accumulate) {
                        float ov[8];
                        unpack8(*reinterpret_cast<const u32x4*>(dst), ov);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += ov[e];
                    }
                    *reinterpret_cast<u32x4*>(dst) = pack8(v);
                }
            }
        } else {
            float* C = reinterpret_cast<float*>(p.C);
            const int cq = (lane & 15) * 4;
            const int gn = gn0 + cq;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && gn + 4 <= p.N) {
                const u32x2 w = *reinterpret_cast<const u32x2*>(p.bias + gn);
                bv[0] = bf16lo(w[0]); bv[1] = bf16hi(w[0]); bv[2] = bf16lo(w[1]); bv[3] = bf16hi(w[1]);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4);
                const int gm = gm0 + row;
                if (gm < p.M && gn + 4 <= p.N) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 64 + cq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(p.alpha * v[e] + bv[e], p.act);
                    if (p.residual) {
                        const u32x2 w = *reinterpret_cast<const u32x2*>(p.residual + (size_t)gm * p.ldr + gn);
                        v[0] += bf16lo(w[0]); v[1] += bf16hi(w[0]); v[2] += bf16lo(w[1]); v[3] += bf16hi(w[1]);
                    }
                    float* dst = C + (size_t)gm * p.ldc + gn;
                    if (p.accumulate) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += o[e];
                    }
                    *reinterpret_cast<f32x4*>(dst) = v;
                }
            }
        }
        __syncthreads();
    };
    epilogue_pass(std::integral_constant<int, 0>{});
    epilogue_pass(std::integral_constant<int, 1>{});
}

static int g_gemm256_mode = -1;   // -1 unset, 0 disabled (VLR_GEMM256=0), 1 enabled

bool vlr_gemm256_try_launch(int layout, const GemmParams& p, hipStream_t stream) {
    if (g_gemm256_mode < 0) {
        const char* e = getenv("VLR_GEMM256");
        g_gemm256_mode = (e && e[0] == '0') ? 0 : 1;
        if (g_gemm256_mode) {
            hipFuncSetAttribute((const void*)gemm256_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
            hipFuncSetAttribute((const void*)gemm256_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
            hipFuncSetAttribute((const void*)gemm256_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        }
    }
    if (!g_gemm256_mode) return false;
    if (vlr_gemm256p_try_launch(layout, p, stream)) return true;
    const int tiles = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
    if (tiles < 192) return false;                       // too few workgroups for 256 CUs: the 128x128 kernel fills better
    // LDS-DMA cannot zero-fill a K tail: K-contiguous operands need K % 64 == 0 (true for every decoder GEMM)
    if (layout != 2 && p.K % TK != 0) return false;
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("VLR_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    if (layout == 0 && abl) {
#define ABL_CASE(n) case n: { hipFuncSetAttribute((const void*)gemm256_kernel<false, false, n>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES); \
        hipLaunchKernelGGL((gemm256_kernel<false, false, n>), dim3(tiles), dim3(512), 2 * STAGE_BYTES, stream, p); return true; }
        switch (abl) { ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(5) ABL_CASE(6) ABL_CASE(7) ABL_CASE(8) default: break; }
#undef ABL_CASE
    }
    static int stagger = -1;
    static bf16_t* zero16 = nullptr;
    if (stagger < 0) {
        const char* e = getenv("VLR_GEMM_STAGGER");
        stagger = e ? atoi(e) : 7;         // bit 0 NT, bit 1 NN, bit 2 TN
        hipFuncSetAttribute((const void*)gemm256s_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * SSTAGE);
        hipFuncSetAttribute((const void*)gemm256s_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * SSTAGE);
        hipFuncSetAttribute((const void*)gemm256s_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * SSTAGE);
        if (hipMalloc((void**)&zero16, 256) != hipSuccess || hipMemset(zero16, 0, 256) != hipSuccess) stagger = 0;
    }
    const bool ks_ok = (layout == 0) || (layout == 1 && p.ldb % 8 == 0 && p.N % 8 == 0) ||
                       (layout == 2 && p.lda % 8 == 0 && p.M % 8 == 0 && p.ldb % 8 == 0 && p.N % 8 == 0 && p.K % 8 == 0);
    if (((stagger >> layout) & 1) && p.K >= 2 * SK && ks_ok) {
        if (layout == 0) hipLaunchKernelGGL((gemm256s_kernel<false, false>), dim3(tiles), dim3(512), NSTAGE * SSTAGE, stream, p, (const bf16_t*)zero16);
        else if (layout == 1) hipLaunchKernelGGL((gemm256s_kernel<false, true>), dim3(tiles), dim3(512), NSTAGE * SSTAGE, stream, p, (const bf16_t*)zero16);
        else hipLaunchKernelGGL((gemm256s_kernel<true, true>), dim3(tiles), dim3(512), NSTAGE * SSTAGE, stream, p, (const bf16_t*)zero16);
        return true;
    }
    if (layout == 0) hipLaunchKernelGGL((gemm256_kernel<false, false>), dim3(tiles), dim3(512), 2 * STAGE_BYTES, stream, p);
    else if (layout == 1) hipLaunchKernelGGL((gemm256_kernel<false, true>), dim3(tiles), dim3(512), 2 * STAGE_BYTES, stream, p);
    else hipLaunchKernelGGL((gemm256_kernel<true, true>), dim3(tiles), dim3(512), 2 * STAGE_BYTES, stream, p);
    return true;
}
