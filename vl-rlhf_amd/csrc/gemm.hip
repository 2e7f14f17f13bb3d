// bf16 MFMA GEMM for gfx950 (MI355X): C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ).
//
// One kernel template covers the three operand layouts the DPO step needs
//   NT  forward      Y  = X  . W^T      A: [M,K] k-contiguous      B: [N,K] k-contiguous
//   NN  dgrad        dX = dY . W        A: [M,K] k-contiguous      B: stored [K,N] (k strided)
//   TN  wgrad        dW = dY^T . X      A: stored [K,M] (k strided) B: stored [K,N] (k strided)
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles.
// Global -> registers -> LDS staging (k-strided operands are transposed 4x4/8x4 in registers with v_perm so LDS
// always holds [row][k] with k contiguous), XOR-swizzled 16-byte chunks so every ds_read_b128 lane group hits 16
// distinct slots, LDS double buffer with the next tile's global loads in flight across the MFMA block, one barrier
// per K step.  Epilogue goes through LDS (per-wave 64x64 fp32) so bias / activation / residual / accumulate are
// applied in fp32 and global stores are 16 B per lane along rows.
// Workgroup -> tile map: XCD-aware (block b runs on XCD b%8; each XCD gets a contiguous run of tiles) and grouped
// (8 tile-rows x all tile-columns) so co-resident workgroups share A panels and B panels in their XCD's L2.
#include <stdlib.h>

#include "gemm.h"
#include "gemm128_epilogue.h"

#define BM 128
#define BN 128
#define BK 64

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- k-contiguous operand: 128 rows x 64 k, thread -> (row = t/8 + 32 i, chunk = t%8) ------------------------
// keep bits of the 8 consecutive elements of hash group g (the mask of vlr_dropout: elementwise.hip / common.h)
__device__ __forceinline__ uint32_t gemm_keep8(uint64_t key, long g, uint32_t thr) {
    const uint64_t r0 = vlr_mix64(key ^ (uint64_t)(2 * g)), r1 = vlr_mix64(key ^ (uint64_t)(2 * g + 1));
    uint32_t keep = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        keep |= (uint32_t)(((r0 >> (16 * e)) & 0xffffu) >= thr) << e;
        keep |= (uint32_t)(((r1 >> (16 * e)) & 0xffffu) >= thr) << (4 + e);
    }
    return keep;
}
// zero the dropped ones of 8 packed bf16 (keep bit e = element e)
__device__ __forceinline__ u32x4 mask8(u32x4 v, uint32_t keep) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] &= ((keep >> (2 * i)) & 1 ? 0x0000ffffu : 0u) | ((keep >> (2 * i + 1)) & 1 ? 0xffff0000u : 0u);
    return v;
}
template <bool MASK = false>
__device__ __forceinline__ void load_kc(const bf16_t* __restrict__ P, int ld, int row0, int nrows, int k0, int K, int t,
                                        u32x4 (&r)[4], uint64_t key = 0, uint32_t thr = 0, int mld = 0, const unsigned char* bits = nullptr) {
    const int c = t & 7;
    const int k = k0 + c * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + (t >> 3) + 32 * i;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < nrows && k + 8 <= K) {
            v = *reinterpret_cast<const u32x4*>(P + (size_t)row * ld + k);
            if constexpr (MASK) {         // mld % 8 == 0, k % 8 == 0: one hash group = one byte of the packed mask
                const long gq = ((long)row * mld + k) >> 3;
                v = mask8(v, bits ? (uint32_t)bits[gq] : gemm_keep8(key, gq, thr));
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void store_kc(char* lds, int t, const u32x4 (&r)[4]) {
    const int c = t & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (t >> 3) + 32 * i;
        *reinterpret_cast<u32x4*>(lds + swz_off(row, c)) = r[i];
    }
}
// ---- k-strided operand: stored [K][ncols]; tile 64 k x 128 cols; thread -> (k block = t/32, col quad = t%32) ----
template <bool MASK = false>
__device__ __forceinline__ void load_ks(const bf16_t* __restrict__ P, int ld, int col0, int ncols, int k0, int K, int t,
                                        u32x2 (&r)[8], uint64_t key = 0, uint32_t thr = 0, int mld = 0, const unsigned char* bits = nullptr) {
    const int kb = t >> 5, nq = t & 31;
    const int col = col0 + nq * 4;
    const bool cok = col + 4 <= ncols;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + kb * 8 + j;
        u32x2 v = {0u, 0u};
        if (cok && k < K) {
            v = *reinterpret_cast<const u32x2*>(P + (size_t)k * ld + col);
            if constexpr (MASK) {         // element (row k, columns col .. col + 3): half a hash group
                const long gq = ((long)k * mld + col) >> 3;
                const uint32_t keep = (bits ? (uint32_t)bits[gq] : gemm_keep8(key, gq, thr)) >> (col & 4);
                v[0] &= (keep & 1 ? 0x0000ffffu : 0u) | (keep & 2 ? 0xffff0000u : 0u);
                v[1] &= (keep & 4 ? 0x0000ffffu : 0u) | (keep & 8 ? 0xffff0000u : 0u);
            }
        }
        r[j] = v;
    }
}
__device__ __forceinline__ void store_ks(char* lds, int t, const u32x2 (&r)[8]) {
    const int kb = t >> 5, nq = t & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = nq * 4 + i;
        u32x4 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t lo = r[2 * p][i >> 1], hi = r[2 * p + 1][i >> 1];
            o[p] = (i & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        *reinterpret_cast<u32x4*>(lds + swz_off(row, kb)) = o;
    }
}

template <bool A_KS, bool B_KS, int MASK = 0>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * BK * 2];  // 64 KiB
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    uint64_t mkey = 0;
    if (p.groups > 1 || MASK) {      // grouped launch: this workgroup's problem; the mask of group g is vlr_dropout(mask_seed + g)
        const int g = blockIdx.z;
        p.A += (size_t)g * p.gA;
        p.B += (size_t)g * p.gB;
        p.C = (char*)p.C + (size_t)g * p.gC * (p.out_f32 ? 4 : 2);
        if (p.splitk > 1) p.part += (size_t)g * p.splitk * p.M * p.N;
        mkey = vlr_mix64(p.mask_seed + (uint64_t)g);
        if (p.mask_bits) p.mask_bits += (size_t)g * p.gMask;
    }
    int kabs0 = 0;                   // split-K slices move p.A / p.B: the mask index uses the ABSOLUTE k
    if (p.splitk > 1) {   // split-K slice: this workgroup reduces k in [z*kchunk, (z+1)*kchunk) into its own fp32 partial
        const int z = blockIdx.y, k0 = z * p.kchunk;
        kabs0 = k0;
        p.A += A_KS ? (size_t)k0 * p.lda : (size_t)k0;
        p.B += B_KS ? (size_t)k0 * p.ldb : (size_t)k0;
        p.K = min(p.kchunk, p.K - k0);
        p.C = p.part + (size_t)z * p.M * p.N;
        p.ldc = p.N; p.out_f32 = 1; p.bias = nullptr; p.residual = nullptr; p.accumulate = 0; p.act = 0;
    }

    // ---- XCD-aware, grouped tile map (bijective for any tile count)
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
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
    const int m0 = tm * BM, n0 = tn * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra_c[4], rb_c[4];
    u32x2 ra_s[8], rb_s[8];
    const int nt = (p.K + BK - 1) / BK;

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        if constexpr (A_KS) load_ks(p.A, p.lda, m0, p.M, k0, p.K, t, ra_s);
        else if constexpr (MASK == 1) {
            // x rows [M][mask_ld] masked while staged: the pointer already carries the split-K offset, the hash index needs it back
            load_kc<true>(p.A - kabs0, p.lda, m0, p.M, k0 + kabs0, p.K + kabs0, t, ra_c, mkey, p.mask_thr, p.mask_ld, p.mask_bits);
        } else load_kc(p.A, p.lda, m0, p.M, k0, p.K, t, ra_c);
        if constexpr (B_KS) {
            if constexpr (MASK == 2) load_ks<true>(p.B - (size_t)kabs0 * p.ldb, p.ldb, n0, p.N, k0 + kabs0, p.K + kabs0, t, rb_s, mkey, p.mask_thr, p.mask_ld, p.mask_bits);
            else load_ks(p.B, p.ldb, n0, p.N, k0, p.K, t, rb_s);
        } else load_kc(p.B, p.ldb, n0, p.N, k0, p.K, t, rb_c);
    };
    auto lstore = [&](int buf) {
        char* a = smem + buf * (BM + BN) * BK * 2;
        char* b = a + BM * BK * 2;
        if constexpr (A_KS) store_ks(a, t, ra_s);
        else store_kc(a, t, ra_c);
        if constexpr (B_KS) store_ks(b, t, rb_s);
        else store_kc(b, t, rb_c);
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nt) gload(kt + 1);
        const char* a = smem + cur * (BM + BN) * BK * 2;
        const char* b = a + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[2], fb[2];
            const int c = kk * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + (lane & 31);
                fa[i] = *reinterpret_cast<const bf16x8*>(a + swz_off(row, c));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + (lane & 31);
                fb[j] = *reinterpret_cast<const bf16x8*>(b + swz_off(row, c));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nt) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue through LDS: per-wave 64x64 fp32 staging
    float* stage = reinterpret_cast<float*>(smem) + wave * 64 * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = j * 32 + (lane & 31);
                stage[row * 64 + col] = acc[i][j][r];
            }
    __syncthreads();
    gemm128_copy_out<64>(p, stage, m0 + wm * 64, n0 + wn * 64, lane);
}

// ------------------------------------------------------------------------------------------------------------
// LoRA input gradient of the targets that share one input, in ONE pass over dx:
//   dx [M][in] (+)= sum_t keep_t(row * in + col) ? alpha * (v_t [M][r] . A_t [r][in]) : 0,  keep_t = the mask of vlr_dropout(seed + t)
// (q, k, v -> d xn1; gate, up -> d xn2; o -> d attn; down -> d act).  Per term: a K = r loop (two 64-deep steps at r = 128) into the
// MFMA accumulators, through the per-wave fp32 LDS stage into row-major registers (8 rows x 8 consecutive columns per lane: one hash
// group each), masked and summed there; then ONE read-modify-write (or write) of the bf16 tile.  GemmParams: A = v (lda = ldv, term
// stride gA = r columns), B = A_t stack (ldb = in, term stride gB = r * in), C = dx, N = in, K = r, groups = terms.
__global__ __launch_bounds__(256) void dropacc_multi_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * BK * 2 * 2];  // 64 KiB: operand tiles / the fp32 stage
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        const int q = nwg >> 3, rem = nwg & 7;
        pid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int m0 = (pid / tiles_n) * BM, n0 = (pid % tiles_n) * BN;      // row-major walk: neighbours share the v rows
    const int gm0 = m0 + wm * 64, gn0 = n0 + wn * 64;
    const int cq = (lane & 7) * 8, gn = gn0 + cq;
    float sum[8][8];
#pragma unroll
    for (int it = 0; it < 8; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[it][e] = 0.f;
    const int nt = (p.K + BK - 1) / BK;
    float* stage = reinterpret_cast<float*>(smem) + wave * 64 * 64;
    for (int term = 0; term < p.groups; ++term) {
        const bf16_t* A = p.A + (size_t)term * p.gA;
        const bf16_t* B = p.B + (size_t)term * p.gB;
        const uint64_t key = vlr_mix64(p.mask_seed + (uint64_t)term);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nt; ++kt) {
            u32x4 ra[4];
            u32x2 rb[8];
            load_kc(A, p.lda, m0, p.M, kt * BK, p.K, t, ra);
            load_ks(B, p.ldb, n0, p.N, kt * BK, p.K, t, rb);
            __syncthreads();                               // the previous step's fragment reads / the previous term's stage reads are done
            store_kc(smem, t, ra);
            store_ks(smem + BM * BK * 2, t, rb);
            __syncthreads();
            const char* a = smem;
            const char* b = smem + BM * BK * 2;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 fa[2], fb[2];
                const int c = kk * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(a + swz_off(wm * 64 + i * 32 + (lane & 31), c));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(b + swz_off(wn * 64 + j * 32 + (lane & 31), c));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int gm = gm0 + row;
            if (gm < p.M && gn + 8 <= p.N) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + cq);
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(stage + row * 64 + cq + 4);
                const long gq = ((long)gm * p.mask_ld + gn) >> 3;
                const uint32_t keep = p.mask_bits ? (uint32_t)p.mask_bits[(size_t)term * p.gMask + gq] : gemm_keep8(key, gq, p.mask_thr);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if ((keep >> e) & 1) sum[it][e] += p.alpha * s0[e];
                    if ((keep >> (4 + e)) & 1) sum[it][4 + e] += p.alpha * s1[e];
                }
            }
        }
    }
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int gm = gm0 + it * 8 + (lane >> 3);
        if (gm < p.M && gn + 8 <= p.N) {
            bf16_t* dst = C + (size_t)gm * p.ldc + gn;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = sum[it][e];
            if (p.accumulate) {
                float ov[8];
                unpack8(*reinterpret_cast<const u32x4*>(dst), ov);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += ov[e];
            }
            *reinterpret_cast<u32x4*>(dst) = pack8(v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// split-K epilogue: C = act(sum_z part[z] + bias) + residual (+ C), 4 columns per thread (the partials carry alpha already)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N, void* Cv,
                                                            int ldc, int out_f32, int accumulate,
                                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual,
                                                            int ldr, int act, int res_f32, long gC) {
    // grouped launches: blockIdx.y = group (partials [group][slice][M][N], C + group * gC elements)
    part += (size_t)blockIdx.y * splits * M * N;
    Cv = (char*)Cv + (size_t)blockIdx.y * gC * (out_f32 ? 4 : 2);
    const long n4 = (long)M * N / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(part + i * 4);
        for (int z = 1; z < splits; ++z) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(part + (size_t)z * M * N + i * 4);
            v += w;
        }
        const long m = (i * 4) / N, n = (i * 4) % N;
        if (bias) {
            const u32x2 w = *reinterpret_cast<const u32x2*>(bias + n);
            v[0] += bf16lo(w[0]); v[1] += bf16hi(w[0]); v[2] += bf16lo(w[1]); v[3] += bf16hi(w[1]);
        }
        if (act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
        }
        if (residual) {
            if (res_f32) {
                v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(residual) + m * ldr + n);
            } else {
                const u32x2 w = *reinterpret_cast<const u32x2*>(residual + m * ldr + n);
                v[0] += bf16lo(w[0]); v[1] += bf16hi(w[0]); v[2] += bf16lo(w[1]); v[3] += bf16hi(w[1]);
            }
        }
        if (out_f32) {
            float* dst = reinterpret_cast<float*>(Cv) + m * ldc + n;
            if (accumulate) v += *reinterpret_cast<const f32x4*>(dst);
            *reinterpret_cast<f32x4*>(dst) = v;
        } else {
            bf16_t* dst = reinterpret_cast<bf16_t*>(Cv) + m * ldc + n;
            if (accumulate) {
                const u32x2 o = *reinterpret_cast<const u32x2*>(dst);
                v[0] += bf16lo(o[0]); v[1] += bf16hi(o[0]); v[2] += bf16lo(o[1]); v[3] += bf16hi(o[1]);
            }
            u32x2 w;
            w[0] = pack_bf16(v[0], v[1]);
            w[1] = pack_bf16(v[2], v[3]);
            *reinterpret_cast<u32x2*>(dst) = w;
        }
    }
}

// The registered scratch is cut in two slots, one per stream (the DPO step runs the frozen reference forward on a side
// stream next to the policy forward): the first two distinct streams that ask get a slot each, any further stream runs
// un-split.  Re-registering forgets the stream assignment.
#define SPLITK_SLOT_BYTES (64L << 20)
#define SPLITK_MAX_SLOTS 8
static float* g_splitk_ws = nullptr;
static long g_splitk_bytes = 0;          // bytes per slot
static int g_splitk_nslots = 0;
static hipStream_t g_splitk_stream[SPLITK_MAX_SLOTS];
static int g_splitk_nstream = 0;
extern "C" int vlr_gemm_set_splitk_workspace(void* ws, long bytes) {
    VLR_REQUIRE((ws && bytes > 0) || (!ws && bytes == 0), "vlr_gemm_set_splitk_workspace: (ptr, bytes) or (NULL, 0)");
    g_splitk_ws = (float*)ws;
    // slots of 128 MiB, one per stream, at most 8; below 256 MiB: 64 MiB slots (the split-K partials of the 7B shapes fit those); a still
    // smaller buffer is cut in two.  (Round 3 kept per-wave flags and accumulator slabs of the stream-K experiments in these slots; gone.)
    int n = (int)(bytes / (128L << 20));
    if (n > SPLITK_MAX_SLOTS) n = SPLITK_MAX_SLOTS;
    if (n >= 2) {
        g_splitk_nslots = n; g_splitk_bytes = 128L << 20;
    } else {
        n = (int)(bytes / SPLITK_SLOT_BYTES);
        if (n > SPLITK_MAX_SLOTS) n = SPLITK_MAX_SLOTS;
        if (n >= 2) { g_splitk_nslots = n; g_splitk_bytes = SPLITK_SLOT_BYTES; }
        else { g_splitk_nslots = ws ? 2 : 0; g_splitk_bytes = (bytes / 2) & ~255L; }
    }
    g_splitk_nstream = 0;
    return VLR_OK;
}
// A/B switches of the continuous-pipeline GEMM kernels (GemmParams::sched): bit 3 (8) adapter K tiles on the general staging path,
// bit 4 (16) the two wave groups of a workgroup run their epilogues one after the other (the order before round 4), bit 5 (32) the
// shared-panel tile map (gemm_tilemap.h; the production default since round 5).  Default from VLR_GEMM_SCHED (else 32);
// vlr_gemm_set_sched(-1) re-reads the environment.  The tile schedules of round 3 (bits 0-2: stream-K tail, XCD
// rotation, XCD round barrier) measured slower or neutral and were removed.
static int g_sched = -1;
int vlr_gemm_sched_mode() {
    if (g_sched < 0) { const char* e = getenv("VLR_GEMM_SCHED"); g_sched = e ? (atoi(e) & 56) : VLR_SCHED_DEFAULT; }
    return g_sched;
}
extern "C" int vlr_gemm_set_sched(int mode) {
    VLR_REQUIRE(mode == -1 || (mode >= 0 && (mode & ~56) == 0),
                "vlr_gemm_set_sched: a sum of 8 (adapter tiles on the general staging path), 16 (serial epilogue order), 32 (shared-panel tile map), or -1, got %d "
                "(the stream-K / rotation / round-barrier schedules 1-7 of round 3 were removed: measured slower)", mode);
    g_sched = mode;
    return VLR_OK;
}
// Which split a GEMM takes must not depend on which OTHER streams happened to run split-K GEMMs earlier in the process: the
// reference pass (side stream) and the policy pass (main stream) of one step have to produce bit-identical results for
// identical weights (policy == reference => loss == ln 2 exactly).  With two slots, a third stream - e.g. a new trainer's side
// stream after an earlier trainer's - silently ran un-split and summed in a different order (found by the full-size
// LLaVA-Next test, which failed or passed depending on the test order).  Now: one slot per stream, eight of them, and the
// engine re-registers the workspace (forgetting the assignment) when it is constructed.
static float* splitk_slot(hipStream_t st) {
    if (!g_splitk_ws) return nullptr;
    for (int i = 0; i < g_splitk_nstream; ++i)
        if (g_splitk_stream[i] == st) return (float*)((char*)g_splitk_ws + (size_t)i * g_splitk_bytes);
    if (g_splitk_nstream < g_splitk_nslots) {
        g_splitk_stream[g_splitk_nstream] = st;
        return (float*)((char*)g_splitk_ws + (size_t)(g_splitk_nstream++) * g_splitk_bytes);
    }
    return nullptr;
}
// split-K launch of the 128x128 kernel: few output tiles, long reduction.  Returns false when it does not apply.
// one launch of the 128x128-tile GEMM: the LDS-DMA ring kernel (gemm128p.hip) when the operands qualify, else the register-staged one
static void launch128(int layout, const GemmParams& p, dim3 grid, hipStream_t stream) {
    if (vlr_gemm128p_try_launch(layout, p, grid, stream)) return;
    if (layout == 0) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, dim3(256), 0, stream, p);
    else if (layout == 1) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, dim3(256), 0, stream, p);
}
static bool launch_splitk128(int layout, GemmParams p, hipStream_t stream, int min_k) {
    if (p.N % 4 != 0 || p.K < min_k) return false;
    const int tiles128 = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (tiles128 >= 256) return false;
    static int target = -1;       // workgroups a split launch aims at (VLR_SPLITK_TARGET; 2 per CU)
    if (target < 0) { const char* e = getenv("VLR_SPLITK_TARGET"); target = e ? atoi(e) : 512; if (target < 64) target = 512; }
    int splits = target / tiles128;
    if (splits > 16) splits = 16;
    if (splits > p.K / 256) splits = p.K / 256;
    if (splits < 2) return false;
    float* ws = splitk_slot(stream);
    if (!ws) return false;
    const long per = (long)p.M * p.N * 4;
    const long cap = g_splitk_bytes < SPLITK_SLOT_BYTES ? g_splitk_bytes : SPLITK_SLOT_BYTES;     // one bound for every user of a slot (gemm_grouped has the same)
    if ((long)splits * per > cap) splits = (int)(cap / per);
    if (splits < 2) return false;
    const int kchunk = (((p.K + splits - 1) / splits) + 63) / 64 * 64;      // whole K tiles per slice
    splits = (p.K + kchunk - 1) / kchunk;
    p.splitk = splits; p.kchunk = kchunk; p.part = ws;
    launch128(layout, p, dim3(tiles128, splits), stream);
    const long n4 = (long)p.M * p.N / 4;
    int rg = (int)((n4 + 255) / 256);
    if (rg > 2048) rg = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, stream, ws, splits, p.M, p.N, p.C, p.ldc, p.out_f32,
                       p.accumulate, p.bias, p.residual, p.ldr, p.act, p.res_f32, 0L);
    return true;
}

static GemmParams fused_params(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc);
// The skinny per-target GEMMs of a LoRA group in ONE launch of the 128x128 kernel (GemmParams::groups): group g runs [M, N, K] on
// A + g gA, B + g gB, C + g gC (bf16 out, C = alpha * A.B (+ C)); split along K when that fills the chip; mask_on 1 / 2 zeroes the
// dropped elements of the activation operand while it is staged (keep mask of vlr_dropout(seed + g) over [rows][mask_ld]).
static int gemm_grouped(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int groups,
                        long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed, float p_drop, int mask_ld,
                        const void* mask_bits, long mask_gstride, hipStream_t stream, const unsigned char* rowskip = nullptr,
                        const int* ktlist = nullptr) {
    VLR_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && groups >= 1 && groups <= 8, "gemm_grouped: bad arguments");
    VLR_REQUIRE(N % 8 == 0 && ldc % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && gC % 4 == 0, "gemm_grouped: alignment (N %d lda %d ldb %d ldc %d)", N, lda, ldb, ldc);
    VLR_REQUIRE(!mask_on || (mask_ld % 8 == 0 && ((mask_on == 1 && layout == 0) || ((mask_on == 2 || mask_on == 3) && layout == 2))), "gemm_grouped: mask on the NT A / TN B operand only");
    VLR_REQUIRE(mask_on != 3 || mask_bits, "gemm_grouped: mask_on 3 needs the K-tile-blocked transposed masks of vlr_dropout_bits2");
    GemmParams p = fused_params(A, B, C, M, N, K, lda, ldb, ldc);
    p.alpha = alpha; p.accumulate = accumulate;
    p.rowskip = (layout != 2 && !accumulate) ? rowskip : nullptr;      // A row-major: its rows are the output rows
    p.ktlist = (layout == 2 && K % 64 == 0) ? ktlist : nullptr;        // (only the ring kernel reads it; any other kernel contracts over all of K: same product)
    p.groups = groups; p.gA = gA; p.gB = gB; p.gC = gC;
    p.mask_on = mask_on; p.mask_seed = seed; p.mask_thr = vlr_dropout_thr(p_drop); p.mask_ld = mask_ld;
    p.mask_bits = mask_on ? (const unsigned char*)mask_bits : nullptr; p.gMask = mask_gstride;
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int splits = 1;
    float* ws = nullptr;
    static int gtarget = -1;      // workgroups a grouped launch aims at when it splits along K (VLR_GROUPED_TARGET)
    if (gtarget < 0) { const char* e = getenv("VLR_GROUPED_TARGET"); gtarget = e ? atoi(e) : 512; if (gtarget < 64) gtarget = 512; }
    if (tiles * groups < gtarget * 3 / 4 && K >= 1024) {
        splits = gtarget / (tiles * groups);       // (rounded down on purpose: 2 x 300 workgroups of half the depth measured slower than 300 whole ones)
        if (splits > 16) splits = 16;
        if (splits > K / 256) splits = K / 256;
        ws = splits >= 2 ? splitk_slot(stream) : nullptr;
        const long per = (long)groups * M * N * 4;
        const long cap = g_splitk_bytes < SPLITK_SLOT_BYTES ? g_splitk_bytes : SPLITK_SLOT_BYTES;
        if (ws && (long)splits * per > cap) splits = (int)(cap / per);
        if (!ws || splits < 2) splits = 1;
    }
    if (splits > 1) {
        const int kchunk = (((K + splits - 1) / splits) + 63) / 64 * 64;      // whole K tiles per slice (the ring kernel masks whole tiles)
        splits = (K + kchunk - 1) / kchunk;
        p.splitk = splits; p.kchunk = kchunk; p.part = ws;
    }
    const dim3 grid(tiles, splits, groups);
    const int pi = vlr_prof_begin(layout, 2.0 * M * N * K * groups, stream);
    if (mask_on && layout == 0) {       // packed masks: the LDS-DMA ring kernel masks the fragments; else staged through registers
        if (!vlr_gemm128p_try_launch(layout, p, grid, stream)) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, 1>), grid, dim3(256), 0, stream, p);
    } else if (mask_on == 3) {          // TN with the transposed packed masks: ring kernel, else the register-staged kernel hashing the same mask
        if (!vlr_gemm128p_try_launch(layout, p, grid, stream)) {
            p.mask_on = 2; p.mask_bits = nullptr;
            hipLaunchKernelGGL((gemm_bf16_kernel<true, true, 2>), grid, dim3(256), 0, stream, p);
        }
    } else if (mask_on) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, 2>), grid, dim3(256), 0, stream, p);
    else launch128(layout, p, grid, stream);
    if (splits > 1) {
        const long n4 = (long)M * N / 4;
        int rg = (int)((n4 + 255) / 256);
        if (rg > 1024) rg = 1024;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg, groups), dim3(256), 0, stream, ws, splits, M, N, C, ldc, 0, accumulate,
                           (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0, 0, 0, gC);
    }
    vlr_prof_end(pi, stream);
    return vlr_check_launch("gemm_grouped");
}
// C-ABI face of the grouped skinny GEMM (the decoder-layer LoRA passes of layers.cpp; tests)
extern "C" int vlr_gemm_grouped(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                int groups, long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed,
                                float p_drop, int mask_ld, hipStream_t stream) {
    VLR_REQUIRE(layout >= 0 && layout <= 2, "vlr_gemm_grouped: layout %d", layout);
    VLR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "vlr_gemm_grouped: 0 <= p < 1, got %g", (double)p_drop);
    return gemm_grouped(layout, A, B, C, M, N, K, lda, ldb, ldc, groups, gA, gB, gC, alpha, accumulate, mask_on, seed, p_drop, mask_ld, nullptr, 0, stream);
}
// the same with the keep masks drawn beforehand (vlr_dropout_bits(seed + g) over the [rows][mask_ld] operand, group g at mask_bits +
// g * mask_gstride bytes): the kernels read one byte per eight elements instead of hashing
extern "C" int vlr_gemm_grouped_bits(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                     int groups, long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed,
                                     float p_drop, int mask_ld, const void* mask_bits, long mask_gstride, hipStream_t stream) {
    VLR_REQUIRE(layout >= 0 && layout <= 2, "vlr_gemm_grouped_bits: layout %d", layout);
    VLR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "vlr_gemm_grouped_bits: 0 <= p < 1, got %g", (double)p_drop);
    return gemm_grouped(layout, A, B, C, M, N, K, lda, ldb, ldc, groups, gA, gB, gC, alpha, accumulate, mask_on, seed, p_drop, mask_ld, mask_bits,
                        mask_gstride, stream);
}

// ... layout 2 (C = A^T B, K = token rows) contracting only over the K tiles of `ktlist` (vlr_rows_tile_list: the 64-row tiles that hold a
// marked row; the caller guarantees that the product of every other row is zero - PLoRA's dB = dy^T u and dA = v^T drop(x), u / v zero on
// the text rows).  K % 64 != 0 or a kernel other than the 128x128 ring kernel: the list is ignored (same result, all of K read).
extern "C" int vlr_gemm_grouped_bits_ktiles(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                            int groups, long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed,
                                            float p_drop, int mask_ld, const void* mask_bits, long mask_gstride, const int* ktlist,
                                            hipStream_t stream) {
    VLR_REQUIRE(layout == 2, "vlr_gemm_grouped_bits_ktiles: layout 2 only, got %d", layout);
    VLR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "vlr_gemm_grouped_bits_ktiles: 0 <= p < 1, got %g", (double)p_drop);
    return gemm_grouped(layout, A, B, C, M, N, K, lda, ldb, ldc, groups, gA, gB, gC, alpha, accumulate, mask_on, seed, p_drop, mask_ld, mask_bits,
                        mask_gstride, stream, nullptr, ktlist);
}
// ... and with a row mask [M] (1 = the row takes part): layouts 0 / 1 (A row-major); 128-row output tiles without a marked row are NOT computed, the
// caller zeroes the unmarked rows afterwards (vlr_rows_mask) - the u GEMMs of InternLM-XComposer2's PLoRA skip the all-text tiles
extern "C" int vlr_gemm_grouped_bits_rows(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                          int groups, long gA, long gB, long gC, float alpha, int accumulate, int mask_on, uint64_t seed,
                                          float p_drop, int mask_ld, const void* mask_bits, long mask_gstride, const unsigned char* rowmask,
                                          hipStream_t stream) {
    VLR_REQUIRE((layout == 0 || layout == 1) && !accumulate, "vlr_gemm_grouped_bits_rows: layout 0 / 1, accumulate 0 only (layout %d accumulate %d)", layout, accumulate);
    VLR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "vlr_gemm_grouped_bits_rows: 0 <= p < 1, got %g", (double)p_drop);
    return gemm_grouped(layout, A, B, C, M, N, K, lda, ldb, ldc, groups, gA, gB, gC, alpha, accumulate, mask_on, seed, p_drop, mask_ld, mask_bits,
                        mask_gstride, stream, rowmask);
}

// dx [M][in] (+)= sum over the n targets t of scale / (1 - p) * keep_t . (v_t . A_t): v [M][ldv] holds the n blocks of r columns side by
// side, A the n stacked [r][in] lora_A matrices, keep_t = the mask of vlr_dropout(seed + t) over [M][in] (p = 0: no mask).  ONE pass
// over dx whatever n is (the per-target vlr_gemm_dropout_acc makes n).  accumulate = 0 writes dx instead of adding to it.
extern "C" int vlr_gemm_dropout_acc_multi_bits(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p,
                                               uint64_t seed, float scale, int accumulate, const void* bits, long bits_gstride,
                                               hipStream_t stream);
extern "C" int vlr_gemm_dropout_acc_multi(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p,
                                          uint64_t seed, float scale, int accumulate, hipStream_t stream) {
    return vlr_gemm_dropout_acc_multi_bits(n, v, ldv, A, dx, M, in, r, p, seed, scale, accumulate, nullptr, 0, stream);
}
// bits != NULL: the packed keep masks of the n targets (vlr_dropout_bits(seed + t), target t at bits + t * bits_gstride bytes)
extern "C" int vlr_gemm_dropout_acc_multi_rows(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p,
                                               uint64_t seed, float scale, int accumulate, const void* bits, long bits_gstride,
                                               const unsigned char* rowmask, hipStream_t stream);
extern "C" int vlr_gemm_dropout_acc_multi_bits(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p,
                                               uint64_t seed, float scale, int accumulate, const void* bits, long bits_gstride,
                                               hipStream_t stream) {
    return vlr_gemm_dropout_acc_multi_rows(n, v, ldv, A, dx, M, in, r, p, seed, scale, accumulate, bits, bits_gstride, nullptr, stream);
}
// rowmask [M] (or NULL): the caller guarantees that the v rows of unmarked rows are ZERO (vlr_rows_mask) - with accumulate = 1 the
// streaming kernel then skips every 64-row slab without a marked row (dx += 0)
extern "C" int vlr_gemm_dropout_acc_multi_rows(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p,
                                               uint64_t seed, float scale, int accumulate, const void* bits, long bits_gstride,
                                               const unsigned char* rowmask, hipStream_t stream) {
    VLR_REQUIRE(v && A && dx, "vlr_gemm_dropout_acc_multi: null operand");
    VLR_REQUIRE(n >= 1 && n <= 8 && M > 0 && in > 0 && r > 0 && in % 8 == 0 && r % 8 == 0 && ldv % 8 == 0 && ldv >= n * r,
                "vlr_gemm_dropout_acc_multi: bad shape n=%d M=%d in=%d r=%d ldv=%d", n, M, in, r, ldv);
    VLR_REQUIRE(p >= 0.f && p < 1.f, "vlr_gemm_dropout_acc_multi: 0 <= p < 1 required, got %g", (double)p);
    VLR_REQUIRE(!(((uintptr_t)v | (uintptr_t)A | (uintptr_t)dx) & 15), "vlr_gemm_dropout_acc_multi: 16-byte aligned operands");
    {   // the streaming kernel (lora_dx.hip): v rows in registers, A_t slices through LDS, one read-modify-write of dx
        const int pj = vlr_prof_begin(VLR_K_GEMM_NN, 2.0 * M * in * r * n, stream);
        const bool took = vlr_lora_dx_try_launch(n, v, ldv, A, dx, M, in, r, p, seed, scale, accumulate, bits, bits_gstride, stream, rowmask);
        vlr_prof_end(took ? pj : -1, stream);
        if (took) return vlr_check_launch("vlr_gemm_dropout_acc_multi(streamed)");
    }
    GemmParams g = fused_params(v, A, dx, M, in, r, ldv, in, in);
    g.alpha = scale / (1.f - p); g.accumulate = accumulate;
    g.groups = n; g.gA = r; g.gB = (long)r * in;
    g.mask_seed = seed; g.mask_thr = vlr_dropout_thr(p); g.mask_ld = in;
    g.mask_bits = (const unsigned char*)bits; g.gMask = bits_gstride;
    const int tiles = ((M + BM - 1) / BM) * ((in + BN - 1) / BN);
    const int pi = vlr_prof_begin(VLR_K_GEMM_NN, 2.0 * M * in * r * n, stream);
    hipLaunchKernelGGL(dropacc_multi_kernel, dim3(tiles), dim3(256), 0, stream, g);
    vlr_prof_end(pi, stream);
    return vlr_check_launch("vlr_gemm_dropout_acc_multi");
}

static int gemm_impl_ex(int layout, const void* A, const void* B, void* C, const void* bias, const void* residual, int M, int N,
                        int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate, int out_f32, float alpha, int res_f32,
                        hipStream_t stream);
static int gemm_impl(int layout, const void* A, const void* B, void* C, const void* bias, const void* residual, int M, int N,
                     int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate, int out_f32, float alpha,
                     hipStream_t stream) {
    return gemm_impl_ex(layout, A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, act, accumulate, out_f32, alpha, 0, stream);
}

// wave quantisation (see gemm_impl): how many of the last 256-row tile rows to peel off so that the 256x256-tile part is a
// whole number of rounds on the 256 CUs; tn = workgroup tiles per tile row
// most tile rows a launch may give to the 128x128 kernel: 3 on the whole chip (the shapes of the path were tuned with it), 6 when CUs are left
// to RCCL - 240-CU rounds fit the 7B shapes worse (12792 x 4096: 800 tiles = 3.33 rounds; 45 tile rows = 3 rounds exactly + 1272 rows peeled)
static int peel_max(int ncu) {
    static int env = -1;
    if (env < 0) { const char* e = getenv("VLR_PEEL_MAX"); env = e ? atoi(e) : 0; if (env < 0 || env > 8) env = 0; }
    if (env) return env;
    int dev = 0, cus = 0;
    static int dev_cus = 0;
    if (!dev_cus) dev_cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 8) ? (cus & ~7) : 256;
    return ncu < dev_cus ? 6 : 3;
}
static int choose_peel(int M, int N, int tn, int K = 0, hipStream_t stream = nullptr, bool tail256 = false) {
    const int tm256 = (M + 255) / 256;
    int peel = 0;
    const int ncu = vlr_compute_cus();          // workgroups of a persistent round (256 on MI355X; fewer when CUs are left to RCCL)
    if ((long)tm256 * tn >= 2 * ncu) {
        const int full = tm256 * tn;
        const double base = (double)((full + ncu - 1) / ncu);
        double best = base;
        const int rmax = peel_max(ncu);
        for (int r = 1; r <= rmax && tm256 - r >= 2; ++r) {
            const int t1 = (tm256 - r) * tn;
            const int rem_rows = M - (tm256 - r) * 256;
            const long t128 = (long)((rem_rows + 127) / 128) * ((N + 127) / 128);
            // the peeled rows run on the 128x128 kernel (two workgroups per CU, ~0.7 of a 256x256 tile time per round of 2 x ncu), or -
            // fused epilogues (tail256) - as a second launch of the same 256x256 kernel, one cold tile per workgroup
            const double est = (double)((t1 + ncu - 1) / ncu) +
                               (tail256 ? 1.15 * (double)((r * tn + ncu - 1) / ncu) : 0.7 * (double)((t128 + 2 * ncu - 1) / (2 * ncu)));
            if (est < best - 0.05) { best = est; peel = r; }
        }
    }
    return peel;
}

extern "C" int vlr_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* bias, const void* residual,
                             int M, int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate,
                             int out_f32, hipStream_t stream) {
    return gemm_impl(layout, A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, act, accumulate, out_f32, 1.0f, stream);
}
// same, with the accumulator scaled by alpha before the epilogue (LoRA merge: W_eff = W + (alpha/r) B A)
extern "C" int vlr_gemm_bf16_scaled(int layout, const void* A, const void* B, void* C, const void* bias,
                                    const void* residual, int M, int N, int K, int lda, int ldb, int ldc, int ldr, int act,
                                    int accumulate, int out_f32, float alpha, hipStream_t stream) {
    return gemm_impl(layout, A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, act, accumulate, out_f32, alpha, stream);
}

// Two weight-gradient GEMMs of one layer (TN: C_i [M_i][N_i] = A_i^T B_i over the same K token rows) as ONE persistent launch: their tile
// counts add up before they are rounded to whole rounds of the CUs (LLaVA-1.5-7B: dW_gate|up 1376 tiles + dW_down 688 = 8.06 rounds
// instead of 6 + 3; one tile row of 16 tiles goes to the 128x128 kernel split along K and 8 rounds remain).  Falls back to two calls.
// rounds of the two launches apart, and the best (problem, tile rows peeled to the 128x128 kernel) for the joint launch, priced like choose_peel
static double tn_pair_plan(int M0, int N0, int M1, int N1, int* sep_out, int* bq_out, int* br_out) {
    const int ncu = vlr_compute_cus();
    const int tm[2] = {(M0 + 255) / 256, (M1 + 255) / 256}, tn[2] = {(N0 + 255) / 256, (N1 + 255) / 256};
    const int Ms[2] = {M0, M1}, Ns[2] = {N0, N1};
    *sep_out = (tm[0] * tn[0] + ncu - 1) / ncu + (tm[1] * tn[1] + ncu - 1) / ncu;
    double best = 1e30;
    const int rmax = peel_max(ncu);
    *bq_out = -1; *br_out = 0;
    for (int q = 0; q < 2; ++q)
        for (int r = 0; r <= rmax && tm[q] - r >= 1; ++r) {
            const int t = tm[0] * tn[0] + tm[1] * tn[1] - r * tn[q];
            const int rem_rows = r ? Ms[q] - (tm[q] - r) * 256 : 0;
            const long t128 = (long)((rem_rows + 127) / 128) * ((Ns[q] + 127) / 128);
            const double est = (double)((t + ncu - 1) / ncu) + 0.7 * (double)((t128 + 2 * ncu - 1) / (2 * ncu));
            if (est < best - 1e-9) { best = est; *bq_out = q; *br_out = r; }
        }
    return best;
}
// rounds the joint launch of two TN problems saves over two launches on the CUs the compute kernels have NOW (layers.cpp: dW_qkv + dW_o are
// 3 + 1 rounds of 256 CUs either way, but 4 + 2 against 5 of 240)
double vlr_internal_tn_pair_saves(int M0, int N0, int M1, int N1) {
    int sep, bq, br;
    const double best = tn_pair_plan(M0, N0, M1, N1, &sep, &bq, &br);
    return (double)sep - best;
}
extern "C" int vlr_gemm_bf16_tn_pair(const void* A0, const void* B0, void* C0, int M0, int N0, int lda0, int ldb0, int ldc0,
                                     const void* A1, const void* B1, void* C1, int M1, int N1, int lda1, int ldb1, int ldc1, int K,
                                     int accumulate, hipStream_t stream) {
    VLR_REQUIRE(A0 && B0 && C0 && A1 && B1 && C1 && M0 > 0 && N0 > 0 && M1 > 0 && N1 > 0 && K > 0, "vlr_gemm_bf16_tn_pair: bad arguments");
    const int tm[2] = {(M0 + 255) / 256, (M1 + 255) / 256};
    const int Ms[2] = {M0, M1};
    int sep, bq, br;
    const double best = tn_pair_plan(M0, N0, M1, N1, &sep, &bq, &br);
    static int pair_on = -1;
    if (pair_on < 0) { const char* e = getenv("VLR_GEMM_PAIR"); pair_on = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1); }
    // the joint launch was measured neutral at equal round counts on the LLaVA-7B shapes only - shapes that were never A/B'd
    // take it when it SAVES at least a quarter of a round (the criterion of vlr_internal_tn_pair_saves' callers); VLR_GEMM_PAIR=2: also at equal counts
    const double need = pair_on == 2 ? -1e-9 : 0.25;
    if (pair_on && !accumulate && (double)sep - best >= need && K % 8 == 0) {
        GemmParams p[2] = {fused_params(A0, B0, C0, M0, N0, K, lda0, ldb0, ldc0), fused_params(A1, B1, C1, M1, N1, K, lda1, ldb1, ldc1)};
        GemmParams rest = p[bq];
        if (br) {
            const int keep = (tm[bq] - br) * 256;
            p[bq].M = keep;
            rest.M = Ms[bq] - keep;
            rest.A = p[bq].A + keep;                                  // TN: the output rows are columns of A [K][lda]
            rest.C = (bf16_t*)p[bq].C + (size_t)keep * p[bq].ldc;
        }
        const int pi = vlr_prof_begin(2, 2.0 * K * ((double)M0 * N0 + (double)M1 * N1), stream);
        if (vlr_gemm256p_tn_pair_try_launch(p[0], p[1], stream)) {
            if (br && !launch_splitk128(2, rest, stream, 4096))
                launch128(2, rest, dim3(((rest.M + BM - 1) / BM) * ((rest.N + BN - 1) / BN)), stream);
            vlr_prof_end(pi, stream);
            return vlr_check_launch("vlr_gemm_bf16_tn_pair");
        }
        vlr_prof_end(-1, stream);
    }
    int rc = vlr_gemm_bf16(2, A0, B0, C0, nullptr, nullptr, M0, N0, K, lda0, ldb0, ldc0, 0, 0, accumulate, 0, stream);
    if (rc != VLR_OK) return rc;
    return vlr_gemm_bf16(2, A1, B1, C1, nullptr, nullptr, M1, N1, K, lda1, ldb1, ldc1, 0, 0, accumulate, 0, stream);
}

// fp32 residual stream (vlr_llama_cfg::resid_f32; o_proj / down_proj of the decoder layer): C fp32 [M][ldc] = A . B + residual
// fp32 [M][ldr] (NULL: none) - the stream is never rounded.  Same layouts and dispatch as vlr_gemm_bf16.
extern "C" int vlr_gemm_bf16_f32res(int layout, const void* A, const void* B, float* C, const float* residual, int M, int N, int K,
                                    int lda, int ldb, int ldc, int ldr, hipStream_t stream) {
    return gemm_impl_ex(layout, A, B, C, nullptr, residual, M, N, K, lda, ldb, ldc, ldr, 0, 0, 1, 1.0f, 1, stream);
}

static VlrGemmTail* g_gemm_tail = nullptr;
void vlr_internal_set_gemm_tail(VlrGemmTail* t) { g_gemm_tail = t; if (t) { t->used = 0; t->M1 = 0; } }

static int gemm_impl_ex(int layout, const void* A, const void* B, void* C, const void* bias, const void* residual, int M, int N,
                        int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate, int out_f32, float alpha, int res_f32,
                        hipStream_t stream) {
    VlrGemmTail* tail = g_gemm_tail;      // (common.h) consumed by this call whatever path it takes
    g_gemm_tail = nullptr;
    VLR_REQUIRE(!res_f32 || out_f32, "vlr_gemm_bf16: an fp32 residual needs an fp32 output");
    VLR_REQUIRE(layout >= 0 && layout <= 2, "vlr_gemm_bf16: layout must be 0 (NT), 1 (NN) or 2 (TN), got %d", layout);
    VLR_REQUIRE(M > 0 && N > 0 && K > 0, "vlr_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    VLR_REQUIRE(A && B && C, "vlr_gemm_bf16: null operand");
    const bool a_ks = layout == 2, b_ks = layout != 0;
    // 16-byte (k-contiguous) / 8-byte (k-strided) vector loads, 16-byte stores
    VLR_REQUIRE(a_ks ? (lda % 4 == 0 && M % 4 == 0) : (lda % 8 == 0 && K % 8 == 0),
                "vlr_gemm_bf16: A alignment (layout %d lda %d M %d K %d)", layout, lda, M, K);
    VLR_REQUIRE(b_ks ? (ldb % 4 == 0 && N % 4 == 0) : (ldb % 8 == 0 && K % 8 == 0),
                "vlr_gemm_bf16: B alignment (layout %d ldb %d N %d K %d)", layout, ldb, N, K);
    VLR_REQUIRE(out_f32 ? (N % 4 == 0 && ldc % 4 == 0) : (N % 8 == 0 && ldc % 8 == 0),
                "vlr_gemm_bf16: C alignment (N %d ldc %d)", N, ldc);
    VLR_REQUIRE(!residual || ldr % (out_f32 ? 4 : 8) == 0, "vlr_gemm_bf16: residual ld %d", ldr);
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = (const bf16_t*)bias; p.residual = (const bf16_t*)residual;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.act = act; p.accumulate = accumulate; p.out_f32 = out_f32;
    static int gflags = -1;
    if (gflags < 0) { const char* e = getenv("VLR_GEMM_FLAGS"); gflags = e ? atoi(e) : 0; }
    p.flags = gflags;
    p.alpha = alpha;
    p.splitk = 1; p.kchunk = 0; p.part = nullptr;
    p.fuse = 0; p.store_c = 1; p.C2 = nullptr; p.ldc2 = 0; p.pos = nullptr; p.rope_cos = p.rope_sin = nullptr; p.max_pos = 0; p.rope_cols = 0;
    p.f0 = p.f1 = nullptr;
    p.A2 = p.B2 = nullptr; p.lda2 = p.ldb2 = p.K2 = 0; p.seg_b0 = p.seg_b1 = 0x7fffffff; p.drop_key = 0; p.drop_thr = 0; p.drop_ld = 0;
    p.res_f32 = residual ? res_f32 : 0;
    p.sched = 0;
    p.A1 = p.B1 = nullptr; p.C1 = nullptr; p.M1 = p.N1 = p.lda1 = p.ldb1 = p.ldc1 = 0;
    p.groups = 1; p.gA = p.gB = p.gC = 0; p.mask_on = 0; p.mask_seed = 0; p.mask_thr = 0; p.mask_ld = 0;
    p.mask_bits = nullptr; p.gMask = 0; p.drop_bits = nullptr; p.rowskip = nullptr; p.ktlist = nullptr; p.seg_skip = nullptr; p.seg_keep = 0;
    const int pi = vlr_prof_begin(layout, 2.0 * M * N * K, stream);
    // ---- split-K for problems whose output is a handful of tiles but whose reduction is long: the LoRA adapter gradients
    // (TN: dB = dy^T u [out x r], dA = v^T x [r x in], reduction over all tokens) - 32..96 workgroups would leave most CUs idle
    if (layout == 2 && launch_splitk128(layout, p, stream, 2048)) {
        vlr_prof_end(pi, stream);
        return vlr_check_launch("vlr_gemm_bf16(split-K)");
    }
    // ---- tall-and-skinny forward / dgrad products (LoRA: u = x A^T and v = dy B with r = 128 output columns): [12792 x 128] is
    // 100 tiles of 128x128 - 39 % of the CUs - each with the full reduction; split along K they fill the chip
    // (profiles/r02_steady_state_kernel_breakdown_lora_before.txt: 85 ms/step in 1185 such launches)
    if (layout != 2 && N <= 512 && M >= 2048 && ((M + BM - 1) / BM) * ((N + BN - 1) / BN) < 200 && launch_splitk128(layout, p, stream, 2048)) {
        vlr_prof_end(pi, stream);
        return vlr_check_launch("vlr_gemm_bf16(skinny split-K)");
    }
    // ---- wave quantisation: a 256x256-tile grid of T tiles runs ceil(T/256) rounds on the 256 CUs; when the last round is
    // nearly empty (e.g. 12792 x 4096 -> 800 tiles = 3.125 rounds) the last tile-rows are peeled off and run as 128x128
    // tiles (2 workgroups per CU) so the big-tile part is a whole number of rounds.  VLR_GEMM_SPLIT=0 disables.
    static int split_on = -1;
    if (split_on < 0) { const char* e = getenv("VLR_GEMM_SPLIT"); split_on = (e && e[0] == '0') ? 0 : 1; }
    const int tm256 = (M + 255) / 256, tn256 = (N + 255) / 256;
    // (launches that take the per-tile kernel - bias, activation, accumulate, a bf16 residual - cannot use the stream-K tail: peel as before)
    const bool sk_elig = !bias && !accumulate && act == ACT_NONE && (out_f32 || !residual);
    const int peel = split_on ? choose_peel(M, N, tn256, sk_elig ? K : 0, stream) : 0;
    if (peel) {
        const int M1 = (tm256 - peel) * 256;
        GemmParams p1 = p, p2 = p;
        p1.M = M1;
        p2.M = M - M1;
        const size_t esz = out_f32 ? 4 : 2;
        p2.A = layout == 2 ? p.A + M1 : p.A + (size_t)M1 * lda;
        p2.C = (char*)p.C + (size_t)M1 * ldc * esz;
        if (p.residual) p2.residual = (const bf16_t*)((const char*)p.residual + (size_t)M1 * ldr * (p.res_f32 ? 4 : 2));
        if (vlr_gemm256p_try_launch(layout, p1, stream)) {
            const int t2 = ((p2.M + BM - 1) / BM) * ((N + BN - 1) / BN);
            // the peeled rows on the caller's side stream (VlrGemmTail, common.h): behind the main part, beside the caller's next kernel
            hipStream_t ps = stream;
            if (tail) {
                hipEventRecord(tail->fork, stream);
                hipStreamWaitEvent(tail->side, tail->fork, 0);
                ps = tail->side;
            }
            // the peeled rows are few tiles with the full reduction depth (e.g. 504 x 4096 x 22016 = 128 tiles x 688 k-steps):
            // split them along K so that they fill the chip
            if (!launch_splitk128(layout, p2, ps, 4096)) launch128(layout, p2, dim3(t2), ps);
            if (tail) {
                hipEventRecord(tail->done, ps);
                tail->M1 = M1;
                tail->used = 1;
            }
            vlr_prof_end(pi, stream);
            return vlr_check_launch("vlr_gemm_bf16(256+128)");
        }
    }
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (vlr_gemm256p_try_launch(layout, p, stream)) {
        vlr_prof_end(pi, stream);
        return vlr_check_launch("vlr_gemm_bf16(256)");
    }
    // few tiles with a long reduction (small batches against the decoder weights): split along K so that they fill the chip
    if (launch_splitk128(layout, p, stream, 4096)) {
        vlr_prof_end(pi, stream);
        return vlr_check_launch("vlr_gemm_bf16(128 split-K)");
    }
    launch128(layout, p, dim3(tiles), stream);
    vlr_prof_end(pi, stream);
    return vlr_check_launch("vlr_gemm_bf16");
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused forward GEMMs of the decoder layer (include/vlr.h): the elementwise op that follows the projection is applied to the
// fp32 accumulators in the epilogue of the 256x256 continuous-pipeline kernel - one rounding instead of two, and no separate
// pass over the [M][2I] / [M][3H] tensor.  Rows the persistent kernel does not cover (peeled tile rows, small problems) go
// through the plain GEMM + the elementwise kernel.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int vlr_swiglu_fwd(const void* gu, void* act, int M, int I, hipStream_t st);
extern "C" int vlr_rope_heads(void* qkv, const int* pos, const float* cos_t, const float* sin_t, int M, int n_heads, int head_dim,
                              int ld, int max_pos, int backward, hipStream_t st);

static GemmParams fused_params(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc) {
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = nullptr; p.residual = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = 0;
    p.act = 0; p.accumulate = 0; p.out_f32 = 0; p.flags = 0; p.alpha = 1.f; p.splitk = 1; p.kchunk = 0; p.part = nullptr;
    p.fuse = 0; p.store_c = 1; p.C2 = nullptr; p.ldc2 = 0; p.pos = nullptr; p.rope_cos = p.rope_sin = nullptr; p.max_pos = 0; p.rope_cols = 0;
    p.f0 = p.f1 = nullptr;
    p.A2 = p.B2 = nullptr; p.lda2 = p.ldb2 = p.K2 = 0; p.seg_b0 = p.seg_b1 = 0x7fffffff; p.drop_key = 0; p.drop_thr = 0; p.drop_ld = 0; p.res_f32 = 0;
    p.sched = 0;
    p.A1 = p.B1 = nullptr; p.C1 = nullptr; p.M1 = p.N1 = p.lda1 = p.ldb1 = p.ldc1 = 0;
    p.groups = 1; p.gA = p.gB = p.gC = 0; p.mask_on = 0; p.mask_seed = 0; p.mask_thr = 0; p.mask_ld = 0;
    p.mask_bits = nullptr; p.gMask = 0; p.drop_bits = nullptr; p.rowskip = nullptr; p.ktlist = nullptr; p.seg_skip = nullptr; p.seg_keep = 0;
    return p;
}

// LoRA adapter segment of a fused linear group (GemmParams::A2...): u = s * dropout_t(x) A_t^T of the group's targets side by
// side [M][ldu] (r columns each), Bl = lora_B rows [N][r]; the output blocks are [0,b0) [b0,b1) [b1,N).  u == nullptr: none.
struct SegArgs { const void* u; int ldu; const void* Bl; int r; int b0, b1; };
// Row-tile skip of the NEXT adapter-segment GEMM call of this thread (vlr_gemm_seg_rowskip, include/vlr.h): flags [ceil(M / 256)] (1 = the 256-row
// tile runs only the first `keep` K elements of every sub-target's r-wide block - the caller guarantees that the rest of u is ZERO on
// those rows), consumed by the call whatever path it takes (only the persistent segment kernel uses it; every other path computes the
// same sums over the zeros).
static thread_local const unsigned char* g_seg_skip = nullptr;      // per calling thread, as include/vlr.h documents
static thread_local int g_seg_keep = 0;
extern "C" int vlr_gemm_seg_rowskip(const unsigned char* tile_flags, int keep) {
    VLR_REQUIRE(!tile_flags || (keep >= 0 && keep % 64 == 0), "vlr_gemm_seg_rowskip: keep must be a multiple of 64 (0 = the whole segment is skipped), got %d", keep);
    g_seg_skip = tile_flags; g_seg_keep = tile_flags ? keep : 0;
    return VLR_OK;
}
struct SegSkip {     // taken once per public call, handed to the launches of that call (main rows / peeled rows)
    const unsigned char* flags; int keep;
    SegSkip() : flags(g_seg_skip), keep(g_seg_keep) { g_seg_skip = nullptr; g_seg_keep = 0; }
};
static void seg_set(GemmParams& p, const SegArgs* sg, const SegSkip* sk = nullptr, int row0 = 0) {
    if (!sg || !sg->u) return;
    p.A2 = (const bf16_t*)sg->u; p.lda2 = sg->ldu; p.B2 = (const bf16_t*)sg->Bl; p.ldb2 = sg->r; p.K2 = sg->r;
    p.seg_b0 = sg->b0; p.seg_b1 = sg->b1;
    if (sk && sk->flags && sk->keep < sg->r && sg->r % 64 == 0 && row0 % 256 == 0) { p.seg_skip = sk->flags + row0 / 256; p.seg_keep = sk->keep; }
}
// rows [row0, row0 + Mr) that the segment kernel did not take: y[:, block t] += u_t Bl_t^T, one skinny GEMM per block
static int seg_fallback_add(const SegArgs* sg, void* y, int ldy, int row0, int Mr, int N, hipStream_t stream, int y_f32 = 0) {
    if (!sg || !sg->u) return VLR_OK;
    const int bounds[4] = {0, sg->b0 < N ? sg->b0 : N, sg->b1 < N ? sg->b1 : N, N};
    for (int t = 0; t < 3; ++t) {
        const int lo = bounds[t], w = bounds[t + 1] - bounds[t];
        if (w <= 0) continue;
        int rc = gemm_impl(0, (const bf16_t*)sg->u + (size_t)row0 * sg->ldu + (size_t)t * sg->r, (const bf16_t*)sg->Bl + (size_t)lo * sg->r,
                           (char*)y + ((size_t)row0 * ldy + lo) * (y_f32 ? 4 : 2), nullptr, nullptr, Mr, w, sg->r, sg->ldu, sg->r, ldy, 0, 0, 1, y_f32, 1.0f, stream);
        if (rc != VLR_OK) return rc;
    }
    return VLR_OK;
}
static int seg_check(const char* who, const SegArgs* sg) {
    if (!sg || !sg->u) return VLR_OK;
    VLR_REQUIRE(sg->Bl && sg->r > 0 && sg->r % 8 == 0 && sg->ldu % 8 == 0 && sg->ldu >= sg->r, "%s: adapter segment r=%d ldu=%d", who, sg->r, sg->ldu);
    VLR_REQUIRE(sg->b0 > 0 && sg->b1 >= sg->b0 && (sg->b0 % 8 == 0 || sg->b0 == 0x7fffffff) && (sg->b1 % 8 == 0 || sg->b1 == 0x7fffffff),
                "%s: adapter block bounds %d %d", who, sg->b0, sg->b1);
    return VLR_OK;
}

static int gemm_swiglu_impl(const void* x, const void* wgu, void* gu, void* act, int M, int I, int K, int ldx, int store_gu,
                            const SegArgs* sg, hipStream_t stream) {
    const SegSkip skp;       // (taken first: a pending vlr_gemm_seg_rowskip must not survive a call that fails its argument checks)
    VLR_REQUIRE(x && wgu && gu && act, "vlr_gemm_swiglu: null operand");
    VLR_REQUIRE(M > 0 && I > 0 && K > 0 && I % 8 == 0 && K % 8 == 0 && ldx % 8 == 0, "vlr_gemm_swiglu: bad shape M=%d I=%d K=%d ldx=%d", M, I, K, ldx);
    { int rc = seg_check("vlr_gemm_swiglu_lora", sg); if (rc != VLR_OK) return rc; }
    const bool seg = sg && sg->u;
    const int tn = (I + 127) / 128;
    const int peel = choose_peel(M, 2 * I, tn, seg ? 0 : K, stream, true);
    const int tm256 = (M + 255) / 256;
    const int M1 = peel ? (tm256 - peel) * 256 : M;
    GemmParams p = fused_params(x, wgu, gu, M1, 2 * I, K, ldx, K, 2 * I);
    p.fuse = 1; p.store_c = store_gu; p.C2 = act; p.ldc2 = I;
    seg_set(p, sg, &skp, 0);
    int done = 0;
    const int pi_ = vlr_prof_begin(VLR_K_GEMM_NT, 2.0 * M1 * 2 * I * K, stream);    // per-layout totals of the bench line (the fallback rows below are counted by gemm_impl)
    const bool took_ = seg ? vlr_gemm256p_seg_try_launch(p, stream) : vlr_gemm256p_fused_try_launch(p, stream);
    vlr_prof_end(took_ ? pi_ : -1, stream);
    if (took_) {
        int rc = vlr_check_launch("vlr_gemm_swiglu(fused)");
        if (rc != VLR_OK) return rc;
        done = M1;
    }
    if (done > 0 && done < M) {
        // the peeled last tile rows take the SAME fused epilogue (one tile per workgroup): every row of the batch sees the same
        // arithmetic - act from the fp32 accumulators - whatever the batch size
        GemmParams p2 = fused_params((const bf16_t*)x + (size_t)done * ldx, wgu, (bf16_t*)gu + (size_t)done * 2 * I, M - done, 2 * I, K, ldx, K, 2 * I);
        p2.fuse = 1; p2.store_c = store_gu; p2.C2 = (bf16_t*)act + (size_t)done * I; p2.ldc2 = I;
        if (seg) { SegArgs s2 = *sg; s2.u = (const bf16_t*)sg->u + (size_t)done * sg->ldu; seg_set(p2, &s2, &skp, done); }
        const int pj_ = vlr_prof_begin(VLR_K_GEMM_NT, 2.0 * (M - done) * 2 * I * K, stream);
        const bool took2_ = seg ? vlr_gemm256p_seg_try_launch(p2, stream) : vlr_gemm256p_fused_try_launch(p2, stream);
        vlr_prof_end(took2_ ? pj_ : -1, stream);
        if (took2_) {
            int rc = vlr_check_launch("vlr_gemm_swiglu(fused tail)");
            if (rc != VLR_OK) return rc;
            done = M;
        }
    }
    if (done < M) {       // remaining rows: plain GEMM (+ the adapter GEMMs) + SwiGLU kernel
        const bf16_t* xa = (const bf16_t*)x + (size_t)done * ldx;
        bf16_t* gur = (bf16_t*)gu + (size_t)done * 2 * I;
        int rc = gemm_impl(0, xa, wgu, gur, nullptr, nullptr, M - done, 2 * I, K, ldx, K, 2 * I, 0, 0, 0, 0, 1.0f, stream);
        if (rc != VLR_OK) return rc;
        rc = seg_fallback_add(sg, gu, 2 * I, done, M - done, 2 * I, stream);
        if (rc != VLR_OK) return rc;
        return vlr_swiglu_fwd(gur, (bf16_t*)act + (size_t)done * I, M - done, I, stream);
    }
    return VLR_OK;
}
extern "C" int vlr_gemm_swiglu(const void* x, const void* wgu, void* gu, void* act, int M, int I, int K, int ldx, int store_gu,
                               hipStream_t stream) {
    return gemm_swiglu_impl(x, wgu, gu, act, M, I, K, ldx, store_gu, nullptr, stream);
}
// the same with the LoRA adapters of gate_proj / up_proj riding the K loop: u [M][ldu] = s drop(x) A_gate^T | s drop(x) A_up^T
// (r columns each), Bl = [lora_B gate ; lora_B up] [2I][r].  gate | up are always stored (the backward needs them).
extern "C" int vlr_gemm_swiglu_lora(const void* x, const void* wgu, void* gu, void* act, int M, int I, int K, int ldx, const void* u,
                                    int ldu, const void* Bl, int r, hipStream_t stream) {
    if (!(u && Bl)) { const SegSkip drop; (void)drop; }      // a pending vlr_gemm_seg_rowskip does not survive a call that fails here either
    VLR_REQUIRE(u && Bl, "vlr_gemm_swiglu_lora: null adapter operand");
    const SegArgs sg = {u, ldu, Bl, r, I, 0x7fffffff};
    return gemm_swiglu_impl(x, wgu, gu, act, M, I, K, ldx, 1, &sg, stream);
}

static int gemm_qkv_rope_impl(const void* x, const void* wqkv, void* qkv, const int* pos, const float* cos_t, const float* sin_t,
                              int M, int N, int rope_cols, int K, int ldx, int head_dim, int max_pos, const SegArgs* sg, const void* bias,
                              hipStream_t stream) {
    const SegSkip skp;
    VLR_REQUIRE(x && wqkv && qkv && pos && cos_t && sin_t, "vlr_gemm_qkv_rope: null operand");
    VLR_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0 && ldx % 8 == 0, "vlr_gemm_qkv_rope: bad shape M=%d N=%d K=%d", M, N, K);
    VLR_REQUIRE(head_dim % 16 == 0 && rope_cols % head_dim == 0 && rope_cols <= N, "vlr_gemm_qkv_rope: rope_cols %d / head_dim %d / N %d", rope_cols, head_dim, N);
    { int rc = seg_check("vlr_gemm_qkv_rope_lora", sg); if (rc != VLR_OK) return rc; }
    const bool seg = sg && sg->u;
    int done = 0;
    if (head_dim == 128 && (!bias || !((uintptr_t)bias & 7))) {
        const int tn = (N + 255) / 256;
        const int peel = choose_peel(M, N, tn, seg ? 0 : K, stream, true);
        const int tm256 = (M + 255) / 256;
        const int M1 = peel ? (tm256 - peel) * 256 : M;
        GemmParams p = fused_params(x, wqkv, qkv, M1, N, K, ldx, K, N);
        p.fuse = 2; p.pos = pos; p.rope_cos = cos_t; p.rope_sin = sin_t; p.max_pos = max_pos; p.rope_cols = rope_cols;
        p.bias = (const bf16_t*)bias;
        seg_set(p, sg, &skp, 0);
        const int pi_ = vlr_prof_begin(VLR_K_GEMM_NT, 2.0 * M1 * N * K, stream);
        const bool took_ = seg ? vlr_gemm256p_seg_try_launch(p, stream) : vlr_gemm256p_fused_try_launch(p, stream);
        vlr_prof_end(took_ ? pi_ : -1, stream);
        if (took_) {
            int rc = vlr_check_launch("vlr_gemm_qkv_rope(fused)");
            if (rc != VLR_OK) return rc;
            done = M1;
        }
    }
    if (done > 0 && done < M && head_dim == 128 && (!bias || !((uintptr_t)bias & 7))) {       // the peeled rows: same fused epilogue, one tile per workgroup
        GemmParams p2 = fused_params((const bf16_t*)x + (size_t)done * ldx, wqkv, (bf16_t*)qkv + (size_t)done * N, M - done, N, K, ldx, K, N);
        p2.fuse = 2; p2.pos = pos + done; p2.rope_cos = cos_t; p2.rope_sin = sin_t; p2.max_pos = max_pos; p2.rope_cols = rope_cols;
        p2.bias = (const bf16_t*)bias;
        if (seg) { SegArgs s2 = *sg; s2.u = (const bf16_t*)sg->u + (size_t)done * sg->ldu; seg_set(p2, &s2, &skp, done); }
        const int pj_ = vlr_prof_begin(VLR_K_GEMM_NT, 2.0 * (M - done) * N * K, stream);
        const bool took2_ = seg ? vlr_gemm256p_seg_try_launch(p2, stream) : vlr_gemm256p_fused_try_launch(p2, stream);
        vlr_prof_end(took2_ ? pj_ : -1, stream);
        if (took2_) {
            int rc = vlr_check_launch("vlr_gemm_qkv_rope(fused tail)");
            if (rc != VLR_OK) return rc;
            done = M;
        }
    }
    if (done < M) {
        bf16_t* qr = (bf16_t*)qkv + (size_t)done * N;
        int rc = gemm_impl(0, (const bf16_t*)x + (size_t)done * ldx, wqkv, qr, bias, nullptr, M - done, N, K, ldx, K, N, 0, 0, 0, 0, 1.0f, stream);
        if (rc != VLR_OK) return rc;
        rc = seg_fallback_add(sg, qkv, N, done, M - done, N, stream);
        if (rc != VLR_OK) return rc;
        // the q and k column blocks are rope_cols / head_dim consecutive heads
        return vlr_rope_heads(qr, pos + done, cos_t, sin_t, M - done, rope_cols / head_dim, head_dim, N, max_pos, 0, stream);
    }
    return VLR_OK;
}
extern "C" int vlr_gemm_qkv_rope(const void* x, const void* wqkv, void* qkv, const int* pos, const float* cos_t, const float* sin_t,
                                 int M, int N, int rope_cols, int K, int ldx, int head_dim, int max_pos, hipStream_t stream) {
    return gemm_qkv_rope_impl(x, wqkv, qkv, pos, cos_t, sin_t, M, N, rope_cols, K, ldx, head_dim, max_pos, nullptr, nullptr, stream);
}
// the same with a bias [N] on the projection (Qwen c_attn), added to the accumulators before the rotation
extern "C" int vlr_gemm_qkv_rope_bias(const void* x, const void* wqkv, const void* bias, void* qkv, const int* pos, const float* cos_t,
                                      const float* sin_t, int M, int N, int rope_cols, int K, int ldx, int head_dim, int max_pos, hipStream_t stream) {
    return gemm_qkv_rope_impl(x, wqkv, qkv, pos, cos_t, sin_t, M, N, rope_cols, K, ldx, head_dim, max_pos, nullptr, bias, stream);
}
// the same with the LoRA adapters of q_proj / k_proj / v_proj: u [M][ldu] = the three s drop_t(x) A_t^T side by side, Bl = lora_B
// rows of q | k | v [N][r]; q_cols / kv_cols = widths of the q and of the k (= v) output blocks
extern "C" int vlr_gemm_qkv_rope_lora(const void* x, const void* wqkv, const void* bias, void* qkv, const int* pos, const float* cos_t,
                                      const float* sin_t, int M, int N, int rope_cols, int K, int ldx, int head_dim, int max_pos,
                                      const void* u, int ldu, const void* Bl, int r, int q_cols, int kv_cols, hipStream_t stream) {
    const bool widths_ok = (kv_cols == 0 && q_cols == N) || (q_cols > 0 && kv_cols > 0 && q_cols + 2 * kv_cols == N);
    if (!(u && Bl) || !widths_ok) { const SegSkip drop; (void)drop; }      // (as vlr_gemm_swiglu_lora)
    VLR_REQUIRE(u && Bl, "vlr_gemm_qkv_rope_lora: null adapter operand");
    VLR_REQUIRE((kv_cols == 0 && q_cols == N) || (q_cols > 0 && kv_cols > 0 && q_cols + 2 * kv_cols == N),
                "vlr_gemm_qkv_rope_lora: q_cols %d + 2 * kv_cols %d != N %d", q_cols, kv_cols, N);
    const SegArgs sg = {u, ldu, Bl, r, kv_cols ? q_cols : 0x7fffffff, kv_cols ? q_cols + kv_cols : 0x7fffffff};
    return gemm_qkv_rope_impl(x, wqkv, qkv, pos, cos_t, sin_t, M, N, rope_cols, K, ldx, head_dim, max_pos, &sg, bias, stream);
}

extern "C" int vlr_dropout(const void* x, void* out, long n, float p, uint64_t seed, float alpha, int add, hipStream_t st);
// dx [M][in] += scale / (1 - p) * mask_seed .* (v [M][ldv] . A [r][in]): the input-gradient term of one LoRA target with
// lora_dropout (mask index = row * in + col, the mask the forward applied to x).  `scratch` [M][in] is used only when the fused
// kernel does not take the shape (then: product -> scratch, vlr_dropout(add) -> dx).
extern "C" int vlr_gemm_dropout_acc_bits(const void* v, int ldv, const void* A, void* dx, void* scratch, int M, int in, int r, float p,
                                         uint64_t seed, float scale, const void* bits, hipStream_t stream);
extern "C" int vlr_gemm_dropout_acc(const void* v, int ldv, const void* A, void* dx, void* scratch, int M, int in, int r, float p,
                                    uint64_t seed, float scale, hipStream_t stream) {
    return vlr_gemm_dropout_acc_bits(v, ldv, A, dx, scratch, M, in, r, p, seed, scale, nullptr, stream);
}
// bits != NULL: the packed keep mask of vlr_dropout_bits(seed) over [M][in] (read by the 128x128 kernel's epilogue; the other paths hash)
extern "C" int vlr_gemm_dropout_acc_bits(const void* v, int ldv, const void* A, void* dx, void* scratch, int M, int in, int r, float p,
                                         uint64_t seed, float scale, const void* bits, hipStream_t stream) {
    VLR_REQUIRE(v && A && dx && scratch, "vlr_gemm_dropout_acc: null operand");
    VLR_REQUIRE(M > 0 && in > 0 && r > 0 && in % 8 == 0 && r % 8 == 0 && ldv % 8 == 0, "vlr_gemm_dropout_acc: bad shape M=%d in=%d r=%d ldv=%d", M, in, r, ldv);
    VLR_REQUIRE(p >= 0.f && p < 1.f, "vlr_gemm_dropout_acc: 0 <= p < 1 required, got %g", (double)p);
    GemmParams g = fused_params(v, A, dx, M, in, r, ldv, in, in);
    g.fuse = 6; g.accumulate = 1; g.alpha = scale / (1.f - p);
    g.drop_key = vlr_mix64(seed); g.drop_thr = vlr_dropout_thr(p); g.drop_ld = in; g.drop_bits = (const unsigned char*)bits;
    // K is the adapter rank: the launch is one read-modify-write pass over dx with a few MFMAs per tile.  The 256x256 kernel holds one
    // tile per CU and its load - add - store of the 128 KB output tile is exposed (157 us per [12792 x 4096] launch = 1.3 TB/s);
    // the 128x128 kernel keeps several workgroups per CU in flight and adds in fp32 before the one rounding.  VLR_GEMM_DROPACC: 2 (default)
    // 128x128, 1 256x256, 0 product + dropout-accumulate kernel
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("VLR_GEMM_DROPACC"); mode = e ? atoi(e) : 2; }
    if (mode == 2 && in % 8 == 0 && !(((uintptr_t)v | (uintptr_t)A | (uintptr_t)dx) & 15)) {
        const int tiles = ((M + BM - 1) / BM) * ((in + BN - 1) / BN);
        const int pi = vlr_prof_begin(1, 2.0 * M * in * r, stream);
        hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), dim3(tiles), dim3(256), 0, stream, g);
        vlr_prof_end(pi, stream);
        return vlr_check_launch("vlr_gemm_dropout_acc(128)");
    }
    if (mode && vlr_gemm256p_dropacc_try_launch(g, stream)) return vlr_check_launch("vlr_gemm_dropout_acc(fused)");
    int rc = gemm_impl(1, v, A, scratch, nullptr, nullptr, M, in, r, ldv, in, in, 0, 0, 0, 0, 1.0f, stream);
    if (rc != VLR_OK) return rc;
    return vlr_dropout(scratch, dx, (long)M * in, p, seed, scale, 1, stream);
}

// y [M][ldy] = x W^T + u Bl^T (+ residual): one adapted linear (o_proj, down_proj) with its LoRA adapter riding the K loop
static int gemm_lora_impl(const void* x, int ldx, const void* W, void* y, int ldy, const void* residual, int ldr, int M, int N, int K,
                          const void* u, int ldu, const void* Bl, int r, int f32, hipStream_t stream);
extern "C" int vlr_gemm_lora(const void* x, int ldx, const void* W, void* y, int ldy, const void* residual, int ldr, int M, int N, int K,
                             const void* u, int ldu, const void* Bl, int r, hipStream_t stream) {
    return gemm_lora_impl(x, ldx, W, y, ldy, residual, ldr, M, N, K, u, ldu, Bl, r, 0, stream);
}
// the same on the fp32 residual stream: y fp32 [M][ldy] = x W^T + u Bl^T + residual fp32 [M][ldr]
extern "C" int vlr_gemm_lora_f32res(const void* x, int ldx, const void* W, float* y, int ldy, const float* residual, int ldr, int M, int N,
                                    int K, const void* u, int ldu, const void* Bl, int r, hipStream_t stream) {
    return gemm_lora_impl(x, ldx, W, y, ldy, residual, ldr, M, N, K, u, ldu, Bl, r, 1, stream);
}
static int gemm_lora_impl(const void* x, int ldx, const void* W, void* y, int ldy, const void* residual, int ldr, int M, int N, int K,
                          const void* u, int ldu, const void* Bl, int r, int f32, hipStream_t stream) {
    const SegSkip skp;
    VLR_REQUIRE(x && W && y && u && Bl, "vlr_gemm_lora: null operand");
    VLR_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "vlr_gemm_lora: bad shape M=%d N=%d K=%d", M, N, K);
    const SegArgs sg = {u, ldu, Bl, r, 0x7fffffff, 0x7fffffff};
    VLR_REQUIRE(r > 0 && r % 8 == 0 && ldu % 8 == 0 && ldu >= r, "vlr_gemm_lora: adapter segment r=%d ldu=%d", r, ldu);
    const int tn = (N + 255) / 256;
    const int peel = choose_peel(M, N, tn);          // (adapter-segment kernels run plain rounds)
    const int tm256 = (M + 255) / 256;
    const int M1 = peel ? (tm256 - peel) * 256 : M;
    GemmParams p = fused_params(x, W, y, M1, N, K, ldx, K, ldy);
    p.residual = (const bf16_t*)residual; p.ldr = ldr;
    p.out_f32 = f32; p.res_f32 = (f32 && residual) ? 1 : 0;
    seg_set(p, &sg, &skp, 0);
    int done = 0;
    const int pi_ = vlr_prof_begin(VLR_K_GEMM_NT, 2.0 * M1 * N * K, stream);
    const bool took_ = vlr_gemm256p_seg_try_launch(p, stream);
    vlr_prof_end(took_ ? pi_ : -1, stream);
    if (took_) {
        int rc = vlr_check_launch("vlr_gemm_lora(fused)");
        if (rc != VLR_OK) return rc;
        done = M1;
    }
    if (done < M) {
        const size_t esz = f32 ? 4 : 2;
        int rc = gemm_impl_ex(0, (const bf16_t*)x + (size_t)done * ldx, W, (char*)y + (size_t)done * ldy * esz, nullptr,
                              residual ? (const char*)residual + (size_t)done * ldr * esz : nullptr, M - done, N, K, ldx, K, ldy, ldr, 0, 0, f32, 1.0f,
                              f32, stream);
        if (rc != VLR_OK) return rc;
        return seg_fallback_add(&sg, y, ldy, done, M - done, N, stream, f32);
    }
    return VLR_OK;
}

extern "C" int vlr_swiglu_bwd(void* gu_inout, const void* dact, int M, int I, hipStream_t st);

// d gate | d up (in place in gu [M][2I]) = SwiGLU'(gate, up) * (dy [M][H] . wdown [H][I]): the dgrad of the down projection with
// the SwiGLU backward in its epilogue - d act [M][I] is never written (dact_ws is only used for rows / shapes that fall back)
static int gemm_swiglu_bwd_impl(const void* dy, const void* wdown, void* gu, void* dact_ws, const void* dact_add, int M, int I, int H,
                                hipStream_t stream);
extern "C" int vlr_gemm_swiglu_bwd(const void* dy, const void* wdown, void* gu, void* dact_ws, int M, int I, int H, hipStream_t stream) {
    return gemm_swiglu_bwd_impl(dy, wdown, gu, dact_ws, nullptr, M, I, H, stream);
}
// the same with an addend on d act before the SwiGLU backward: d act = dy . wdown + dact_add [M][I] (bf16) - the LoRA adapter term of
// down_proj, which must reach d act first.  dact_add may be the dact_ws buffer itself.
extern "C" int vlr_gemm_swiglu_bwd_add(const void* dy, const void* wdown, void* gu, void* dact_ws, const void* dact_add, int M, int I,
                                       int H, hipStream_t stream) {
    VLR_REQUIRE(dact_add, "vlr_gemm_swiglu_bwd_add: null addend");
    return gemm_swiglu_bwd_impl(dy, wdown, gu, dact_ws, dact_add, M, I, H, stream);
}
static int gemm_swiglu_bwd_impl(const void* dy, const void* wdown, void* gu, void* dact_ws, const void* dact_add, int M, int I, int H,
                                hipStream_t stream) {
    VLR_REQUIRE(dy && wdown && gu && dact_ws, "vlr_gemm_swiglu_bwd: null operand");
    VLR_REQUIRE(M > 0 && I > 0 && H > 0 && I % 8 == 0 && H % 8 == 0, "vlr_gemm_swiglu_bwd: bad shape M=%d I=%d H=%d", M, I, H);
    const int tn = (I + 255) / 256;
    const int peel = choose_peel(M, I, tn, H, stream, true);
    const int tm256 = (M + 255) / 256;
    const int M1 = peel ? (tm256 - peel) * 256 : M;
    GemmParams p = fused_params(dy, wdown, dact_ws, M1, I, H, H, I, I);
    p.fuse = 3; p.C2 = gu; p.ldc2 = 2 * I;
    p.residual = (const bf16_t*)dact_add; p.ldr = I;
    int done = 0;
    const int pi_ = vlr_prof_begin(VLR_K_GEMM_NN, 2.0 * M1 * I * H, stream);
    const bool took_ = vlr_gemm256p_swiglu_bwd_try_launch(p, stream);
    vlr_prof_end(took_ ? pi_ : -1, stream);
    if (took_) {
        int rc = vlr_check_launch("vlr_gemm_swiglu_bwd(fused)");
        if (rc != VLR_OK) return rc;
        done = M1;
    }
    if (done > 0 && done < M) {       // the peeled rows: same fused epilogue, one tile per workgroup
        GemmParams p2 = fused_params((const bf16_t*)dy + (size_t)done * H, wdown, (bf16_t*)dact_ws + (size_t)done * I, M - done, I, H, H, I, I);
        p2.fuse = 3; p2.C2 = (bf16_t*)gu + (size_t)done * 2 * I; p2.ldc2 = 2 * I;
        p2.residual = dact_add ? (const bf16_t*)dact_add + (size_t)done * I : nullptr; p2.ldr = I;
        const int pj_ = vlr_prof_begin(VLR_K_GEMM_NN, 2.0 * (M - done) * I * H, stream);
        const bool took2_ = vlr_gemm256p_swiglu_bwd_try_launch(p2, stream);
        vlr_prof_end(took2_ ? pj_ : -1, stream);
        if (took2_) {
            int rc = vlr_check_launch("vlr_gemm_swiglu_bwd(fused tail)");
            if (rc != VLR_OK) return rc;
            done = M;
        }
    }
    if (done < M) {
        const bf16_t* dyr = (const bf16_t*)dy + (size_t)done * H;
        bf16_t* da = (bf16_t*)dact_ws + (size_t)done * I;
        int rc = gemm_impl(1, dyr, wdown, da, nullptr, dact_add ? (const bf16_t*)dact_add + (size_t)done * I : nullptr, M - done, I, H, H, I, I, I, 0, 0, 0, 1.0f, stream);
        if (rc != VLR_OK) return rc;
        return vlr_swiglu_bwd((bf16_t*)gu + (size_t)done * 2 * I, da, M - done, I, stream);
    }
    return VLR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused lm-head + log-softmax pick (VLDPOTrainer.get_batch_logps on the response rows, reference base/trainer.py:148-188): the
// [R][V] logits never reach HBM.  Forward: GEMM epilogue -> per-wave (max, sum exp) partials + the target logit, folded by
// lmhead_fold_kernel into lse and tok_logp.  Backward: the GEMM is recomputed and its epilogue writes d logits (bf16) directly.
// Shapes the persistent kernel does not take go through the fp32 logits buffer (`logits_ws`, [R][V] fp32) as before.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int vlr_logp_rows(const float* logits, const int* row_idx, const int* tgt, int R, int V, long ld, float* tok_logp, float* lse,
                             hipStream_t st);
extern "C" int vlr_dlogits_rows(const float* logits, const int* tgt, const float* lse, const int* seq_off, int nseq, const float* dlogps,
                                int average, int R, int V, long ld, void* dlogits, long ldd, hipStream_t st);

__global__ __launch_bounds__(64) void lmhead_fold_kernel(const float* __restrict__ parts, int nparts, const float* __restrict__ tok_raw,
                                                         float* __restrict__ tok_logp, float* __restrict__ lse) {
    const int r = blockIdx.x, l = threadIdx.x;
    const float* pr = parts + (size_t)r * nparts * 2;
    float m = -INFINITY;
    for (int i = l; i < nparts; i += 64) m = fmaxf(m, pr[2 * i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = l; i < nparts; i += 64) {
        const float mi = pr[2 * i];
        if (mi > -INFINITY) s += pr[2 * i + 1] * __expf(mi - m);
    }
    s = wave_sum(s);                       // fixed order: deterministic
    if (l == 0) {
        const float z = m + logf(s);
        lse[r] = z;
        tok_logp[r] = tok_raw[r] - z;
    }
}
// g[r] = dlogps[sequence of row r] (/ rows of the sequence when averaging)
__global__ void lmhead_rowcoef_kernel(const int* __restrict__ seq_off, int nseq, const float* __restrict__ dlogps, int average, int R,
                                      float* __restrict__ g) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int b = 0;
    while (b + 1 < nseq && seq_off[b + 1] <= r) ++b;
    float v = dlogps[b];
    if (average) v /= (float)(seq_off[b + 1] - seq_off[b]);
    g[r] = v;
}

extern "C" long vlr_lmhead_workspace_bytes(int R, int V) { return ((long)R * vlr_gemm256p_lmhead_parts(V) * 2 + 2L * R) * 4; }

// 1 if the fused kernels take (R, V, H): the caller then needs no fp32 logits buffer
extern "C" int vlr_lmhead_is_fused(int R, int V, int H) {
    const char* e = getenv("VLR_GEMM_FUSE");
    if (e && !((atoi(e) >> 3) & 1)) return 0;
    const long ntiles = (long)((R + 255) / 256) * ((V + 255) / 256);
    return ntiles > vlr_compute_cus() && H >= 256 && H % 8 == 0 && V % 8 == 0;     // the predicate of vlr_gemm256p_lmhead_try_launch
}

extern "C" int vlr_lmhead_logps_fwd(const void* hg, const void* w_lm, const int* tgt, float* tok_logp, float* lse, void* workspace,
                                    float* logits_ws, int R, int V, int H, hipStream_t stream) {
    VLR_REQUIRE(hg && w_lm && tgt && tok_logp && lse, "vlr_lmhead_logps_fwd: null operand");
    VLR_REQUIRE(R > 0 && V % 8 == 0 && H % 8 == 0, "vlr_lmhead_logps_fwd: bad shape R=%d V=%d H=%d", R, V, H);
    if (workspace && vlr_lmhead_is_fused(R, V, H)) {
        const int nparts = vlr_gemm256p_lmhead_parts(V);
        float* parts = (float*)workspace;
        float* tok_raw = parts + (size_t)R * nparts * 2;
        GemmParams p = fused_params(hg, w_lm, nullptr, R, V, H, H, H, V);
        p.fuse = 4; p.pos = tgt; p.C2 = parts; p.f1 = tok_raw;
        const int pi_ = vlr_prof_begin(VLR_K_GEMM_NT, 2.0 * R * V * H, stream);
        const bool took_ = vlr_gemm256p_lmhead_try_launch(p, stream);
        vlr_prof_end(took_ ? pi_ : -1, stream);
        if (took_) {
            hipLaunchKernelGGL(lmhead_fold_kernel, dim3(R), dim3(64), 0, stream, (const float*)parts, nparts, (const float*)tok_raw, tok_logp, lse);
            return vlr_check_launch("vlr_lmhead_logps_fwd(fused)");
        }
    }
    VLR_REQUIRE(logits_ws, "vlr_lmhead_logps_fwd: this shape needs the fp32 logits buffer [R][V]");
    int rc = gemm_impl(0, hg, w_lm, logits_ws, nullptr, nullptr, R, V, H, H, H, V, 0, 0, 0, 1, 1.0f, stream);
    if (rc != VLR_OK) return rc;
    return vlr_logp_rows(logits_ws, nullptr, tgt, R, V, V, tok_logp, lse, stream);
}

extern "C" int vlr_lmhead_logps_bwd(const void* hg, const void* w_lm, const int* tgt, const float* lse, const int* seq_off, int nseq,
                                    const float* dlogps, int average, void* dlogits, void* workspace, float* logits_ws, int R, int V,
                                    int H, hipStream_t stream) {
    VLR_REQUIRE(hg && w_lm && tgt && lse && seq_off && dlogps && dlogits, "vlr_lmhead_logps_bwd: null operand");
    VLR_REQUIRE(R > 0 && V % 8 == 0 && H % 8 == 0 && nseq > 0, "vlr_lmhead_logps_bwd: bad shape R=%d V=%d H=%d", R, V, H);
    if (workspace && vlr_lmhead_is_fused(R, V, H)) {
        float* coef = (float*)workspace + (size_t)R * vlr_gemm256p_lmhead_parts(V) * 2 + R;
        hipLaunchKernelGGL(lmhead_rowcoef_kernel, dim3((R + 255) / 256), dim3(256), 0, stream, seq_off, nseq, dlogps, average, R, coef);
        GemmParams p = fused_params(hg, w_lm, dlogits, R, V, H, H, H, V);
        p.fuse = 5; p.pos = tgt; p.f0 = const_cast<float*>(lse); p.f1 = coef;
        const int pi_ = vlr_prof_begin(VLR_K_GEMM_NT, 2.0 * R * V * H, stream);
        const bool took_ = vlr_gemm256p_lmhead_try_launch(p, stream);
        vlr_prof_end(took_ ? pi_ : -1, stream);
        if (took_) return vlr_check_launch("vlr_lmhead_logps_bwd(fused)");
    }
    VLR_REQUIRE(logits_ws, "vlr_lmhead_logps_bwd: this shape needs the fp32 logits buffer [R][V]");
    int rc = gemm_impl(0, hg, w_lm, logits_ws, nullptr, nullptr, R, V, H, H, H, V, 0, 0, 0, 1, 1.0f, stream);
    if (rc != VLR_OK) return rc;
    return vlr_dlogits_rows(logits_ws, tgt, lse, seq_off, nseq, dlogps, average, R, V, V, dlogits, V, stream);
}
