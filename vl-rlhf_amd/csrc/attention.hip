// Fused attention for gfx950 (MI355X): forward (causal + key-padding mask for the LLaMA decoder, D=128; full
// attention for the CLIP ViT, D=64) and backward (D=128).  No S x S matrix is ever written to HBM.
//
// Tiling is built around v_mfma_f32_32x32x16_bf16 with the SWAPPED product S^T = K . Q^T so that a query's row of
// scores lives in ONE lane pair (l, l^32): softmax statistics are lane-local + one cross-lane exchange, and the
// bf16 P (or dS) fragment that feeds the second MFMA is built from the accumulator registers in place - the MFMA's
// k-slot permutation is absorbed by storing the other operand (V^T / K^T / Q^T / dO^T) in LDS in the same permuted
// order.  K-type tiles are XOR-swizzled 16-byte chunks (conflict-free ds_read_b128); transposed tiles are produced
// in registers (4x4 v_perm transposes) while staging.
//
// Layout: q, k, v are column blocks of one fused [tokens][3H] buffer (row stride ld); token row = b*S + s; head h
// owns columns [h*D, (h+1)*D).  lse is kept in the log2 domain: L2 = m + log2(l) with scores pre-multiplied by
// scale*log2(e), so P = exp2(s*scale*log2e - L2).  Fully masked query rows give O = 0, L2 = +inf (P == 0 in bwd).
#include "common.h"

#define KV_TILE 64
#define LOG2E 1.4426950408889634f

template <int D>
__device__ __forceinline__ int kt_off(int row, int chunk) {   // [64][D] k-contiguous tile, 16-byte chunk swizzle
    if constexpr (D == 128) return row * 256 + ((chunk ^ (row & 15)) << 4);
    else return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
// transposed tile [D][64 cols]: 128-byte rows of 8 chunks; chunk = (16-col block)*2 + g holds cols {4g..4g+3, 8+4g..}
__device__ __forceinline__ int tt_off(int drow, int chunk) { return drow * 128 + ((chunk ^ ((drow >> 1) & 7)) << 4); }

// ---- tile staging, split in two so the global loads of tile t+1 fly under the MFMAs of tile t --------------------------
// A 64-row x D tile is held as 8-byte quads: thread -> (d quad dq = t % (D/4), row quads j = t/(D/4) + JPI*i): rows 4j..4j+3,
// columns 4dq..4dq+3.  The same registers feed the row-major (k-contiguous, swizzled) image and the transposed image.
template <int D>
struct TileRegs {
    static constexpr int DQ = D / 4;
    static constexpr int JPI = 256 / DQ;
    static constexpr int NI = 16 / JPI;
    u32x2 r[NI][4];
};
template <int D>
__device__ __forceinline__ void tile_load(TileRegs<D>& tr, const bf16_t* __restrict__ src, int ld, int row0, int nrows, int t) {
    const int dq = t % TileRegs<D>::DQ;
#pragma unroll
    for (int i = 0; i < TileRegs<D>::NI; ++i) {
        const int j = t / TileRegs<D>::DQ + TileRegs<D>::JPI * i;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = row0 + 4 * j + rr;
            u32x2 v = {0u, 0u};
            if (row < nrows) v = *reinterpret_cast<const u32x2*>(src + (size_t)row * ld + dq * 4);
            tr.r[i][rr] = v;
        }
    }
}
// row-major [64][D] image, 16-byte chunk swizzle (kt_off); each thread writes its 8-byte halves
template <int D>
__device__ __forceinline__ void tile_store_rows(const TileRegs<D>& tr, char* lds, int t) {
    const int dq = t % TileRegs<D>::DQ;
#pragma unroll
    for (int i = 0; i < TileRegs<D>::NI; ++i) {
        const int j = t / TileRegs<D>::DQ + TileRegs<D>::JPI * i;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            *reinterpret_cast<u32x2*>(lds + kt_off<D>(4 * j + rr, dq >> 1) + (dq & 1) * 8) = tr.r[i][rr];
    }
}
// transposed [D][64] image (permuted 16-col blocks, tt_off): 4x4 register transposes with v_perm
template <int D>
__device__ __forceinline__ void tile_store_transposed(const TileRegs<D>& tr, char* lds, int t) {
    const int dq = t % TileRegs<D>::DQ;
#pragma unroll
    for (int i = 0; i < TileRegs<D>::NI; ++i) {
        const int j = t / TileRegs<D>::DQ + TileRegs<D>::JPI * i;
        const int ks = j >> 2, qi = j & 3;
        const int chunk = ks * 2 + (qi & 1);
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const uint32_t sel = (dd & 1) ? 0x07060302u : 0x05040100u;
            u32x2 o;
            o[0] = __builtin_amdgcn_perm(tr.r[i][1][dd >> 1], tr.r[i][0][dd >> 1], sel);
            o[1] = __builtin_amdgcn_perm(tr.r[i][3][dd >> 1], tr.r[i][2][dd >> 1], sel);
            *reinterpret_cast<u32x2*>(lds + tt_off(dq * 4 + dd, chunk) + (qi >> 1) * 8) = o;
        }
    }
}

__device__ __forceinline__ int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

__device__ __forceinline__ bf16x8 pack_frag(const f32x16& s, int h) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16(s[8 * h + 2 * i], s[8 * h + 2 * i + 1]);
    return __builtin_bit_cast(bf16x8, w);
}

// write a transposed accumulator tile set acc[db][r] = X^T[d = db*32 + crow(r,g)][row = lane&31] to X[row][d]
template <int D>
__device__ __forceinline__ void write_rows(const f32x16* acc, float mul, bf16_t* __restrict__ dst_row, int g) {
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            u32x2 w;
            w[0] = pack_bf16(acc[db][4 * rq] * mul, acc[db][4 * rq + 1] * mul);
            w[1] = pack_bf16(acc[db][4 * rq + 2] * mul, acc[db][4 * rq + 3] * mul);
            *reinterpret_cast<u32x2*>(dst_row + db * 32 + 8 * rq + 4 * g) = w;
        }
}

// ------------------------------------------------------------------------------------------------------------
// Forward.  grid (ceil(S/128), heads, batch), 256 threads; wave w owns queries [q0 + 32w, q0 + 32w + 32).
// ------------------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, int ld, bf16_t* __restrict__ o,
                                                       int ldo, float* __restrict__ lse, const int* __restrict__ kmask,
                                                       int S, int Sp, float scale_log2) {
    __shared__ __attribute__((aligned(16))) char smem[KV_TILE * D * 2 * 2 + KV_TILE * 4];
    char* k_lds = smem;
    char* vt_lds = smem + KV_TILE * D * 2;
    float* bias_lds = reinterpret_cast<float*>(smem + KV_TILE * D * 4);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z, nh = gridDim.y;
    const size_t tok0 = (size_t)b * S;
    const bf16_t* qh = q + tok0 * ld + head * D;
    const bf16_t* kh = k + tok0 * ld + head * D;
    const bf16_t* vh = v + tok0 * ld + head * D;
    const int qblk = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;   // causal: heaviest blocks first
    const int qw0 = qblk * 128 + wave * 32;
    const int qi = qw0 + (lane & 31);
    const int qrow = qi < S ? qi : S - 1;

    bf16x8 qf[D / 16];
#pragma unroll
    for (int st = 0; st < D / 16; ++st)
        qf[st] = *reinterpret_cast<const bf16x8*>(qh + (size_t)qrow * ld + 16 * st + 8 * g);

    f32x16 acc[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int q_end = min(S, qblk * 128 + 128);
    const int nkv = CAUSAL ? (q_end + KV_TILE - 1) / KV_TILE : (S + KV_TILE - 1) / KV_TILE;
    TileRegs<D> kreg, vreg;
    tile_load<D>(kreg, kh, ld, 0, S, t);
    tile_load<D>(vreg, vh, ld, 0, S, t);
    for (int it = 0; it < nkv; ++it) {
        const int k0 = it * KV_TILE;
        __syncthreads();                                   // every wave is done reading the previous tile
        tile_store_rows<D>(kreg, k_lds, t);
        tile_store_transposed<D>(vreg, vt_lds, t);
        if (t < KV_TILE) {
            const int key = k0 + t;
            bias_lds[t] = (key < S && (!kmask || kmask[tok0 + key] != 0)) ? 0.f : -INFINITY;
        }
        __syncthreads();
        if (it + 1 < nkv) {                                // next tile's global loads fly under this tile's MFMAs
            tile_load<D>(kreg, kh, ld, k0 + KV_TILE, S, t);
            tile_load<D>(vreg, vh, ld, k0 + KV_TILE, S, t);
        }
        if (CAUSAL && k0 > qw0 + 31) continue;   // wave-uniform: whole tile is in this wave's future

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int st = 0; st < D / 16; ++st) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + kt_off<D>(kb * 32 + (lane & 31), 2 * st + g));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s[kb], 0, 0, 0);
            }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + kb * 32 + 8 * rq + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e;
                    const int key = k0 + kb * 32 + crow(r, g);
                    float val = s[kb][r] * scale_log2 + bb[e];
                    if (CAUSAL && key > qi) val = -INFINITY;
                    s[kb][r] = val;
                    mloc = fmaxf(mloc, val);
                }
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m, mloc);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m - m_use);
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f(s[kb][r] - m_use);
                s[kb][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 pf = pack_frag(s[ks >> 1], ks & 1);
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vt_lds + tt_off(db * 32 + (lane & 31), ks * 2 + g));
                acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, acc[db], 0, 0, 0);
            }
        }
    }
    if (qi < S) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        write_rows<D>(acc, inv, o + (tok0 + qi) * (size_t)ldo + head * D, g);
    }
    // lse rows are padded to Sp (multiple of 64) and the tail holds +inf so the backward's P is exactly 0 there
    if (lse && g == 0 && qi < Sp) lse[((size_t)b * nh + head) * Sp + qi] = (qi < S && l > 0.f) ? m + log2f(l) : INFINITY;
}

// ------------------------------------------------------------------------------------------------------------
// delta[b][h][s] = sum_d dO[s][h*128+d] * O[s][h*128+d]   (D = 128)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                         int ldo, float* __restrict__ delta, int S, int Sp, int nh) {
    const size_t row = blockIdx.x;   // token row b*S+s
    const int b = (int)(row / S), s = (int)(row % S);
    for (int c = threadIdx.x * 8; c < nh * 128; c += 256 * 8) {
        float a[8], d[8];
        unpack8(*reinterpret_cast<const u32x4*>(dout + row * ldo + c), a);
        unpack8(*reinterpret_cast<const u32x4*>(out + row * ldo + c), d);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += a[e] * d[e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if ((threadIdx.x & 15) == 0) delta[((size_t)b * nh + c / 128) * Sp + s] = sum;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Backward, dQ:  grid (ceil(S/128), heads, batch); wave owns 32 queries, loops over KV tiles.
//   S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta),  dQ^T += K^T dS^T * scale
// ------------------------------------------------------------------------------------------------------------
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                          const bf16_t* __restrict__ v, int ld,
                                                          const bf16_t* __restrict__ dout, int ldo,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          const int* __restrict__ kmask, bf16_t* __restrict__ dq, int lddq,
                                                          int S, int Sp, float scale) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(16))) char smem[KV_TILE * D * 2 * 3 + KV_TILE * 4];
    char* k_lds = smem;
    char* v_lds = smem + KV_TILE * D * 2;
    char* kt_lds = smem + KV_TILE * D * 4;
    float* bias_lds = reinterpret_cast<float*>(smem + KV_TILE * D * 6);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z, nh = gridDim.y;
    const size_t tok0 = (size_t)b * S;
    const bf16_t* qh = q + tok0 * ld + head * D;
    const bf16_t* kh = k + tok0 * ld + head * D;
    const bf16_t* vh = v + tok0 * ld + head * D;
    const bf16_t* doh = dout + tok0 * ldo + head * D;
    const int qblk = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int qw0 = qblk * 128 + wave * 32;
    const int qi = qw0 + (lane & 31);
    const int qrow = qi < S ? qi : S - 1;
    const float scale_log2 = scale * LOG2E;

    bf16x8 qf[8], dof[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        qf[st] = *reinterpret_cast<const bf16x8*>(qh + (size_t)qrow * ld + 16 * st + 8 * g);
        dof[st] = *reinterpret_cast<const bf16x8*>(doh + (size_t)qrow * ldo + 16 * st + 8 * g);
    }
    const float L2 = qi < S ? lse[((size_t)b * nh + head) * Sp + qrow] : INFINITY;
    const float dl = delta[((size_t)b * nh + head) * Sp + qrow];

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int q_end = min(S, qblk * 128 + 128);
    const int nkv = CAUSAL ? (q_end + KV_TILE - 1) / KV_TILE : (S + KV_TILE - 1) / KV_TILE;
    TileRegs<D> kreg, vreg;
    tile_load<D>(kreg, kh, ld, 0, S, t);
    tile_load<D>(vreg, vh, ld, 0, S, t);
    for (int it = 0; it < nkv; ++it) {
        const int k0 = it * KV_TILE;
        __syncthreads();
        tile_store_rows<D>(kreg, k_lds, t);
        tile_store_transposed<D>(kreg, kt_lds, t);
        tile_store_rows<D>(vreg, v_lds, t);
        if (t < KV_TILE) {
            const int key = k0 + t;
            bias_lds[t] = (key < S && (!kmask || kmask[tok0 + key] != 0)) ? 0.f : -INFINITY;
        }
        __syncthreads();
        if (it + 1 < nkv) {
            tile_load<D>(kreg, kh, ld, k0 + KV_TILE, S, t);
            tile_load<D>(vreg, vh, ld, k0 + KV_TILE, S, t);
        }
        if (CAUSAL && k0 > qw0 + 31) continue;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int off = kt_off<D>(kb * 32 + (lane & 31), 2 * st + g);
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + off);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(v_lds + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[st], dp, 0, 0, 0);
            }
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + kb * 32 + 8 * rq + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e;
                    const int key = k0 + kb * 32 + crow(r, g);
                    float val = s[r] * scale_log2 + bb[e];
                    if (CAUSAL && key > qi) val = -INFINITY;
                    const float p = exp2f(val - L2);
                    s[r] = p > 0.f ? p * (dp[r] - dl) * scale : 0.f;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 dsf = pack_frag(s, h);
                const int ks = kb * 2 + h;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const bf16x8 ktf = *reinterpret_cast<const bf16x8*>(kt_lds + tt_off(db * 32 + (lane & 31), ks * 2 + g));
                    acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, acc[db], 0, 0, 0);
                }
            }
        }
    }
    if (qi < S) write_rows<D>(acc, 1.f, dq + (tok0 + qi) * (size_t)lddq + head * D, g);
}

// ------------------------------------------------------------------------------------------------------------
// Backward, dK and dV: grid (ceil(S/128), heads, batch); wave owns 32 keys, loops over 64-query tiles.
//   S = Q K^T, dP = dO V^T (key lane-local),  dV^T += dO^T P,  dK^T += Q^T dS * scale
// ------------------------------------------------------------------------------------------------------------
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, int ld,
                                                           const bf16_t* __restrict__ dout, int ldo,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           const int* __restrict__ kmask, bf16_t* __restrict__ dk,
                                                           bf16_t* __restrict__ dv, int lddkv, int S, int Sp, float scale) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(16))) char smem[KV_TILE * D * 2 * 4];
    char* q_lds = smem;
    char* do_lds = smem + KV_TILE * D * 2;
    char* qt_lds = smem + KV_TILE * D * 4;
    char* dot_lds = smem + KV_TILE * D * 6;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z, nh = gridDim.y;
    const size_t tok0 = (size_t)b * S;
    const bf16_t* qh = q + tok0 * ld + head * D;
    const bf16_t* kh = k + tok0 * ld + head * D;
    const bf16_t* vh = v + tok0 * ld + head * D;
    const bf16_t* doh = dout + tok0 * ldo + head * D;
    const float* lse_h = lse + ((size_t)b * nh + head) * Sp;   // rows padded to Sp: 16-byte aligned, tail = +inf
    const float* dl_h = delta + ((size_t)b * nh + head) * Sp;
    const int kw0 = blockIdx.x * 128 + wave * 32;
    const int ki = kw0 + (lane & 31);
    const int krow = ki < S ? ki : S - 1;
    const bool key_ok = ki < S && (!kmask || kmask[tok0 + krow] != 0);
    const float scale_log2 = scale * LOG2E;

    bf16x8 kf[8], vf[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        kf[st] = *reinterpret_cast<const bf16x8*>(kh + (size_t)krow * ld + 16 * st + 8 * g);
        vf[st] = *reinterpret_cast<const bf16x8*>(vh + (size_t)krow * ld + 16 * st + 8 * g);
    }
    f32x16 adk[4], adv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { adk[i][r] = 0.f; adv[i][r] = 0.f; }

    const int q_start = CAUSAL ? ((int)blockIdx.x * 128 / KV_TILE) * KV_TILE : 0;
    TileRegs<D> qreg, doreg;
    tile_load<D>(qreg, qh, ld, q_start, S, t);
    tile_load<D>(doreg, doh, ldo, q_start, S, t);
    for (int q0 = q_start; q0 < S; q0 += KV_TILE) {
        __syncthreads();
        tile_store_rows<D>(qreg, q_lds, t);
        tile_store_transposed<D>(qreg, qt_lds, t);
        tile_store_rows<D>(doreg, do_lds, t);
        tile_store_transposed<D>(doreg, dot_lds, t);
        __syncthreads();
        if (q0 + KV_TILE < S) {
            tile_load<D>(qreg, qh, ld, q0 + KV_TILE, S, t);
            tile_load<D>(doreg, doh, ldo, q0 + KV_TILE, S, t);
        }
        if (CAUSAL && q0 + KV_TILE - 1 < kw0) continue;   // every query of the tile precedes this wave's keys
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int off = kt_off<D>(qb * 32 + (lane & 31), 2 * st + g);
                const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(q_lds + off);
                const bf16x8 dofr = *reinterpret_cast<const bf16x8*>(do_lds + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[st], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr, vf[st], dp, 0, 0, 0);
            }
            f32x16 p;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 l2 = *reinterpret_cast<const f32x4*>(lse_h + q0 + qb * 32 + 8 * rq + 4 * g);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(dl_h + q0 + qb * 32 + 8 * rq + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e;
                    const int qq = q0 + qb * 32 + crow(r, g);
                    float pv = exp2f(s[r] * scale_log2 - l2[e]);
                    if (!key_ok || (CAUSAL && ki > qq)) pv = 0.f;
                    p[r] = pv;
                    s[r] = pv > 0.f ? pv * (dp[r] - dl[e]) * scale : 0.f;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 pf = pack_frag(p, h);
                const bf16x8 dsf = pack_frag(s, h);
                const int ks = qb * 2 + h;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const int off = tt_off(db * 32 + (lane & 31), ks * 2 + g);
                    const bf16x8 dotf = *reinterpret_cast<const bf16x8*>(dot_lds + off);
                    const bf16x8 qtf = *reinterpret_cast<const bf16x8*>(qt_lds + off);
                    adv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, adv[db], 0, 0, 0);
                    adk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, adk[db], 0, 0, 0);
                }
            }
        }
    }
    if (ki < S) {
        write_rows<D>(adk, 1.f, dk + (tok0 + ki) * (size_t)lddkv + head * D, g);
        write_rows<D>(adv, 1.f, dv + (tok0 + ki) * (size_t)lddkv + head * D, g);
    }
}

// ============================================================================================================
extern "C" int vlr_attn_fwd(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse,
                            const int* key_mask, int batch, int S, int heads, int head_dim, int causal, float scale,
                            hipStream_t st) {
    VLR_REQUIRE(batch > 0 && S > 0 && heads > 0, "vlr_attn_fwd: bad shape");
    VLR_REQUIRE(head_dim == 128 || head_dim == 64, "vlr_attn_fwd: head_dim must be 64 or 128, got %d", head_dim);
    VLR_REQUIRE(ld % 8 == 0 && ldo % 8 == 0, "vlr_attn_fwd: row strides must be multiples of 8 elements");
    const dim3 grid((S + 127) / 128, heads, batch);
    const float sl2 = scale * LOG2E;
    const int Sp = (S + 63) / 64 * 64;   // lse is [batch][heads][Sp]
    const int pi = vlr_prof_begin(VLR_K_ATTN_FWD, 4.0 * S * S * heads * head_dim * batch * (causal ? 0.5 : 1.0), st);
#define LAUNCH(D_, C_)                                                                                                  \
    hipLaunchKernelGGL((attn_fwd_kernel<D_, C_>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,           \
                       (const bf16_t*)v, ld, (bf16_t*)o, ldo, lse, key_mask, S, Sp, sl2)
    if (head_dim == 128) { if (causal) LAUNCH(128, true); else LAUNCH(128, false); }
    else { if (causal) LAUNCH(64, true); else LAUNCH(64, false); }
#undef LAUNCH
    vlr_prof_end(pi, st);
    return vlr_check_launch("vlr_attn_fwd");
}

extern "C" int vlr_attn_bwd(const void* q, const void* k, const void* v, int ld, const void* o, const void* dout,
                            int ldo, const float* lse, float* delta_ws, const int* key_mask, void* dq, void* dk,
                            void* dv, int ldd, int batch, int S, int heads, int head_dim, int causal, float scale,
                            hipStream_t st) {
    VLR_REQUIRE(batch > 0 && S > 0 && heads > 0, "vlr_attn_bwd: bad shape");
    VLR_REQUIRE(head_dim == 128, "vlr_attn_bwd: head_dim must be 128 (the ViT is frozen), got %d", head_dim);
    VLR_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && ldd % 8 == 0 && delta_ws && lse, "vlr_attn_bwd: strides / workspace");
    const int Sp = (S + 63) / 64 * 64;   // lse and delta_ws are [batch][heads][Sp] floats
    const int pi = vlr_prof_begin(VLR_K_ATTN_BWD, 10.0 * S * S * heads * head_dim * batch * (causal ? 0.5 : 1.0), st);
    hipLaunchKernelGGL(attn_delta_kernel, dim3(batch * S), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)o, ldo,
                       delta_ws, S, Sp, heads);
    const dim3 grid((S + 127) / 128, heads, batch);
    if (causal) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, ld, (const bf16_t*)dout, ldo, lse, delta_ws, key_mask, (bf16_t*)dq, ldd, S, Sp, scale);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<true>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, ld, (const bf16_t*)dout, ldo, lse, delta_ws, key_mask, (bf16_t*)dk, (bf16_t*)dv, ldd, S, Sp, scale);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<false>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, ld, (const bf16_t*)dout, ldo, lse, delta_ws, key_mask, (bf16_t*)dq, ldd, S, Sp, scale);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<false>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, ld, (const bf16_t*)dout, ldo, lse, delta_ws, key_mask, (bf16_t*)dk, (bf16_t*)dv, ldd, S, Sp, scale);
    }
    vlr_prof_end(pi, st);
    return vlr_check_launch("vlr_attn_bwd");
}
