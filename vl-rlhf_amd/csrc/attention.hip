// Fused attention for gfx950 (MI355X): forward (causal + key-padding mask for the LLaMA decoder, D=128; full
// attention for the CLIP ViT, D=64) and backward (D=128).  No S x S matrix is ever written to HBM.
//
// Tiling is built around v_mfma_f32_32x32x16_bf16 with the SWAPPED product S^T = K . Q^T so that a query's row of
// scores lives in ONE lane pair (l, l^32): softmax statistics are lane-local + one cross-lane exchange, and the
// bf16 P (or dS) fragment that feeds the second MFMA is packed from the accumulator registers in place.  Every LDS tile
// is a plain row-major [64 rows][D] image (16-byte chunks XOR-swizzled so that both access patterns below are
// bank-conflict free):
//   * "row" fragments  (K for S^T, V for dP^T, Q / dO in the dK,dV kernel): one ds_read_b128 per 32x16 operand;
//   * "transposed" fragments (V^T for P.V, K^T for dQ, Q^T / dO^T for dK / dV): two ds_read_b64_tr_b16 - the hardware
//     transposes 4x16 blocks on the way out of LDS, and the two reads are aimed at the row groups {4g..4g+3} and
//     {8+4g..8+4g+3} so the fragment's k-slot order equals the accumulator layout of the first MFMA.
// Staging is global -> registers -> LDS with the next tile's loads in flight under the current tile's MFMAs; rows past
// the end are clamped (their scores are masked), so the hot loop has no divergent load guards.  Masks (causal /
// key padding / tail) are only applied on tiles that need them (wave-uniform test).
//
// Layout: q, k, v are column blocks of one fused [tokens][3H] buffer (row stride ld); token row = b*S + s; head h
// owns columns [h*D, (h+1)*D).  lse is kept in the log2 domain: L2 = m + log2(l) with scores pre-multiplied by
// scale*log2(e), so P = exp2(s*scale*log2e - L2).  Fully masked query rows give O = 0, L2 = +inf (P == 0 in bwd).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#define KV_TILE 64
#define LOG2E 1.4426950408889634f

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// byte offset of 16-byte chunk `chunk` of row `row` in a [64][D] tile
template <int D>
__device__ __forceinline__ int tile_off(int row, int chunk) {
    if constexpr (D == 128) return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
    else return row * 128 + ((chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4);
}

template <int D>
struct TileRegs {
    static constexpr int CPR = D / 8;         // 16-byte chunks per row
    static constexpr int RPI = 256 / CPR;     // rows per pass of the 256 threads
    static constexpr int NI = 64 / RPI;
    u32x4 r[NI];
};
// global rows [row0, row0+64) x D -> registers (rows past nrows-1 are clamped: finite data, masked by the caller)
template <int D>
__device__ __forceinline__ void tile_load(TileRegs<D>& tr, const bf16_t* __restrict__ src, int ld, int row0, int nrows, int t) {
    const int c = t % TileRegs<D>::CPR;
#pragma unroll
    for (int i = 0; i < TileRegs<D>::NI; ++i) {
        int row = row0 + t / TileRegs<D>::CPR + TileRegs<D>::RPI * i;
        row = row < nrows ? row : nrows - 1;
        tr.r[i] = *reinterpret_cast<const u32x4*>(src + (size_t)row * ld + c * 8);
    }
}
template <int D>
__device__ __forceinline__ void tile_store(const TileRegs<D>& tr, char* lds, int t) {
    const int c = t % TileRegs<D>::CPR;
#pragma unroll
    for (int i = 0; i < TileRegs<D>::NI; ++i)
        *reinterpret_cast<u32x4*>(lds + tile_off<D>(t / TileRegs<D>::CPR + TileRegs<D>::RPI * i, c)) = tr.r[i];
}
// 32 x 16 operand whose 32-index is the tile ROW (rows rbase..rbase+31) and whose k-slots are 16 columns at chunk cchunk
template <int D>
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int rbase, int cchunk, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + tile_off<D>(rbase + (lane & 31), cchunk + (lane >> 5)));
}
// 32 x 16 operand whose 32-index is the tile COLUMN (cols cbase..cbase+31) and whose k-slots are the 16 rows of block
// `ks`, in accumulator order: lane group g holds rows {4g..4g+3, 8+4g..8+4g+3}.  ds_read_b64_tr_b16: within a 16-lane
// group lane p supplies the address of (row p/4, 4 columns at (p%4)*4) and receives column p of that 4 x 16 block.
template <int D>
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int cbase, int ks, int lane) {
    const int q4 = lane >> 4, pq = lane & 15;
    const int col = cbase + 16 * (q4 & 1) + (pq & 3) * 4;
    const int row = ks * 16 + 4 * (q4 >> 1) + (pq >> 2);
    const int sub = ((col >> 2) & 1) * 8;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + tile_off<D>(row, col >> 3) + sub));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(tile + tile_off<D>(row + 8, col >> 3) + sub));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x16& s, int h) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16(s[8 * h + 2 * i], s[8 * h + 2 * i + 1]);
    return __builtin_bit_cast(bf16x8, w);
}

// write a transposed accumulator tile set acc[db][r] = X^T[d = db*32 + crow(r,g)][row = lane&31] to X[row][d],
// crow(r,g) = (r&3) + 8*(r>>2) + 4*g
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

// Same result through LDS: the accumulator layout gives every lane ONE row, so write_rows issues 16 8-byte stores per lane that
// each touch 32 different rows (store-issue bound, guide T21).  Here the wave drops its 32 x D tile into a private LDS image
// (16-byte chunks XOR-swizzled by the row: the 8-byte writes are 2-way conflicted at worst, the reads conflict free) and
// stores WHOLE rows: one wave instruction = 64 / (D/8) rows x D bf16, contiguous.  Rows >= nvalid are not written.
template <int D>
__device__ __forceinline__ void write_rows_staged(const f32x16* acc, float mul, char* stage, bf16_t* __restrict__ dst0, size_t ld,
                                                  int nvalid) {
    constexpr int CPR = D / 8, RB = D * 2, RPI = 64 / CPR;
    // the lane id is recomputed here (v_mbcnt) instead of being kept alive across the tile loop: the kernels sit at the 256-register
    // edge and one more live VGPR spills inside the loop (a scratch reload there also drains the DMA queue)
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int row = lane & 31, g = lane >> 5;
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
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
        if (r < nvalid) *reinterpret_cast<u32x4*>(dst0 + (size_t)r * ld + c * 8) = v;
    }
}

// key-validity of a 64-key tile -> additive bias row in LDS; returns (block-uniform) whether any key is masked
__device__ __forceinline__ int stage_bias(float* bias_lds, const int* __restrict__ kmask, size_t tok0, int k0, int S, int t) {
    int bad = 0;
    if (t < KV_TILE) {
        const int key = k0 + t;
        const bool ok = key < S && (!kmask || kmask[tok0 + key] != 0);
        bias_lds[t] = ok ? 0.f : -INFINITY;
        bad = !ok;
    }
    return __syncthreads_or(bad);
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
    char* v_lds = smem + KV_TILE * D * 2;
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
    float m = -INFINITY, l = 0.f;      // m in the scaled log2 domain

    const int q_end = min(S, qblk * 128 + 128);
    const int nkv = CAUSAL ? (q_end + KV_TILE - 1) / KV_TILE : (S + KV_TILE - 1) / KV_TILE;
    TileRegs<D> kreg, vreg;
    tile_load<D>(kreg, kh, ld, 0, S, t);
    tile_load<D>(vreg, vh, ld, 0, S, t);
    for (int it = 0; it < nkv; ++it) {
        const int k0 = it * KV_TILE;
        __syncthreads();                                   // every wave is done reading the previous tile
        tile_store<D>(kreg, k_lds, t);
        tile_store<D>(vreg, v_lds, t);
        const int masked_keys = stage_bias(bias_lds, kmask, tok0, k0, S, t);   // barrier inside
        if (it + 1 < nkv) {                                // next tile's global loads fly under this tile's MFMAs
            tile_load<D>(kreg, kh, ld, k0 + KV_TILE, S, t);
            tile_load<D>(vreg, vh, ld, k0 + KV_TILE, S, t);
        }
        if (CAUSAL && k0 > qw0 + 31) continue;             // wave-uniform: whole tile is in this wave's future

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int st = 0; st < D / 16; ++st)
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(k_lds, kb * 32, 2 * st, lane), qf[st], s[kb], 0, 0, 0);
        }
        // masks only where a tile needs them: diagonal tiles (causal) and tiles holding padded / out-of-range keys
        if ((CAUSAL && k0 + KV_TILE - 1 > qw0) || masked_keys) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + kb * 32 + 8 * rq + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = k0 + kb * 32 + 8 * rq + 4 * g + e;
                        if (bb[e] != 0.f || (CAUSAL && key > qi)) s[kb][4 * rq + e] = -INFINITY;
                    }
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx * scale_log2);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m - m_use);
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], scale_log2, -m_use));
                s[kb][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {   // the running max moved for some query of this wave
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 pf = pack_frag(s[ks >> 1], ks & 1);
#pragma unroll
            for (int db = 0; db < D / 32; ++db)
                acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(v_lds, db * 32, ks, lane), pf, acc[db], 0, 0, 0);
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
// Forward, LDS-DMA version.  Same tiling and arithmetic as attn_fwd_kernel; what changes is how K / V tiles reach LDS and
// how many waves share a SIMD:
//   * K and V tiles are double buffered in LDS and fetched by LDS-DMA (global_load_lds_dwordx4, one 1-KiB piece = 1024/(2D)
//     rows per wave instruction, 16-byte chunks permuted on the SOURCE side so that the lane-linear LDS image equals
//     tile_off<D>); no staging registers, no ds_write, ONE barrier per tile: {vmcnt(0); barrier; issue tile t+1; compute t};
//   * key validity is turned into one 64-bit mask per KV tile in the prologue (no per-tile bias row / block-wide OR);
//   * <= 128 VGPR + 128 AGPR per wave and 64.5 KiB LDS per block -> TWO blocks per CU: a wave's softmax (VALU) runs under
//     the MFMAs of the wave it shares the SIMD with.
// The DMA is issued through inline asm so that hipcc does not put s_waitcnt vmcnt(0) before the transposing LDS reads.
// ------------------------------------------------------------------------------------------------------------
// Workgroup -> (batch, head, 128-row block) map of the LDS-DMA kernels.  Workgroup L runs on XCD L % 8 (round-robin dispatch)
// and every XCD has its own L2, so ALL blocks that stream the same K/V (the nblk row blocks of one head, and under
// grouped-query attention the `group` query heads that share a K/V head) are given to ONE XCD, consecutively: the K/V (or
// Q/dO) panels are then fetched into one L2 once instead of into eight (the (x = block, y = head, z = batch) grid spread the
// blocks of a head over all XCDs: 3.5x the algorithmic HBM traffic, measured, profiles/r01_pmc_hbm_traffic_*).
#include "attn_grid.h"
typedef __attribute__((address_space(3))) void attn_lvoid_t;
__device__ __forceinline__ void attn_dma16(const bf16_t* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// 4 bytes per lane -> LDS [m0 + lane*4]: one wave instruction moves 64 consecutive floats (per-tile lse / delta rows)
__device__ __forceinline__ void attn_dma4(const float* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// value of the partner lane l ^ 32 without the LDS crossbar: v_permlane32_swap exchanges lanes 32..63 of its first register with lanes
// 0..31 of its second; with x in both, one register then holds the lower lane's x everywhere and the other the upper lane's.  Inline asm:
// hipcc (ROCm 7.2) folds fmaxf(r[0], r[1]) of __builtin_amdgcn_permlane32_swap(u, u) to r[0] (tools/probe_permlane32_swap.hip).  s_nop 1 =
// the two wait states between a VALU write of an operand and the swap.  VLR_ATTN_SWAP=0 (compile time: ATTN_SWAP) keeps ds_bpermute.
#ifndef ATTN_SWAP
#define ATTN_SWAP 1
#endif
__device__ __forceinline__ float attn_pair_max(float x) {
#if ATTN_SWAP
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
#else
    return fmaxf(x, __shfl_xor(x, 32));
#endif
}
// dS = P (dP - delta) as two scalar VALU instructions: left to hipcc, the SLP vectoriser packs adjacent elements into v_pk_add_f32 /
// v_pk_mul_f32, which beside MFMAs cost +22..26 cycles per gap (MI355X_MICROARCH.md).  p usually comes straight out of v_exp_f32: the
// v_sub in front of the v_mul that reads it is the wait state gfx950 needs between a transcendental and a VALU reader (hipcc does not
// pad asm consumers).
__device__ __forceinline__ float attn_ds(float p, float dp, float delta) {
    float d;
    asm("v_sub_f32 %0, %1, %2\n\tv_mul_f32 %0, %0, %3" : "=&v"(d) : "v"(dp), "v"(delta), "v"(p));
    return d;
}
// key-validity masks of the first nkv KV tiles, one 64-bit word per tile (bit = key is masked or past S).  Wave w takes tiles
// w, w+4, ...; four tiles' loads are issued before the first ballot (one load latency per four tiles instead of one per tile:
// this runs before the first MFMA of every workgroup)
__device__ __forceinline__ void attn_tile_masks(unsigned long long* tilemask, const int* __restrict__ kmask, size_t tok0, int S,
                                                int nkv, int wave, int lane) {
    for (int base = wave; base < nkv; base += 16) {
        int ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = (base + 4 * u) * KV_TILE + lane;
            ok[u] = key < S ? (kmask ? kmask[tok0 + key] : 1) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long bad = __builtin_amdgcn_ballot_w64(ok[u] == 0);
            if (lane == 0 && base + 4 * u < nkv) tilemask[base + 4 * u] = bad;
        }
    }
}
// Retire the prologue's global loads of loop-invariant fragments HERE.  hipcc places the s_waitcnt vmcnt(N) for a loaded register at
// its first use - inside the tile loop - and cannot see the LDS-DMA instructions (inline asm) issued there: vmcnt retires in
// order, so that wait also drained the NEXT tile's DMA in the middle of the current tile's MFMAs, every iteration (found in
// the r01 ISA: vmcnt(7)..vmcnt(0) between the QK^T MFMAs; the double buffering never overlapped anything).
#define ATTN_RETIRE(frag) asm volatile("" : "+v"(frag))
// s_setprio(1) around the MFMA groups of the LDS-DMA kernels: two independent workgroups share a CU, so a wave in its matrix
// section should win the issue arbitration against its SIMD partner's softmax / address arithmetic (guide T5)
#ifndef ATTN_SETPRIO
#define ATTN_SETPRIO 1
#endif
#define ATTN_PRIO(n) do { if (ATTN_SETPRIO) __builtin_amdgcn_s_setprio(n); } while (0)
#ifndef ATTN_SGB
#define ATTN_SGB 4           // forward: K fragment reads in flight ahead of the K . Q^T MFMAs (sched_group_barrier pipeline); 0: hipcc's own order
#endif
#ifndef ATTN_SGB_PV
#define ATTN_SGB_PV 2        // forward: V fragments in flight ahead of the V^T . P^T MFMAs; 0: hipcc's own order
#endif
#ifndef ATTN_SGB_DQ
#define ATTN_SGB_DQ 4        // dQ kernel: fragment reads in flight ahead of the dP / S MFMAs; 0: hipcc's own order
#endif
#ifndef ATTN_SGB_DQ2
#define ATTN_SGB_DQ2 2       // dQ kernel: transposed K fragments in flight ahead of the dQ MFMAs; 0: hipcc's own order
#endif
#define ATTN_MAX_TILES 128   // S <= 8192
#ifdef ATTN_TRACE2
// timing probe of attn_fwd2_kernel (diagnostics build only: build_hip.py --define ATTN_TRACE2=1 --tag _tr2, tools/attn_fwd2_trace.py): wave 0 of
// the workgroup that takes work item 0 stamps s_memtime at fixed points of its tiles (a stamp costs ~200 cycles itself)
__device__ unsigned long long a2_trace_buf[4096];
#define A2_STAMP(slot) do { if (a2_on && a2_n < 4000) { a2_trace_buf[a2_n++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); } } while (0)
#else
#define A2_STAMP(slot) do {} while (0)
#endif
template <int D>
struct AttnFwd2 {
    static constexpr int TILE_BYTES = KV_TILE * D * 2;
    static constexpr int LDS_BYTES = 4 * TILE_BYTES + ATTN_MAX_TILES * 8;
    static constexpr int RPP = 1024 / (2 * D);          // rows per DMA piece
    static constexpr int NPIECE = KV_TILE / RPP / 4;    // pieces per wave and tile (4 waves)
};
template <int D, bool CAUSAL>
__device__ __forceinline__ unsigned attn_fwd2_block(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                    const bf16_t* __restrict__ v, int ld, bf16_t* __restrict__ o,
                                                    int ldo, float* __restrict__ lse, const int* __restrict__ kmask,
                                                    int S, int Sp, float scale_log2, const AttnGrid& ag, int L, char* smem,
                                                    unsigned* ctr) {
    using CF = AttnFwd2<D>;
    unsigned long long* tilemask = reinterpret_cast<unsigned long long*>(smem + 4 * CF::TILE_BYTES);
    const int t = threadIdx.x, lane = t & 63, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int head, kvhead, b, qslot;
    unsigned nxt = 0;          // persistent kernel: the workgroup's next ticket (thread 0), drawn under the LAST tile
    if (!ag.decode(L, head, kvhead, b, qslot)) return (ctr && t == 0) ? atomicAdd(ctr, 1u) : 0u;
    const int nh = ag.heads;
    const size_t tok0 = (size_t)b * S;
    const bf16_t* qh = q + tok0 * ld + head * D;
    const bf16_t* kh = k + tok0 * ld + kvhead * D;
    const bf16_t* vh = v + tok0 * ld + kvhead * D;
    const int qblk = CAUSAL ? ag.nblk - 1 - qslot : qslot;   // causal: heaviest blocks first
    const int qw0 = qblk * 128 + wave * 32;
    const int qi = qw0 + (lane & 31);
    const int qrow = qi < S ? qi : S - 1;
    const int q_end = min(S, qblk * 128 + 128);
    const int nkv = CAUSAL ? (q_end + KV_TILE - 1) / KV_TILE : (S + KV_TILE - 1) / KV_TILE;

    // DMA geometry: piece pc = wave + 4*i covers tile rows [pc*RPP, (pc+1)*RPP); lane -> (row, LDS chunk position)
    constexpr int CPR = D / 8;
    const int prow = lane / CPR, ppos = lane % CPR;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lvoid_t*)smem;
    auto issue_tile = [&](int it) {
        const int k0 = it * KV_TILE;
        const uint32_t dst = lds0 + (it & 1) * 2 * CF::TILE_BYTES;
#pragma unroll
        for (int i = 0; i < CF::NPIECE; ++i) {
            const int pc = wave + 4 * i;
            const int r = pc * CF::RPP + prow;                      // row inside the tile
            // LDS position ppos of row r holds chunk c with tile_off(r, c) == r*rowbytes + ppos*16
            const int swz = (tile_off<D>(r, 0) - r * (D * 2)) >> 4;
            const int c = ppos ^ swz;
            int row = k0 + r;
            row = row < S ? row : S - 1;
            const uint32_t off = (uint32_t)(((size_t)row * ld + c * 8) * 2);
            attn_dma16(kh, off, dst + pc * 1024);
            attn_dma16(vh, off, dst + CF::TILE_BYTES + pc * 1024);
        }
    };
    issue_tile(0);

    // key-validity masks, one 64-bit word per KV tile (bit = key is masked)
    attn_tile_masks(tilemask, kmask, tok0, S, nkv, wave, lane);

    bf16x8 qf[D / 16];
#pragma unroll
    for (int st = 0; st < D / 16; ++st)
        qf[st] = *reinterpret_cast<const bf16x8*>(qh + (size_t)qrow * ld + 16 * st + 8 * g);

    f32x16 acc[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;      // m in the scaled log2 domain
#pragma unroll
    for (int st = 0; st < D / 16; ++st) ATTN_RETIRE(qf[st]);

#ifdef ATTN_TRACE2
    const bool a2_on = (L == 0) && wave == 0;
    int a2_n = 0;
#endif
    for (int it = 0; it < nkv; ++it) {
        const int k0 = it * KV_TILE;
        A2_STAMP(1);
        // tile `it` has landed (own pieces: vmcnt, everybody's: barrier) and nobody still reads the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        A2_STAMP(2);
        if (it + 1 < nkv) issue_tile(it + 1);
        else if (ctr && t == 0) nxt = atomicAdd(ctr, 1u);  // late, so that the blocks stay dynamically balanced to the end
        A2_STAMP(3);
        if (CAUSAL && k0 > qw0 + 31) continue;             // wave-uniform: whole tile is in this wave's future
        const char* k_lds = smem + (it & 1) * 2 * CF::TILE_BYTES;
        const char* v_lds = k_lds + CF::TILE_BYTES;

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int st = 0; st < D / 16; ++st)
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(k_lds, kb * 32, 2 * st, lane), qf[st], s[kb], 0, 0, 0);
        }
        // masks only where a tile needs them: diagonal tiles (causal) and tiles holding padded / out-of-range keys
        const unsigned long long mk = tilemask[it];
#if ATTN_SGB
        // issue order of the K . Q^T product (round 6): ATTN_SGB fragment reads ahead of their MFMAs.  Left alone hipcc reads one
        // key block's fragment into ONE temporary right in front of its MFMA (ds_read; s_waitcnt lgkmcnt(0); v_mfma, eight times per
        // tile): an LDS round trip per two MFMAs that only the SIMD's other wave can cover
        __builtin_amdgcn_sched_group_barrier(0x100, ATTN_SGB, 0);
#pragma unroll
        for (int i = 0; i < 2 * (D / 16) - ATTN_SGB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, ATTN_SGB, 0);
#endif
        if (CAUSAL && mk == 0ull) {
            if (k0 + KV_TILE - 1 > qw0) {
                // diagonal tile, no padded key (one per wave and block): key kk = c(kb, r) + 4g is in the query's future iff
                // c > qi - k0 - 4g - a compare against an inline constant and a select per score
                const int thr = qi - k0 - 4 * g;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kb * 32 + (r & 3) + 8 * (r >> 2) > thr) s[kb][r] = -INFINITY;
            }
        } else if ((CAUSAL && k0 + KV_TILE - 1 > qw0) || mk != 0ull) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kk = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    if (((mk >> kk) & 1ull) || (CAUSAL && k0 + kk > qi)) s[kb][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = attn_pair_max(mx);
        A2_STAMP(4);
        const float m_new = fmaxf(m, mx * scale_log2);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m - m_use);
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pp = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], scale_log2, -m_use));
                s[kb][r] = pp;
                rs += pp;
            }
#if ATTN_SWAP
        l = l * alpha + rs;            // this lane's 32 of the tile's 64 keys: the lane pair shares m, its sums are added once, after the loop
#else
        rs += __shfl_xor(rs, 32);
        l = l * alpha + rs;
#endif
        m = m_new;
#ifdef ATTN_TRACE2
        asm volatile("" : "+v"(l));
        A2_STAMP(5);
#endif
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {   // the running max moved for some query of this wave
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 pf = pack_frag(s[ks >> 1], ks & 1);
#pragma unroll
            for (int db = 0; db < D / 32; ++db)
                acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(v_lds, db * 32, ks, lane), pf, acc[db], 0, 0, 0);
        }
#if ATTN_SGB_PV
        // ... and of the V^T . P^T product: ATTN_SGB_PV fragments (two transposing reads each) ahead
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * ATTN_SGB_PV, 1);
#pragma unroll
        for (int i = 0; i < 4 * (D / 32) - ATTN_SGB_PV; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, ATTN_SGB_PV, 1);
#endif
#ifdef ATTN_TRACE2
        asm volatile("" : "+v"(acc[0]));
        A2_STAMP(6);
#endif
    }
#if ATTN_SWAP
    l += __shfl_xor(l, 32);
#endif
    {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        if (ag.epi) {
            // the buffer tile nkv-1 did NOT use is free: every wave passed the last barrier, i.e. finished tile nkv-2
            char* stage = smem + (nkv & 1) * 2 * CF::TILE_BYTES + wave * (32 * D * 2);
            write_rows_staged<D>(acc, inv, stage, o + (tok0 + qw0) * (size_t)ldo + head * D, ldo, S - qw0);
        } else if (qi < S) {
            write_rows<D>(acc, inv, o + (tok0 + qi) * (size_t)ldo + head * D, g);
        }
    }
    if (lse && g == 0 && qi < Sp) lse[((size_t)b * nh + head) * Sp + qi] = (qi < S && l > 0.f) ? m + log2f(l) : INFINITY;
    return nxt;
}
// One workgroup per query block (ag.ctr == NULL), or PERSISTENT workgroups (two per CU) that draw query blocks from a per-XCD
// ticket counter: a workgroup lives ~40 us at S = 1599 and the fixed cost around it (dispatch, LDS / register allocation, the
// prologue's first memory round trip) was 20-30 % of the kernel (tools/attn_sweep.py: TF/s against S at constant work).  The
// next ticket is drawn under the LAST tile of the current block (latency hidden, and no block is promised before a
// workgroup is nearly free: drawing it at the start made the last round static and cost 2-20 %); the order of the blocks inside an XCD
// (heaviest first, all blocks of one K/V head consecutively) is the same as the one-workgroup-per-block grid.  The last
// workgroup of an XCD to leave resets its counters for the next launch on this stream.
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, int ld, bf16_t* __restrict__ o,
                                                           int ldo, float* __restrict__ lse, const int* __restrict__ kmask,
                                                           int S, int Sp, float scale_log2, AttnGrid ag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K tile | V tile] | tile masks
    if (!ag.ctr) {
        attn_fwd2_block<D, CAUSAL>(q, k, v, ld, o, ldo, lse, kmask, S, Sp, scale_log2, ag, blockIdx.x, smem, nullptr);
        return;
    }
    __shared__ int s_item;
    const int xcd = blockIdx.x & 7;
    unsigned* ctr = ag.ctr + xcd * 32;
    if (threadIdx.x == 0) s_item = (int)atomicAdd(ctr, 1u);
    __syncthreads();
    int item = __builtin_amdgcn_readfirstlane(s_item);
    while (item < ag.items) {
        const unsigned nxt = attn_fwd2_block<D, CAUSAL>(q, k, v, ld, o, ldo, lse, kmask, S, Sp, scale_log2, ag, item * 8 + xcd, smem, ctr);
        __syncthreads();                 // every wave is done with the tiles, the masks and s_item
        if (threadIdx.x == 0) s_item = (int)nxt;
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_item);
    }
    if (threadIdx.x == 0 && atomicAdd(ctr + 1, 1u) == (unsigned)(gridDim.x / 8 - 1)) {
        ctr[0] = 0;
        ctr[1] = 0;
    }
}

#include "attn_fwd3.h"

// ------------------------------------------------------------------------------------------------------------
// delta[b][h][s] = sum_d dO[s][h*128+d] * O[s][h*128+d]   (D = 128)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                         int ldo, float* __restrict__ delta, int S, int Sp, int nh) {
    // one workgroup per PADDED row: rows [S, Sp) are written as zeros - the dK,dV kernel multiplies P (exactly 0 there: lse is
    // +inf) by (dP - delta) without a guard, and 0 x (stale NaN) would poison dK
    const int b = (int)(blockIdx.x / Sp), s = (int)(blockIdx.x % Sp);
    if (s >= S) {
        for (int h = threadIdx.x; h < nh; h += 256) delta[((size_t)b * nh + h) * Sp + s] = 0.f;
        return;
    }
    const size_t row = (size_t)b * S + s;   // token row
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
    __shared__ __attribute__((aligned(16))) char smem[KV_TILE * D * 2 * 2 + KV_TILE * 4];
    char* k_lds = smem;
    char* v_lds = smem + KV_TILE * D * 2;
    float* bias_lds = reinterpret_cast<float*>(smem + KV_TILE * D * 4);
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
        tile_store<D>(kreg, k_lds, t);
        tile_store<D>(vreg, v_lds, t);
        const int masked_keys = stage_bias(bias_lds, kmask, tok0, k0, S, t);
        if (it + 1 < nkv) {
            tile_load<D>(kreg, kh, ld, k0 + KV_TILE, S, t);
            tile_load<D>(vreg, vh, ld, k0 + KV_TILE, S, t);
        }
        if (CAUSAL && k0 > qw0 + 31) continue;
        const bool need_mask = (CAUSAL && k0 + KV_TILE - 1 > qw0) || masked_keys;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(k_lds, kb * 32, 2 * st, lane), qf[st], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(v_lds, kb * 32, 2 * st, lane), dof[st], dp, 0, 0, 0);
            }
            if (need_mask) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + kb * 32 + 8 * rq + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = k0 + kb * 32 + 8 * rq + 4 * g + e;
                        if (bb[e] != 0.f || (CAUSAL && key > qi)) s[4 * rq + e] = -INFINITY;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2, -L2));   // masked / L2=+inf -> 0
                s[r] = p * (dp[r] - dl) * scale;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 dsf = pack_frag(s, h);
                const int ks = kb * 2 + h;
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(k_lds, db * 32, ks, lane), dsf, acc[db], 0, 0, 0);
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
    __shared__ __attribute__((aligned(16))) char smem[KV_TILE * D * 2 * 2];
    char* q_lds = smem;
    char* do_lds = smem + KV_TILE * D * 2;
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
    const bool any_bad_key = __builtin_amdgcn_ballot_w64(!key_ok) != 0;
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
        tile_store<D>(qreg, q_lds, t);
        tile_store<D>(doreg, do_lds, t);
        __syncthreads();
        if (q0 + KV_TILE < S) {
            tile_load<D>(qreg, qh, ld, q0 + KV_TILE, S, t);
            tile_load<D>(doreg, doh, ldo, q0 + KV_TILE, S, t);
        }
        if (CAUSAL && q0 + KV_TILE - 1 < kw0) continue;   // every query of the tile precedes this wave's keys
        const bool need_mask = (CAUSAL && q0 < kw0 + 31) || any_bad_key;   // wave-uniform
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(q_lds, qb * 32, 2 * st, lane), kf[st], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(do_lds, qb * 32, 2 * st, lane), vf[st], dp, 0, 0, 0);
            }
            f32x16 p;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 l2 = *reinterpret_cast<const f32x4*>(lse_h + q0 + qb * 32 + 8 * rq + 4 * g);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(dl_h + q0 + qb * 32 + 8 * rq + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e;
                    float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2, -l2[e]));   // l2 = +inf past S -> 0
                    if (need_mask) {
                        const int qq = q0 + qb * 32 + 8 * rq + 4 * g + e;
                        if (!key_ok || (CAUSAL && ki > qq)) pv = 0.f;
                    }
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
                    adv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(do_lds, db * 32, ks, lane), pf, adv[db], 0, 0, 0);
                    adk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(q_lds, db * 32, ks, lane), dsf, adk[db], 0, 0, 0);
                }
            }
        }
    }
    if (ki < S) {
        write_rows<D>(adk, 1.f, dk + (tok0 + ki) * (size_t)lddkv + head * D, g);
        write_rows<D>(adv, 1.f, dv + (tok0 + ki) * (size_t)lddkv + head * D, g);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Backward kernels, LDS-DMA versions (same arithmetic as attn_bwd_dq_kernel / attn_bwd_dkv_kernel; tiles double buffered in
// LDS and fetched by LDS-DMA, one barrier per tile, key masks as 64-bit words - see attn_fwd2_kernel).
// ------------------------------------------------------------------------------------------------------------
// LDS-DMA of one [64][128] tile whose rows are rows [row0, row0+64) (clamped to nrows-1) of src (row stride ld)
__device__ __forceinline__ void attn_issue_tile128(const bf16_t* src, int ld, int row0, int nrows, uint32_t dst, int wave, int lane) {
    const int prow = lane >> 4, ppos = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = wave + 4 * i;
        const int r = pc * 4 + prow;
        const int c = ppos ^ (((r & 3) << 2) | ((r >> 2) & 3));      // tile_off<128>
        int row = row0 + r;
        row = row < nrows ? row : nrows - 1;
        attn_dma16(src, (uint32_t)(((size_t)row * ld + c * 8) * 2), dst + pc * 1024);
    }
}
#define ATTN_TILE_BARRIER()                               \
    do {                                                  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                \
        __builtin_amdgcn_s_barrier();                     \
        asm volatile("" ::: "memory");                    \
        __builtin_amdgcn_sched_barrier(0);                \
    } while (0)

template <bool CAUSAL>
__global__ __launch_bounds__(256, CAUSAL ? 2 : 1) void attn_bwd_dq2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                              const bf16_t* __restrict__ v, int ld,
                                                              const bf16_t* __restrict__ dout, int ldo,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              const int* __restrict__ kmask, bf16_t* __restrict__ dq, int lddq,
                                                              int S, int Sp, float scale, AttnGrid ag) {
    constexpr int D = 128, TB = KV_TILE * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K | V] | tile masks
    unsigned long long* tilemask = reinterpret_cast<unsigned long long*>(smem + 4 * TB);
    const int t = threadIdx.x, lane = t & 63, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int head, kvhead, b, qslot;
    if (!ag.decode(blockIdx.x, head, kvhead, b, qslot)) return;
    const int nh = ag.heads;
    const size_t tok0 = (size_t)b * S;
    const bf16_t* qh = q + tok0 * ld + head * D;
    const bf16_t* kh = k + tok0 * ld + kvhead * D;
    const bf16_t* vh = v + tok0 * ld + kvhead * D;
    const bf16_t* doh = dout + tok0 * ldo + head * D;
    const int qblk = CAUSAL ? ag.nblk - 1 - qslot : qslot;
    const int qw0 = qblk * 128 + wave * 32;
    const int qi = qw0 + (lane & 31);
    const int qrow = qi < S ? qi : S - 1;
    const float scale_log2 = scale * LOG2E;
    const int q_end = min(S, qblk * 128 + 128);
    const int nkv = CAUSAL ? (q_end + KV_TILE - 1) / KV_TILE : (S + KV_TILE - 1) / KV_TILE;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lvoid_t*)smem;
    auto issue = [&](int it) {
        const uint32_t dst = lds0 + (it & 1) * 2 * TB;
        attn_issue_tile128(kh, ld, it * KV_TILE, S, dst, wave, lane);
        attn_issue_tile128(vh, ld, it * KV_TILE, S, dst + TB, wave, lane);
    };
    issue(0);
    attn_tile_masks(tilemask, kmask, tok0, S, nkv, wave, lane);

    bf16x8 qf[8], dof[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        qf[st] = *reinterpret_cast<const bf16x8*>(qh + (size_t)qrow * ld + 16 * st + 8 * g);
        dof[st] = *reinterpret_cast<const bf16x8*>(doh + (size_t)qrow * ldo + 16 * st + 8 * g);
    }
    float L2 = qi < S ? lse[((size_t)b * nh + head) * Sp + qrow] : INFINITY;
    float dl = delta[((size_t)b * nh + head) * Sp + qrow];

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) { ATTN_RETIRE(qf[st]); ATTN_RETIRE(dof[st]); }
    ATTN_RETIRE(L2);
    ATTN_RETIRE(dl);

    for (int it = 0; it < nkv; ++it) {
        const int k0 = it * KV_TILE;
        ATTN_TILE_BARRIER();
        if (it + 1 < nkv) issue(it + 1);
        if (CAUSAL && k0 > qw0 + 31) continue;
        const char* k_lds = smem + (it & 1) * 2 * TB;
        const char* v_lds = k_lds + TB;
        const unsigned long long mk = tilemask[it];
        const bool need_mask = (CAUSAL && k0 + KV_TILE - 1 > qw0) || mk != 0ull;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            ATTN_PRIO(1);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                // dP first, S last: the asm dS arithmetic below (attn_ds) reads dP without hipcc's MFMA -> VALU hazard padding (it pads
                // nothing for an asm reader), but it pads the exp's read of S - whose last MFMA now issues AFTER dP's last one
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(v_lds, kb * 32, 2 * st, lane), dof[st], dp, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(k_lds, kb * 32, 2 * st, lane), qf[st], s, 0, 0, 0);
            }
#if ATTN_SGB_DQ
            // ATTN_SGB_DQ fragment reads ahead of the sixteen dP / S MFMAs (see the forward kernel)
            __builtin_amdgcn_sched_group_barrier(0x100, ATTN_SGB_DQ, 0);
#pragma unroll
            for (int i = 0; i < 16 - ATTN_SGB_DQ; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, ATTN_SGB_DQ, 0);
#endif
            ATTN_PRIO(0);
            if (need_mask) {
                // a REAL branch: hipcc if-converted this wave-uniform block into 80 predicated instructions per key block that every
                // tile executed (v_and / v_cmp_ne_u64 / v_cmp_gt / s_or / v_cndmask per score: a third of the steady-state tile, found in
                // the round-4 ISA census); an asm statement cannot be speculated.  Only the diagonal tiles and tiles with padded keys come here.
                asm volatile("" ::: "memory");
                const uint32_t m32 = (uint32_t)(mk >> (32 * kb)) >> (4 * g);
                const int thr = qi - k0 - 32 * kb - 4 * g;          // key e (this lane's keys are e + 4 g) is in the query's future iff e > thr
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = (r & 3) + 8 * (r >> 2);
                    if (((m32 >> e) & 1u) || (CAUSAL && e > thr)) s[r] = -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pp = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2, -L2));   // masked / L2=+inf -> 0
                s[r] = attn_ds(pp, dp[r], dl);     // x scale: once, on dQ in the epilogue
            }
            ATTN_PRIO(1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 dsf = pack_frag(s, h);
                const int ks = kb * 2 + h;
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(k_lds, db * 32, ks, lane), dsf, acc[db], 0, 0, 0);
            }
#if ATTN_SGB_DQ2
            // ... and ATTN_SGB_DQ2 transposed K fragments (two reads each) ahead of the eight dQ MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * ATTN_SGB_DQ2, 1);
#pragma unroll
            for (int i = 0; i < 8 - ATTN_SGB_DQ2; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, ATTN_SGB_DQ2, 1);
#endif
            ATTN_PRIO(0);
        }
    }
    if (ag.epi) {
        char* stage = smem + (nkv & 1) * 2 * TB + wave * (32 * D * 2);   // the buffer the last tile did not use
        write_rows_staged<D>(acc, scale, stage, dq + (tok0 + qw0) * (size_t)lddq + head * D, lddq, S - qw0);
    } else if (qi < S) {
        write_rows<D>(acc, scale, dq + (tok0 + qi) * (size_t)lddq + head * D, g);
    }
}

// Round 4, dK,dV kernel (one wave per SIMD, 423 registers): two things the 64-query forward kernel (attn_fwd3.h) taught.
//  * With 512 registers per wave hipcc selects the AGPR form for EVERY v_mfma, so the scores S and dP landed in AGPRs and were read
//    back one register at a time for the exp / dS arithmetic: 64 v_accvgpr_read per tile.  The S / dP products are inline-asm MFMAs with
//    VGPR destinations now (dkv_mma_v); dK / dV stay builtin MFMAs (AGPR accumulators, which is where they belong).  hipcc pads no
//    hazard of an asm statement: the first reader of S / dP (pds) runs four MFMAs after the last one that wrote them (>= 128 cycles;
//    12 wait states are needed).
//  * The ~40 lane-derived LDS addresses were recomputed per tile from an opaque copy of the lane id (held across the loop they
//    spilled): ~110 address instructions per tile.  The swizzle does not depend on the query block / k step / tile, so 16 per-lane
//    base addresses (8 row-fragment chunks, 4 d blocks x {rows 0-7, rows 8-15} for the transposing reads) + 16 v_add of the stage
//    offset per tile + immediates cover every fragment (DkvAddr).
// VLR_ATTN_DKV_ASM=0 (compile time: ATTN_DKV_ASM) keeps the round-3 body for A/B.
#ifndef ATTN_DKV_ASM
#define ATTN_DKV_ASM 1
#endif
typedef __attribute__((address_space(3))) const bf16x8 attn_lds_bf16x8_t;
struct DkvAddr {
    uint32_t row[8];        // byte address of chunk 2 st + (lane >> 5) of row lane & 31 (tile at LDS offset 0)
    uint32_t tr[4][2];      // transposing reads: d block db, low / high row group, 16-key step 0
};
__device__ __forceinline__ void dkv_addr_init(DkvAddr& a, uint32_t lds0, int lane) {
    const int l31 = lane & 31, g = lane >> 5;
#pragma unroll
    for (int st = 0; st < 8; ++st) a.row[st] = lds0 + (uint32_t)tile_off<128>(l31, 2 * st + g);
    const int q4 = lane >> 4, pq = lane & 15;
    const int row = 4 * (q4 >> 1) + (pq >> 2);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const int col = db * 32 + 16 * (q4 & 1) + (pq & 3) * 4;
        const int sub = ((col >> 2) & 1) * 8;
        a.tr[db][0] = lds0 + (uint32_t)(tile_off<128>(row, col >> 3) + sub);
        a.tr[db][1] = lds0 + (uint32_t)(tile_off<128>(row + 8, col >> 3) + sub);
    }
}
// 32 x 16 row fragment: rows rb .. rb + 31 (rb = 0 | 32), chunk pair st, of the tile at byte offset `off` behind the bases
__device__ __forceinline__ bf16x8 dkv_frag_row(const uint32_t (&row)[8], int st, int rb, int off) {
    return *(attn_lds_bf16x8_t*)(uintptr_t)(row[st] + (uint32_t)(rb * 256 + off));
}
__device__ __forceinline__ bf16x8 dkv_frag_tr(const uint32_t (&tr)[4][2], int db, int ks, int off) {
    const uint32_t o = (uint32_t)(ks * 16 * 256 + off);
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(uintptr_t)(tr[db][0] + o));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(uintptr_t)(tr[db][1] + o));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <bool FIRST>
__device__ __forceinline__ void dkv_mma_v(f32x16& d, const bf16x8& a, const bf16x8& b) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// PIPE: the tile body as fenced half-units so that every LDS fragment is requested two half-units (8 MFMAs, 256 cycles) before
// the MFMA that consumes it.  Left to itself hipcc sinks each ds_read next to its MFMA (`ds_read; s_waitcnt lgkmcnt(0);
// v_mfma`, 64 times per tile): with ONE wave per SIMD (423 registers) nothing hides that latency and the kernel ran at 20 % MFMA
// utilisation - 4.5 us per tile against 0.9 us of MFMA work.  The unit table is at the tile body.
// Same arithmetic in the same order per accumulator as the unpipelined body (PIPE = 0, VLR_ATTN_PIPE=0): bit-identical.
template <bool CAUSAL, int PIPE>
__global__ __launch_bounds__(256) void attn_bwd_dkv2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, int ld,
                                                            const bf16_t* __restrict__ dout, int ldo,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            const int* __restrict__ kmask, bf16_t* __restrict__ dk,
                                                            bf16_t* __restrict__ dv, int lddkv, int S, int Sp, float scale,
                                                            AttnGrid ag) {
    constexpr int D = 128, TB = KV_TILE * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][Q | dO] | [2][lse 64 f32 | delta 64 f32]
    const int t = threadIdx.x, lane = t & 63, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int kvhead, b, kblk;
    if (!ag.decode_kv(blockIdx.x, kvhead, b, kblk)) return;
    const int nh = ag.heads, group = ag.group;
    const size_t tok0 = (size_t)b * S;
    const bf16_t* kh = k + tok0 * ld + kvhead * D;
    const bf16_t* vh = v + tok0 * ld + kvhead * D;
    const int kw0 = kblk * 128 + wave * 32;
    const int ki = kw0 + (lane & 31);
    const int krow = ki < S ? ki : S - 1;
    const bool key_ok = ki < S && (!kmask || kmask[tok0 + krow] != 0);
    const bool any_bad_key = __builtin_amdgcn_ballot_w64(!key_ok) != 0;
    const float scale_log2 = scale * LOG2E;
    const int q_start = CAUSAL ? (kblk * 128 / KV_TILE) * KV_TILE : 0;
    const int nq = (S - q_start + KV_TILE - 1) / KV_TILE;
    const int nit = nq * group;        // grouped-query attention: the `group` query heads of this K/V head, one after the other
    const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lvoid_t*)smem;
    auto issue = [&](int j) {
        const int head = kvhead * group + j / nq, it = j % nq;
        const uint32_t dst = lds0 + (j & 1) * 2 * TB;
        attn_issue_tile128(q + tok0 * ld + head * D, ld, q_start + it * KV_TILE, S, dst, wave, lane);
        attn_issue_tile128(dout + tok0 * ldo + head * D, ldo, q_start + it * KV_TILE, S, dst + TB, wave, lane);
        // the tile's 64 lse / delta values ride the same DMA queue (rows are padded to Sp: always in range) - a global load in
        // the loop would make the compiler wait vmcnt(0), i.e. for the next tile's DMA as well
        if (wave < 2) {
            const float* src = (wave == 0 ? lse : delta) + ((size_t)b * nh + head) * Sp + q_start + it * KV_TILE;
            attn_dma4(src, (uint32_t)lane * 4u, lds0 + 4 * TB + (j & 1) * 512 + wave * 256);
        }
    };
    issue(0);

    bf16x8 kf[8], vf[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        kf[st] = *reinterpret_cast<const bf16x8*>(kh + (size_t)krow * ld + 16 * st + 8 * g);
        vf[st] = *reinterpret_cast<const bf16x8*>(vh + (size_t)krow * ld + 16 * st + 8 * g);
    }
    DkvAddr fa0;
    dkv_addr_init(fa0, lds0, lane);
    f32x16 adk[4], adv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { adk[i][r] = 0.f; adv[i][r] = 0.f; }

#pragma unroll
    for (int st = 0; st < 8; ++st) { ATTN_RETIRE(kf[st]); ATTN_RETIRE(vf[st]); }

    if constexpr (PIPE > 0) {
        // one tile: barrier, next tile's DMA, the fenced half-units.  Two loops per query head - the leading tiles that need masks
        // (the diagonal of a causal block; every tile when the key block holds padded keys), then the rest - so that each loop
        // holds ONE body: with both bodies in one loop hipcc moved all 128 accumulator registers between AGPRs and VGPRs at
        // every iteration.  Every wave runs nq iterations per head whatever its split, so the barriers match.
        auto tile = [&](int j, int it, auto mask_c) {
            const int q0 = q_start + it * KV_TILE;
            const float* lse_t = reinterpret_cast<const float*>(smem + 4 * TB + (j & 1) * 512);   // this tile's 64 lse | 64 delta (tail = +inf)
            const float* dl_t = lse_t + 64;
            ATTN_TILE_BARRIER();
            if (j + 1 < nit) issue(j + 1);
            if (CAUSAL && q0 + KV_TILE - 1 < kw0) return;     // every query of the tile precedes this wave's keys
            const char* q_lds = smem + (j & 1) * 2 * TB;
            const char* do_lds = q_lds + TB;
            auto tile_body = [&](auto mask_c) {
                constexpr bool MASK = decltype(mask_c)::value;
                // per-tile opaque copy of the lane id: the ~40 lane-derived LDS addresses are recomputed per tile instead of being held
                // (and spilled) across the loop
#if ATTN_DKV_ASM
                const int gl = g;
                uint32_t arow[8], atr[4][2];         // this tile's fragment bases: the per-lane bases + the stage's offset
                {
                    const uint32_t so = (uint32_t)((j & 1) * 2 * TB);
#pragma unroll
                    for (int i = 0; i < 8; ++i) arow[i] = fa0.row[i] + so;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { atr[i][0] = fa0.tr[i][0] + so; atr[i][1] = fa0.tr[i][1] + so; }
                }
#else
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int gl = ln >> 5;
#endif
                // sixteen half-units of 4 MFMAs, each fed by 4 LDS fragments (16 registers) requested AHEAD half-units earlier:
                //   x = 0..7   S / dP:  qb = x >> 2, k-slots {2 (x & 3), 2 (x & 3) + 1}
                //   x = 8..15  dV / dK: qb = (x - 8) >> 2, h = ((x - 8) >> 1) & 1, d blocks {2 (x & 1), 2 (x & 1) + 1}
                // P / dS of accumulator rows 4 rq .. 4 rq + 3 of qb are VALU work beside half-unit 4 + 4 qb + rq (S, dP of qb are complete
                // after x = 4 qb + 3; dV / dK of (qb, h) start at x = 8 + 4 qb + 2 h); they are packed to bf16 as soon as a half (h) is done
                constexpr int AHEAD = PIPE;
                bf16x8 u[16][4];
                f32x4 l2v[8], dlv[8];
                f32x16 sc[2], dpc[2];
                f32x4 pmq[8];
                bf16x8 pf[2][2], dsf[2][2];
                auto ldh = [&](auto xc) {
                    constexpr int X = decltype(xc)::value;
                    if constexpr (X < 8) {
                        constexpr int qb = X >> 2;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
#if ATTN_DKV_ASM
                            u[X][2 * i] = dkv_frag_row(arow, 2 * (X & 3) + i, qb * 32, 0);
                            u[X][2 * i + 1] = dkv_frag_row(arow, 2 * (X & 3) + i, qb * 32, TB);
#else
                            u[X][2 * i] = frag_row<D>(q_lds, qb * 32, 2 * (2 * (X & 3) + i), ln);
                            u[X][2 * i + 1] = frag_row<D>(do_lds, qb * 32, 2 * (2 * (X & 3) + i), ln);
#endif
                        }
                    } else {
                        constexpr int qb = (X - 8) >> 2, h = ((X - 8) >> 1) & 1;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
#if ATTN_DKV_ASM
                            u[X][2 * i] = dkv_frag_tr(atr, 2 * (X & 1) + i, qb * 2 + h, TB);
                            u[X][2 * i + 1] = dkv_frag_tr(atr, 2 * (X & 1) + i, qb * 2 + h, 0);
#else
                            u[X][2 * i] = frag_tr<D>(do_lds, (2 * (X & 1) + i) * 32, qb * 2 + h, ln);
                            u[X][2 * i + 1] = frag_tr<D>(q_lds, (2 * (X & 1) + i) * 32, qb * 2 + h, ln);
#endif
                        }
                    }
                };
                auto ldl = [&](auto pc) {                    // lse / delta of the queries of P / dS slice pc = 4 qb + rq
                    constexpr int P = decltype(pc)::value, qb = P >> 2, rq = P & 3;
                    l2v[P] = *reinterpret_cast<const f32x4*>(lse_t + qb * 32 + 8 * rq + 4 * gl);
                    dlv[P] = *reinterpret_cast<const f32x4*>(dl_t + qb * 32 + 8 * rq + 4 * gl);
                };
                auto mmh = [&](auto xc) {
                    constexpr int X = decltype(xc)::value;
                    if constexpr (X < 8) {
                        constexpr int qb = X >> 2;
#if ATTN_DKV_ASM
                        dkv_mma_v<(X & 3) == 0>(sc[qb], u[X][0], kf[2 * (X & 3)]);
                        dkv_mma_v<(X & 3) == 0>(dpc[qb], u[X][1], vf[2 * (X & 3)]);
                        dkv_mma_v<false>(sc[qb], u[X][2], kf[2 * (X & 3) + 1]);
                        dkv_mma_v<false>(dpc[qb], u[X][3], vf[2 * (X & 3) + 1]);
#else
                        if constexpr ((X & 3) == 0) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) { sc[qb][r] = 0.f; dpc[qb][r] = 0.f; }
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            sc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[X][2 * i], kf[2 * (X & 3) + i], sc[qb], 0, 0, 0);
                            dpc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[X][2 * i + 1], vf[2 * (X & 3) + i], dpc[qb], 0, 0, 0);
                        }
#endif
                    } else {
                        constexpr int qb = (X - 8) >> 2, h = ((X - 8) >> 1) & 1;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            adv[2 * (X & 1) + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[X][2 * i], pf[qb][h], adv[2 * (X & 1) + i], 0, 0, 0);
                            adk[2 * (X & 1) + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u[X][2 * i + 1], dsf[qb][h], adk[2 * (X & 1) + i], 0, 0, 0);
                        }
                    }
                };
                auto pds = [&](auto pc) {
                    constexpr int P = decltype(pc)::value, qb = P >> 2, rq = P & 3;
                    // MASK: one word of "masked" bits per lane (= key): bit c <-> query q0 + 32 qb + c + 4g (the lane's scores sit at
                    // c = 8 rq + e); a padded key masks all of them, causality the queries before the key: c < ki - query0.  A select on
                    // a bit test - the `!key_ok || ki > qq` form compiled to a divergent branch around every exp
                    uint32_t bad = 0;
                    if constexpr (MASK) {
                        const int thr = ki - q0 - 32 * qb - 4 * gl;
                        bad = !key_ok ? 0xffffffffu : (!CAUSAL || thr <= 0 ? 0u : (thr >= 32 ? 0xffffffffu : ((1u << thr) - 1u)));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * rq + e;
                        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[qb][r], scale_log2, -l2v[P][e]));   // l2 = +inf past S -> 0
                        if constexpr (MASK) pv = (bad & (1u << (8 * rq + e))) ? 0.f : pv;
                        pmq[P][e] = pv;
                        sc[qb][r] = attn_ds(pv, dpc[qb][r], dlv[P][e]);   // x scale: once, on dK in the epilogue; delta is finite on padded rows
                    }
                    if constexpr (rq & 1) {                  // rows 8h .. 8h+7 done: the bf16 operands of dV / dK (pack_frag order)
                        constexpr int h = rq >> 1;
                        u32x4 wp, wd;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4& pp = pmq[P - 1 + (i >> 1)];
                            wp[i] = pack_bf16(pp[2 * (i & 1)], pp[2 * (i & 1) + 1]);
                            wd[i] = pack_bf16(sc[qb][8 * h + 2 * i], sc[qb][8 * h + 2 * i + 1]);
                        }
                        pf[qb][h] = __builtin_bit_cast(bf16x8, wp);
                        dsf[qb][h] = __builtin_bit_cast(bf16x8, wd);
                    }
                };
#define DKV_FENCE() __builtin_amdgcn_sched_barrier(0)
                auto step = [&](auto xc) {
                    constexpr int X = decltype(xc)::value;
                    if constexpr (X + AHEAD < 16) ldh(std::integral_constant<int, X + AHEAD>{});
#if ATTN_DKV_ASM
                    // P / dS slice p rides half-unit p + 5 (one later than with builtin MFMAs): hipcc pads no hazard of the asm S / dP
                    // products, and inside a half-unit it may order the slice's VALU reads in front of the unit's MFMAs - a whole
                    // half-unit (4 MFMAs, >= 128 cycles) now lies between the last write of S / dP of a query block and their first
                    // read, whatever that order.  Deadlines hold: P / dS of (qb, h) are packed in unit 6 + 4 qb + 2 h, read in 8 + 4 qb + 2 h.
                    if constexpr (X >= 4 && X < 12) ldl(std::integral_constant<int, X - 4>{});
                    DKV_FENCE();
                    mmh(xc);
                    if constexpr (X >= 5 && X < 13) pds(std::integral_constant<int, X - 5>{});
                    DKV_FENCE();
#else
                    if constexpr (X + 1 >= 4 && X + 1 < 12) ldl(std::integral_constant<int, X + 1 - 4>{});
                    DKV_FENCE();
                    mmh(xc);
                    if constexpr (X >= 4 && X < 12) pds(std::integral_constant<int, X - 4>{});
                    DKV_FENCE();
#endif
                };
                ldh(std::integral_constant<int, 0>{});
                if constexpr (AHEAD > 1) ldh(std::integral_constant<int, 1>{});
                DKV_FENCE();
                step(std::integral_constant<int, 0>{});  step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});  step(std::integral_constant<int, 3>{});
                step(std::integral_constant<int, 4>{});  step(std::integral_constant<int, 5>{});
                step(std::integral_constant<int, 6>{});  step(std::integral_constant<int, 7>{});
                step(std::integral_constant<int, 8>{});  step(std::integral_constant<int, 9>{});
                step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
                step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
                step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
#undef DKV_FENCE
            };
            tile_body(mask_c);
        };
        // tiles with q0 < kw0 + 31 hold queries that precede some of this wave's keys
        const int n_mask = any_bad_key ? nq : (CAUSAL ? min(nq, (kw0 + 31 - q_start + KV_TILE - 1) / KV_TILE) : 0);
        int j = 0;
        for (int m = 0; m < group; ++m) {
            int it = 0;
            for (; it < n_mask; ++it, ++j) tile(j, it, std::true_type{});
            for (; it < nq; ++it, ++j) tile(j, it, std::false_type{});
        }
    } else
    for (int j = 0; j < nit; ++j) {
        const int q0 = q_start + (j % nq) * KV_TILE;
        const float* lse_t = reinterpret_cast<const float*>(smem + 4 * TB + (j & 1) * 512);   // this tile's 64 lse | 64 delta (tail = +inf)
        const float* dl_t = lse_t + 64;
        ATTN_TILE_BARRIER();
        if (j + 1 < nit) issue(j + 1);
        if (CAUSAL && q0 + KV_TILE - 1 < kw0) continue;   // every query of the tile precedes this wave's keys
        const char* q_lds = smem + (j & 1) * 2 * TB;
        const char* do_lds = q_lds + TB;
        const bool need_mask = (CAUSAL && q0 < kw0 + 31) || any_bad_key;   // wave-uniform
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s, dp;

#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            ATTN_PRIO(1);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(q_lds, qb * 32, 2 * st, lane), kf[st], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(do_lds, qb * 32, 2 * st, lane), vf[st], dp, 0, 0, 0);
            }
            ATTN_PRIO(0);
            f32x16 pm;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 l2 = *reinterpret_cast<const f32x4*>(lse_t + qb * 32 + 8 * rq + 4 * g);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(dl_t + qb * 32 + 8 * rq + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e;
                    float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2, -l2[e]));   // l2 = +inf past S -> 0
                    if (need_mask) {
                        const int qq = q0 + qb * 32 + 8 * rq + 4 * g + e;
                        if (!key_ok || (CAUSAL && ki > qq)) pv = 0.f;
                    }
                    pm[r] = pv;
                    s[r] = pv * (dp[r] - dl[e]);
                }
            }
            ATTN_PRIO(1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x8 pf = pack_frag(pm, h);
                const bf16x8 dsf = pack_frag(s, h);
                const int ks = qb * 2 + h;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    adv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(do_lds, db * 32, ks, lane), pf, adv[db], 0, 0, 0);
                    adk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(q_lds, db * 32, ks, lane), dsf, adk[db], 0, 0, 0);
                }
            }
            ATTN_PRIO(0);
        }
    }
    if (ag.epi) {
        char* stage = smem + (nit & 1) * 2 * TB + wave * (32 * D * 2);   // the buffer the last tile did not use
        write_rows_staged<D>(adk, scale, stage, dk + (tok0 + kw0) * (size_t)lddkv + kvhead * D, lddkv, S - kw0);
        write_rows_staged<D>(adv, 1.f, stage, dv + (tok0 + kw0) * (size_t)lddkv + kvhead * D, lddkv, S - kw0);
    } else if (ki < S) {
        write_rows<D>(adk, scale, dk + (tok0 + ki) * (size_t)lddkv + kvhead * D, g);
        write_rows<D>(adv, 1.f, dv + (tok0 + ki) * (size_t)lddkv + kvhead * D, g);
    }
}

// ============================================================================================================
static int attn_lpt_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_ATTN_LPT");      // K/V heads per bundle: 1 = head-major, large = slot-major over all heads
        on = e ? atoi(e) : 8;
        if (on < 1) on = 1;
    }
    return on;
}
static int attn_pipe_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_ATTN_PIPE");      // dK,dV kernel: LDS fragments requested this many half-units ahead (0 = unpipelined body)
        on = e ? atoi(e) : 2;
    }
    return on;
}
static int attn_epi_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_ATTN_EPI");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on;
}
// ticket counters of the persistent kernels: one set per stream (the reference forward runs on a side stream beside the policy
// forward), zero at rest - the kernels reset them on the way out
#define ATTN_CTR_SLOTS 8
static unsigned* attn_counters(hipStream_t st) {
    static int on = -1;
    static unsigned* buf = nullptr;
    static int buf_dev = -1;
    static hipStream_t streams[ATTN_CTR_SLOTS];
    static int nstreams = 0;
    if (on < 0) {
        const char* e = getenv("VLR_ATTN_PERSIST");
        on = (e && e[0] == '0') ? 0 : 1;
        if (on && (hipGetDevice(&buf_dev) != hipSuccess ||
                   hipMalloc((void**)&buf, ATTN_CTR_SLOTS * 8 * 32 * sizeof(unsigned)) != hipSuccess ||
                   hipMemset(buf, 0, ATTN_CTR_SLOTS * 8 * 32 * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) on = 0;
    }
    if (!on) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != buf_dev) return nullptr;   // one process per GPU: the counters live on the first device used
    for (int i = 0; i < nstreams; ++i)
        if (streams[i] == st) return buf + (size_t)i * 8 * 32;
    if (nstreams == ATTN_CTR_SLOTS) return nullptr;
    streams[nstreams] = st;
    return buf + (size_t)(nstreams++) * 8 * 32;
}
static int attn_fwd3_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_ATTN_FWD3");      // 64 queries per wave, explicit issue order (attn_fwd3.h); 0 = the 32-query fwd2 kernel
        on = e ? atoi(e) : 0;
    }
    return on;
}
static int attn_dma_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_ATTN_DMA");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on;
}

#ifdef ATTN_TRACE2
extern "C" int vlr_attn_fwd2_trace(unsigned long long* host, int n) {      // diagnostics build: the s_memtime stamps of attn_fwd2_block
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(a2_trace_buf), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#endif
#ifdef F3_TRACE
extern "C" int vlr_attn_fwd3_trace(unsigned long long* host, int n) {      // diagnostics build: the s_memtime stamps of attn_fwd3.h
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(f3_trace_buf), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#endif
extern "C" int vlr_attn_fwd_gqa(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse,
                                const int* key_mask, int batch, int S, int heads, int kv_heads, int head_dim, int causal,
                                float scale, hipStream_t st) {
    VLR_REQUIRE(batch > 0 && S > 0 && heads > 0, "vlr_attn_fwd: bad shape");
    VLR_REQUIRE(kv_heads > 0 && heads % kv_heads == 0, "vlr_attn_fwd: heads %d is not a multiple of kv_heads %d", heads, kv_heads);
    VLR_REQUIRE(head_dim == 128 || head_dim == 64, "vlr_attn_fwd: head_dim must be 64 or 128, got %d", head_dim);
    VLR_REQUIRE(ld % 8 == 0 && ldo % 8 == 0, "vlr_attn_fwd: row strides must be multiples of 8 elements");
    const bool dma = attn_dma_on() && S <= ATTN_MAX_TILES * KV_TILE;
    VLR_REQUIRE(dma || heads == kv_heads, "vlr_attn_fwd: grouped-query attention needs the LDS-DMA kernels (S <= %d, VLR_ATTN_DMA != 0)", ATTN_MAX_TILES * KV_TILE);
    const dim3 grid((S + 127) / 128, heads, batch);
    const float sl2 = scale * LOG2E;
    const int Sp = (S + 63) / 64 * 64;   // lse is [batch][heads][Sp]
    const int pi = vlr_prof_begin(VLR_K_ATTN_FWD, 4.0 * S * S * heads * head_dim * batch * (causal ? 0.5 : 1.0), st);
#define LAUNCH(D_, C_)                                                                                                  \
    hipLaunchKernelGGL((attn_fwd_kernel<D_, C_>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k,           \
                       (const bf16_t*)v, ld, (bf16_t*)o, ldo, lse, key_mask, S, Sp, sl2)
    static bool attr = false;
    if (!attr) {
        attr = true;
        hipFuncSetAttribute((const void*)attn_fwd2_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, AttnFwd2<128>::LDS_BYTES);
        hipFuncSetAttribute((const void*)attn_fwd2_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, AttnFwd2<128>::LDS_BYTES);
        hipFuncSetAttribute((const void*)attn_fwd2_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, AttnFwd2<64>::LDS_BYTES);
        hipFuncSetAttribute((const void*)attn_fwd2_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, AttnFwd2<64>::LDS_BYTES);
    }
    AttnGrid ag;
    ag.heads = heads; ag.kv_heads = kv_heads; ag.group = heads / kv_heads; ag.nblk = (S + 127) / 128; ag.n_kvp = batch * kv_heads;
    ag.epi = attn_epi_on();
    ag.lpt = attn_lpt_on();
    int fgrid = ag.grid(false);
    const int resident = 2 * vlr_compute_cus();       // workgroups the chip holds at once: two per CU, whole XCD octets (minus the CUs left to RCCL)
    ag.ctr = fgrid > resident ? attn_counters(st) : nullptr;   // more query blocks than that: persistent workgroups
    ag.items = fgrid / 8;
    if (ag.ctr) fgrid = resident;
#define LAUNCH2(D_, C_)                                                                                                 \
    hipLaunchKernelGGL((attn_fwd2_kernel<D_, C_>), dim3(fgrid), dim3(256), AttnFwd2<D_>::LDS_BYTES, st, (const bf16_t*)q, \
                       (const bf16_t*)k, (const bf16_t*)v, ld, (bf16_t*)o, ldo, lse, key_mask, S, Sp, sl2, ag)
    if (dma && head_dim == 128 && attn_fwd3_on()) {
        static bool attr3 = false;
        if (!attr3) {
            attr3 = true;
            hipFuncSetAttribute((const void*)attn_fwd3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, AttnFwd3::LDS_BYTES);
            hipFuncSetAttribute((const void*)attn_fwd3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, AttnFwd3::LDS_BYTES);
        }
        // the block map with 64 F3_NW queries per workgroup (one 256-thread workgroup per CU, or two of 128), persistent beyond that
        AttnGrid a3 = ag;
        a3.nblk = (S + 64 * F3_NW - 1) / (64 * F3_NW);
        int g3 = a3.grid(false);
        const int res3 = (F3_NW == 4 ? 1 : 2) * vlr_compute_cus();
        a3.ctr = g3 > res3 ? attn_counters(st) : nullptr;
        a3.items = g3 / 8;
        if (a3.ctr) g3 = res3;
        if (causal) hipLaunchKernelGGL((attn_fwd3_kernel<true>), dim3(g3), dim3(64 * F3_NW), AttnFwd3::LDS_BYTES, st, (const bf16_t*)q, (const bf16_t*)k,
                                       (const bf16_t*)v, ld, (bf16_t*)o, ldo, lse, key_mask, S, Sp, sl2, a3);
        else hipLaunchKernelGGL((attn_fwd3_kernel<false>), dim3(g3), dim3(64 * F3_NW), AttnFwd3::LDS_BYTES, st, (const bf16_t*)q, (const bf16_t*)k,
                                (const bf16_t*)v, ld, (bf16_t*)o, ldo, lse, key_mask, S, Sp, sl2, a3);
    } else if (dma) {
        if (head_dim == 128) { if (causal) LAUNCH2(128, true); else LAUNCH2(128, false); }
        else { if (causal) LAUNCH2(64, true); else LAUNCH2(64, false); }
    } else {
        if (head_dim == 128) { if (causal) LAUNCH(128, true); else LAUNCH(128, false); }
        else { if (causal) LAUNCH(64, true); else LAUNCH(64, false); }
    }
#undef LAUNCH2
#undef LAUNCH
    vlr_prof_end(pi, st);
    return vlr_check_launch("vlr_attn_fwd");
}
extern "C" int vlr_attn_fwd(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse,
                            const int* key_mask, int batch, int S, int heads, int head_dim, int causal, float scale,
                            hipStream_t st) {
    return vlr_attn_fwd_gqa(q, k, v, ld, o, ldo, lse, key_mask, batch, S, heads, heads, head_dim, causal, scale, st);
}

extern "C" int vlr_attn_bwd_gqa(const void* q, const void* k, const void* v, int ld, const void* o, const void* dout,
                                int ldo, const float* lse, float* delta_ws, const int* key_mask, void* dq, void* dk,
                                void* dv, int ldd, int batch, int S, int heads, int kv_heads, int head_dim, int causal,
                                float scale, hipStream_t st) {
    VLR_REQUIRE(batch > 0 && S > 0 && heads > 0, "vlr_attn_bwd: bad shape");
    VLR_REQUIRE(kv_heads > 0 && heads % kv_heads == 0, "vlr_attn_bwd: heads %d is not a multiple of kv_heads %d", heads, kv_heads);
    VLR_REQUIRE(head_dim == 128, "vlr_attn_bwd: head_dim must be 128 (the ViT is frozen), got %d", head_dim);
    VLR_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && ldd % 8 == 0 && delta_ws && lse, "vlr_attn_bwd: strides / workspace");
    const bool dma = attn_dma_on() && S <= ATTN_MAX_TILES * KV_TILE;
    VLR_REQUIRE(dma || heads == kv_heads, "vlr_attn_bwd: grouped-query attention needs the LDS-DMA kernels (S <= %d, VLR_ATTN_DMA != 0)", ATTN_MAX_TILES * KV_TILE);
    const int Sp = (S + 63) / 64 * 64;   // lse and delta_ws are [batch][heads][Sp] floats
    const int pi = vlr_prof_begin(VLR_K_ATTN_BWD, 10.0 * S * S * heads * head_dim * batch * (causal ? 0.5 : 1.0), st);
    hipLaunchKernelGGL(attn_delta_kernel, dim3(batch * Sp), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)o, ldo,
                       delta_ws, S, Sp, heads);
    const dim3 grid((S + 127) / 128, heads, batch);
    constexpr int LDS_DQ = 4 * KV_TILE * 128 * 2 + ATTN_MAX_TILES * 8, LDS_DKV = 4 * KV_TILE * 128 * 2 + 1024;
    static bool attr = false;
    if (!attr) {
        attr = true;
        hipFuncSetAttribute((const void*)attn_bwd_dq2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        hipFuncSetAttribute((const void*)attn_bwd_dq2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
        hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
        hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
        hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
        hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
        hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
    }
    AttnGrid ag;
    ag.heads = heads; ag.kv_heads = kv_heads; ag.group = heads / kv_heads; ag.nblk = (S + 127) / 128; ag.n_kvp = batch * kv_heads;
    ag.epi = attn_epi_on();
    ag.lpt = attn_lpt_on();
    ag.ctr = nullptr; ag.items = 0;
    if (dma) {
        if (causal) {
            hipLaunchKernelGGL((attn_bwd_dq2_kernel<true>), dim3(ag.grid(false)), dim3(256), LDS_DQ, st, (const bf16_t*)q, (const bf16_t*)k,
                               (const bf16_t*)v, ld, (const bf16_t*)dout, ldo, lse, delta_ws, key_mask, (bf16_t*)dq, ldd, S, Sp, scale, ag);
#define DKV_LAUNCH(P_)                                                                                                    \
    hipLaunchKernelGGL((attn_bwd_dkv2_kernel<true, P_>), dim3(ag.grid(true)), dim3(256), LDS_DKV, st, (const bf16_t*)q, (const bf16_t*)k, \
                       (const bf16_t*)v, ld, (const bf16_t*)dout, ldo, lse, delta_ws, key_mask, (bf16_t*)dk, (bf16_t*)dv, ldd, S, Sp, scale, ag)
            if (attn_pipe_on() >= 2) DKV_LAUNCH(2);
            else if (attn_pipe_on() == 1) DKV_LAUNCH(1);
            else DKV_LAUNCH(0);
#undef DKV_LAUNCH
        } else {
            hipLaunchKernelGGL((attn_bwd_dq2_kernel<false>), dim3(ag.grid(false)), dim3(256), LDS_DQ, st, (const bf16_t*)q, (const bf16_t*)k,
                               (const bf16_t*)v, ld, (const bf16_t*)dout, ldo, lse, delta_ws, key_mask, (bf16_t*)dq, ldd, S, Sp, scale, ag);
#define DKV_LAUNCH(P_)                                                                                                    \
    hipLaunchKernelGGL((attn_bwd_dkv2_kernel<false, P_>), dim3(ag.grid(true)), dim3(256), LDS_DKV, st, (const bf16_t*)q, (const bf16_t*)k, \
                       (const bf16_t*)v, ld, (const bf16_t*)dout, ldo, lse, delta_ws, key_mask, (bf16_t*)dk, (bf16_t*)dv, ldd, S, Sp, scale, ag)
            if (attn_pipe_on() >= 2) DKV_LAUNCH(2);
            else if (attn_pipe_on() == 1) DKV_LAUNCH(1);
            else DKV_LAUNCH(0);
#undef DKV_LAUNCH
        }
    } else if (causal) {
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
extern "C" int vlr_attn_bwd(const void* q, const void* k, const void* v, int ld, const void* o, const void* dout,
                            int ldo, const float* lse, float* delta_ws, const int* key_mask, void* dq, void* dk,
                            void* dv, int ldd, int batch, int S, int heads, int head_dim, int causal, float scale,
                            hipStream_t st) {
    return vlr_attn_bwd_gqa(q, k, v, ld, o, dout, ldo, lse, delta_ws, key_mask, dq, dk, dv, ldd, batch, S, heads, heads, head_dim,
                            causal, scale, st);
}
