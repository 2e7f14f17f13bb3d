// LoRA input-gradient term of the targets that share one input, streamed:  dx [M][in] (+)= alpha * sum_t keep_t . (v_t [M][r] . A_t [r][in])
// (q, k, v -> d xn1; gate, up -> d xn2; o -> d attn; down -> d act;  keep_t = the lora_dropout mask of target t, alpha = scaling / (1 - p)).
//
// The reduction is the adapter rank (r = 64..256): the launch is ONE read-modify-write pass over dx with a few MFMAs per element - HBM
// bound (210 MB at [12792 x 4096]: 27 us at 8 TB/s).  gemm.hip ran it as a 128x128-tile GEMM with the mask in the epilogue: a workgroup
// per tile, its loads - MFMAs - read C - write C chain exposed end to end (85-120 us per target and pass, tools/lora_gemm_bench.py).
// Here a workgroup owns 64 ROWS and walks a range of 128-column tiles:
//   * its v rows live in REGISTERS for the whole walk, already in MFMA fragment form (NT targets x 2 row tiles x r/32 k slices);
//   * the [r][128] slices of A_t (L2 resident: A_t is 1-3 MB) stream through a double-buffered LDS stage by global_load_lds, one
//     (column tile, target) unit ahead of the MFMAs; the dx tile and the packed keep masks of the next unit are prefetched into
//     registers the same way.  Every unit starts with s_waitcnt vmcnt(0) + one barrier - no counted waits (this wave's stores of the
//     previous tile are in the same queue and stores retire out of order with loads);
//   * per unit 16 x r/32 v_mfma_f32_16x16x32_bf16 per wave (4 waves = 2 x 2, wave tile 32 x 64); the mask is applied to the fp32
//     accumulators (a lane holds 4 consecutive columns of a row: one nibble of the packed mask, or one hash), the targets are summed in
//     registers and the bf16 tile is read / written once, 8 bytes per lane.
// Grid = (row blocks, column splits): ~4 workgroups per CU, 2-4 resident per CU (LDS 2 x r/64 x 16 KiB), so one workgroup's wait is
// another one's MFMAs.  (r = 256 leaves ONE workgroup per CU; cutting its units in two K halves to fit two was measured slower - 145 us
// against 112 us at [13888 x 4096]: twice the units, and every unit pays a vmcnt(0) + barrier.)
#include <stdlib.h>

#include <type_traits>

#include "gemm.h"

typedef __attribute__((address_space(3))) void dx_lvoid_t;
typedef __attribute__((ext_vector_type(4))) short dx_s16x4_t;
typedef __attribute__((address_space(3))) dx_s16x4_t dx_lds_s16x4_t;

struct LoraDxParams {
    const bf16_t* v; int ldv;
    const bf16_t* A;             // [NT * r][in]
    bf16_t* dx;
    int M, in, r;
    float alpha;
    int accumulate;
    const unsigned char* bits;   // packed keep masks (target t at bits + t * gbits) or null: hash (key of target t = mix64(seed + t))
    long gbits;
    uint64_t seed;
    uint32_t thr;
    int ct_per_wg;               // column tiles per workgroup (blockIdx.y walks the ranges)
    const unsigned char* rowskip; // [M] or null: a 64-row slab whose bytes are all 0 holds zero v rows (PLoRA: text rows) - dx += 0, nothing to do
};

__device__ __forceinline__ void dx_dma16_s(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// K-strided operand tile [64 k][256 B] (gemm256p.hip / gemm128p.hip image): fragment of 16 columns x 32 k
__device__ __forceinline__ bf16x8 dx_frag_ks(const char* tile, int cbase, int s, int lane) {
    const int g = lane >> 4, pq = lane & 15;
    const int krow = s * 32 + g * 8 + (pq >> 2);
    const int col = cbase + (pq & 3) * 4;
    const int swz = ((krow & 3) << 2) | (((krow >> 3) & 1) << 1);
    const int off = krow * 256 + (((col >> 3) ^ swz) << 4) + ((col >> 2) & 1) * 8;
    const dx_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dx_lds_s16x4_t*)(tile + off));
    const dx_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dx_lds_s16x4_t*)(tile + off + 4 * 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

#ifndef VLR_DX_WIDE
#define VLR_DX_WIDE 1      // the dx tile is read and written as 16-byte accesses of 16 consecutive columns per lane (round 5); 0: 8 bytes = 4 columns per lane and tile
#endif
// x[j] (j = 0..3) holds, in the lanes of 16-lane row q, element (j, q) of a 4 x 4 matrix per lane column; afterwards x[c] holds element
// (q, c): the transpose across the four lane rows.  Two exchange stages (v_permlane16_swap: odd rows of the first operand <-> even rows of
// the second; v_permlane32_swap: upper half of the first <-> lower half of the second) as INLINE ASM - hipcc 7.2 folds several calls of the
// builtins on related values into one (gemm256p.hip history).  Its own inverse.
__device__ __forceinline__ void dx_xpose4(uint32_t (&x)[4]) {
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[0]), "+v"(x[1]));
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[2]), "+v"(x[3]));
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[0]), "+v"(x[2]));
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[1]), "+v"(x[3]));
}

// NT targets, KT = r / 64 K tiles per target, MASK: 0 no dropout, 1 keep masks (packed bits when p.bits, else hashed)
// RW = row waves: 2 = a workgroup owns 64 rows (4 waves), 4 = 128 rows (8 waves share every A slice: half the LDS-DMA per row of dx -
// round 6: LDS-DMA is bound at ~40 B/clk per CU, and a unit stages 16 r / 64 KiB of A for 16 KiB of dx)
template <int NT, int KT, int MASK, int RW = 2>
__global__ __launch_bounds__(RW * 128) void lora_dx_kernel(LoraDxParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 units x KT x 16 KiB | 2 x (RW / 2) KiB packed-mask tiles
    constexpr int ROWS = 32 * RW, NWAVE = 2 * RW, NPW = 16 / NWAVE;  // rows per workgroup, waves, A-slice pieces per wave and K tile
    constexpr int KS = 2 * KT;                                        // k slices of 32 per target
    constexpr int UNIT_BYTES = KT * 16384;
    const int t_ = threadIdx.x;
    const int lane = t_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lm = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.x * ROWS;
    const int tiles_n = p.in >> 7;
    const int c_lo = blockIdx.y * p.ct_per_wg;
    const int c_hi = min(tiles_n, c_lo + p.ct_per_wg);
    if (c_lo >= c_hi) return;
    if (p.rowskip) {              // (accumulate launches only: the host passes it with accumulate = 1)
        bool any = false;
#pragma unroll
        for (int h = 0; h < ROWS / 64; ++h) {
            const int row = m0 + h * 64 + lane;
            any = any || (row < p.M && p.rowskip[row] != 0);
        }
        if (__ballot(any) == 0) return;
    }
    const int nunits = (c_hi - c_lo) * NT;

    // ---- v rows of this wave as MFMA fragments (lane: row lm of the 16-row tile, k (lane >> 4) * 8 .. + 8 of the slice)
    bf16x8 vf[NT][2][KS];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = m0 + wr * 32 + i * 16 + lm;
            row = row < p.M ? row : p.M - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                vf[t][i][ks] = *reinterpret_cast<const bf16x8*>(p.v + (size_t)row * p.ldv + t * p.r + ks * 32 + lq * 8);
        }
    // ---- DMA of one unit = the [r][128] slice of A_t: KT tiles of [64 k][256 B], 4 wave instructions per tile and wave
    uint32_t offB[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int r_ = (wave + NWAVE * i) * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ (((r_ & 3) << 2) | (((r_ >> 3) & 1) << 1));
        offB[i] = (uint32_t)(((size_t)r_ * p.in + chunk * 8) * 2);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(dx_lvoid_t*)smem + wave * 1024;
    auto stage_unit = [&](int u) {            // u = (c - c_lo) * NT + t
        const int c = c_lo + u / NT, t = u % NT;
        const uint32_t l = lds0 + (u & 1) * UNIT_BYTES;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const char* b = reinterpret_cast<const char*>(p.A + ((size_t)(t * p.r + kt * 64)) * p.in + c * 128);
#pragma unroll
            for (int i = 0; i < NPW; ++i) dx_dma16_s(b, offB[i], l + kt * 16384 + i * NWAVE * 1024);
        }
    };
    // rows / first column of this lane's 2 x 4 accumulator tiles
    int rowv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) rowv[i] = m0 + wr * 32 + i * 16 + lm;
    const int cl = wc * 64 + 4 * lq;           // + j * 16 inside the column tile
    // prefetched per-unit / per-tile registers
    u32x2 dxr[2][4];                           // the dx tile of the CURRENT column tile (accumulate)
    uint32_t kb[2][4];                         // keep nibbles of the CURRENT unit
    // VLR_DX_WIDE: lane (lm, lq) reads / writes the 16 consecutive columns of 16-column tile j = lq of its row - two 16-byte accesses, a row's
    // four lanes = one 128-byte line - where the accumulator layout (4 columns of each of the four tiles) made it four 8-byte accesses of 32
    // bytes per row and instruction.  d[i][j] stays in the ACCUMULATOR layout for the arithmetic: the loaded words are regrouped by
    // dx_xpose4 (word w of the lane's 16 columns = (column group w >> 1, half w & 1)), the rounded results regrouped back before the store.
    const int clw = wc * 64 + 16 * lq;         // first of this lane's 16 consecutive columns inside the column tile
    auto load_dx = [&](int c, u32x2 (&d)[2][4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#if VLR_DX_WIDE
            u32x4 a = {0u, 0u, 0u, 0u}, b = a;
            if (rowv[i] < p.M) {
                const bf16_t* src = p.dx + (size_t)rowv[i] * p.in + c * 128 + clw;
                a = *reinterpret_cast<const u32x4*>(src);
                b = *reinterpret_cast<const u32x4*>(src + 8);
            }
            d[i][0] = u32x2{a[0], a[1]}; d[i][1] = u32x2{a[2], a[3]}; d[i][2] = u32x2{b[0], b[1]}; d[i][3] = u32x2{b[2], b[3]};      // (column group g of tile lq) - regrouped at use
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x2 w = {0u, 0u};
                if (rowv[i] < p.M) w = *reinterpret_cast<const u32x2*>(p.dx + (size_t)rowv[i] * p.in + c * 128 + cl + j * 16);
                d[i][j] = w;
            }
#endif
        }
    };
    // keep masks of unit u.  Packed bits: wave 0 moves the [64 rows][16 B] tile of (column tile, target) into LDS with ONE LDS-DMA
    // instruction (lane = row) a unit ahead - visible to every wave after the next vmcnt(0) + barrier - and a lane reads its eight nibbles
    // from there (the first version loaded them as 8 scattered global bytes per lane: 86 us of the 196 us of the q, k, v launch).
    // Hash form: computed into registers a unit ahead.
    const uint32_t lbits = (uint32_t)(uintptr_t)(dx_lvoid_t*)smem + 2 * UNIT_BYTES;
    auto stage_keep = [&](int u) {
        if constexpr (MASK) {
            if (p.bits && wave < ROWS / 64) {
                const int c = c_lo + u / NT, t = u % NT;
                int row = m0 + wave * 64 + lane;
                row = row < p.M ? row : p.M - 1;
                const unsigned char* g = p.bits + (size_t)t * p.gbits + (((size_t)row * p.in + c * 128) >> 3);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lbits + (u & 1) * (ROWS * 16) + wave * 1024) : "memory", "m0");
            }
        }
    };
    auto hash_keep = [&](int u, uint32_t (&k)[2][4]) {
        if constexpr (MASK) {
            if (!p.bits) {
                const int c = c_lo + u / NT, t = u % NT;
                const uint64_t key = vlr_mix64(p.seed + (uint64_t)t);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const long idx = (long)rowv[i] * p.in + c * 128 + cl + j * 16;
                        const uint64_t rr = vlr_mix64(key ^ (uint64_t)(2 * (idx >> 3) + ((idx >> 2) & 1)));
                        uint32_t keep = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) keep |= (uint32_t)(((rr >> (16 * e)) & 0xffffu) >= p.thr) << e;
                        k[i][j] = keep << (cl & 4);       // same position as the packed byte's nibble
                    }
            }
        }
    };
    f32x4 sum[2][4];
    stage_unit(0);
    if (p.accumulate) load_dx(c_lo, dxr);
    stage_keep(0);
    hash_keep(0, kb);
    for (int u = 0; u < nunits; ++u) {
        const int c = c_lo + u / NT, t = u % NT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < nunits) stage_unit(u + 1);
        // prefetch for the next unit / the next column tile (used after the next vmcnt(0))
        uint32_t kb_n[2][4];
        u32x2 dx_n[2][4];
        const bool last_t = t == NT - 1;
        if (u + 1 < nunits) {
            stage_keep(u + 1);
            hash_keep(u + 1, kb_n);
            if (last_t && p.accumulate) load_dx(c + 1, dx_n);
        }
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* tile = smem + (u & 1) * UNIT_BYTES;
        const unsigned char* bt_lds = reinterpret_cast<const unsigned char*>(smem) + 2 * UNIT_BYTES + (u & 1) * (ROWS * 16);
        // (static dispatch on the target: the v fragments are a register array)
        auto run = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 fb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = dx_frag_ks(tile + (ks >> 1) * 16384, wc * 64 + j * 16, ks & 1, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], vf[T][i][ks], acc[i][j], 0, 0, 0);
            }
        };
        if constexpr (NT == 1) run(std::integral_constant<int, 0>{});
        else if constexpr (NT == 2) { if (t == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{}); }
        else { if (t == 0) run(std::integral_constant<int, 0>{}); else if (t == 1) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 2>{}); }
        // mask + sum over the targets
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (MASK) {
                    const uint32_t kbyte = p.bits ? (uint32_t)bt_lds[(wr * 32 + i * 16 + lm) * 16 + ((cl + j * 16) >> 3)] : kb[i][j];
                    const uint32_t keep = kbyte >> (cl & 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum[i][j][e] += ((keep >> e) & 1) ? p.alpha * acc[i][j][e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum[i][j][e] += p.alpha * acc[i][j][e];
                }
            }
        if (last_t) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#if VLR_DX_WIDE
                // loaded words: lane row q holds (tile q, column group g) in d[i][g]; the accumulators hold (tile j, column group q) in sum[i][j]:
                // transpose the loaded words to the accumulator layout, add in fp32, round, transpose back, store 2 x 16 bytes
                uint32_t lo[4], hi[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) { lo[g] = dxr[i][g][0]; hi[g] = dxr[i][g][1]; }
                if (p.accumulate) { dx_xpose4(lo); dx_xpose4(hi); }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 o = sum[i][j];
                    if (p.accumulate) { o[0] += bf16lo(lo[j]); o[1] += bf16hi(lo[j]); o[2] += bf16lo(hi[j]); o[3] += bf16hi(hi[j]); }
                    lo[j] = pack_bf16(o[0], o[1]);
                    hi[j] = pack_bf16(o[2], o[3]);
                }
                dx_xpose4(lo); dx_xpose4(hi);
                if (rowv[i] < p.M) {
                    bf16_t* dst = p.dx + (size_t)rowv[i] * p.in + c * 128 + clw;
                    *reinterpret_cast<u32x4*>(dst) = u32x4{lo[0], hi[0], lo[1], hi[1]};
                    *reinterpret_cast<u32x4*>(dst + 8) = u32x4{lo[2], hi[2], lo[3], hi[3]};
                }
#else
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 o = sum[i][j];
                    if (p.accumulate) {
                        const u32x2 w = dxr[i][j];
                        o[0] += bf16lo(w[0]); o[1] += bf16hi(w[0]); o[2] += bf16lo(w[1]); o[3] += bf16hi(w[1]);
                    }
                    u32x2 w;
                    w[0] = pack_bf16(o[0], o[1]);
                    w[1] = pack_bf16(o[2], o[3]);
                    if (rowv[i] < p.M) *reinterpret_cast<u32x2*>(p.dx + (size_t)rowv[i] * p.in + c * 128 + cl + j * 16) = w;
                }
#endif
            }
        }
        if (u + 1 < nunits) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kb[i][j] = kb_n[i][j];
                    if (last_t && p.accumulate) dxr[i][j] = dx_n[i][j];
                }
        }
    }
}

template <int NT, int KT, int RW>
static void dx_launch_m(const LoraDxParams& p, int mask, dim3 grid, hipStream_t stream) {
    const int lds = 2 * KT * 16384 + 2 * (32 * RW * 16);
    if (mask) {
        static bool a1 = false;
        if (!a1) { hipFuncSetAttribute((const void*)lora_dx_kernel<NT, KT, 1, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); a1 = true; }
        hipLaunchKernelGGL((lora_dx_kernel<NT, KT, 1, RW>), grid, dim3(RW * 128), lds, stream, p);
    } else {
        static bool a0 = false;
        if (!a0) { hipFuncSetAttribute((const void*)lora_dx_kernel<NT, KT, 0, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); a0 = true; }
        hipLaunchKernelGGL((lora_dx_kernel<NT, KT, 0, RW>), grid, dim3(RW * 128), lds, stream, p);
    }
}
// (8-wave workgroups hold two waves per SIMD = 256 registers each: the v fragments of NT x KT > 4 do not fit - those shapes keep 64 rows)
template <int NT, int KT> constexpr bool dx_fits128() { return NT * KT <= 4; }
template <int NT, int RW>
static bool dx_launch_k(const LoraDxParams& p, int mask, dim3 grid, hipStream_t stream) {
    switch (p.r) {
        case 64: dx_launch_m<NT, 1, RW>(p, mask, grid, stream); return true;
        case 128:
            if constexpr (RW == 2 || dx_fits128<NT, 2>()) { dx_launch_m<NT, 2, RW>(p, mask, grid, stream); return true; }
            return false;
        case 256:
            if constexpr (NT <= 2 && (RW == 2 || dx_fits128<NT, 4>())) { dx_launch_m<NT, 4, RW>(p, mask, grid, stream); return true; }      // (3 x 2 x 8 v fragments do not fit the register file)
            return false;
        default: return false;
    }
}

// dx (+)= scale / (1 - p) * sum_t keep_t . (v_t A_t) on the streaming kernel; false: a shape it does not take (the caller runs the
// tile kernels of gemm.hip): n <= 3 targets, r in {64, 128, 256}, in % 128 == 0, 16-byte aligned operands.  VLR_LORA_DX=0 disables.
bool vlr_lora_dx_try_launch(int n, const void* v, int ldv, const void* A, void* dx, int M, int in, int r, float p_drop, uint64_t seed,
                            float scale, int accumulate, const void* bits, long bits_gstride, hipStream_t stream, const unsigned char* rowskip) {
    static int on = -1, wg_per_cu = 4;      // 4 workgroups per CU in the grid (2-4 resident): 196 / 88 / 123 / 255 us for the four groups at the 7B shapes against 222 / 106 / 147 / 303 at 2
    if (on < 0) {
        const char* e = getenv("VLR_LORA_DX");
        on = (e && e[0] == '0') ? 0 : 1;
        const char* w = getenv("VLR_LORA_DX_WGS");
        if (w && atoi(w) >= 1 && atoi(w) <= 8) wg_per_cu = atoi(w);
    }
    if (!on || n < 1 || n > 3 || in % 128 != 0 || (r != 64 && r != 128 && r != 256) || ldv % 8 != 0 || M < 1) return false;
    if (((uintptr_t)v | (uintptr_t)A | (uintptr_t)dx) & 15) return false;
    LoraDxParams q;
    q.v = (const bf16_t*)v; q.ldv = ldv; q.A = (const bf16_t*)A; q.dx = (bf16_t*)dx; q.M = M; q.in = in; q.r = r;
    q.alpha = scale / (1.f - p_drop); q.accumulate = accumulate;
    q.bits = (const unsigned char*)bits; q.gbits = bits_gstride; q.seed = seed; q.thr = vlr_dropout_thr(p_drop);
    q.rowskip = accumulate ? rowskip : nullptr;
    // 128-row workgroups (8 waves share every A slice) where the registers allow it (n x r / 64 <= 4) AND it measured faster
    // (tools/lora_gemm_bench.py --reps 20, round 6): rank 256 (InternLM-XComposer2's PLoRA: qkv 88.8 -> 78.0, o 88.6 -> 77.2, down 386 -> 258 us)
    // and wide inputs (LLaVA down_proj, in = 11008: 178.7 -> 167.7 us); at r = 128 / in = 4096 the 64-row workgroups win (o 51.0 against
    // 57.9, gate | up 77.7 against 93.7 us: two independent workgroups per CU de-phase, eight waves behind one barrier do not).
    // VLR_LORA_DX_ROWS=64 | 128 forces one form (A/B).
    static int rows_env = -1;
    if (rows_env < 0) { const char* e = getenv("VLR_LORA_DX_ROWS"); rows_env = e ? atoi(e) : 0; }
    const int kt_ = r / 64;
    const bool rows128 = n * kt_ <= 4 && (rows_env == 128 || (rows_env != 64 && (kt_ == 4 || in >= 8192)));
    const int rows = rows128 ? 128 : 64;
    const int rb = (M + rows - 1) / rows, tiles_n = in / 128;
    static bool wgs_forced = getenv("VLR_LORA_DX_WGS") != nullptr;
    const int wpc = (rows128 && in < 8192 && !wgs_forced) ? 2 : wg_per_cu;      // 128-row workgroups at in = 4096: 2 per CU in the grid (73 / 71 us against 77 at 4, InternLM qkv / o)
    int splits = (wpc * vlr_compute_cus() + rb - 1) / rb;
    if (splits < 1) splits = 1;
    if (splits > tiles_n) splits = tiles_n;
    q.ct_per_wg = (tiles_n + splits - 1) / splits;
    splits = (tiles_n + q.ct_per_wg - 1) / q.ct_per_wg;
    const dim3 grid(rb, splits);
    const int mask = p_drop > 0.f ? 1 : 0;
    if (rows128) {
        if (n == 1) return dx_launch_k<1, 4>(q, mask, grid, stream);
        if (n == 2) return dx_launch_k<2, 4>(q, mask, grid, stream);
        return dx_launch_k<3, 4>(q, mask, grid, stream);
    }
    if (n == 1) return dx_launch_k<1, 2>(q, mask, grid, stream);
    if (n == 2) return dx_launch_k<2, 2>(q, mask, grid, stream);
    return dx_launch_k<3, 2>(q, mask, grid, stream);
}
