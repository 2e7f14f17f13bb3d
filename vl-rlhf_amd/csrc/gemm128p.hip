// 128x128x64-tile bf16 MFMA GEMM on an LDS-DMA ring: the small-grid path of vlr_gemm_bf16 - the peeled last tile rows of the decoder
// GEMMs, the split-K slices of the LoRA adapter products, everything with too few 256x256 tiles to fill the chip.
//
// Its predecessor (gemm.hip: gemm_bf16_kernel) stages global -> registers -> LDS with ONE K tile of loads in flight per workgroup; with
// a 128x128 tile the MFMA work of a K tile (512 cycles per SIMD) covers a fraction of the load latency and the kernel ran at
// 240-390 TF/s whatever the shape (tools/lora_gemm_bench.py, profiles/r03_*).  Here both operands travel global -> LDS by
// global_load_lds_dwordx4 (no staging registers), through a ring of NST stages of 32 KiB (A tile 16 KiB + B tile 16 KiB):
//   iteration kt:  s_waitcnt vmcnt(8 * younger)   - this wave's part of K tile kt has landed (8 DMA instructions per tile per wave;
//                                                    the tiles issued after it stay in flight)
//                  s_barrier                      - everybody's part has, and everybody has finished reading K tile kt - 1
//                  issue K tile kt + NST - 1 into the stage K tile kt - 1 occupied
//                  2 x (8 fragment reads, 16 v_mfma_f32_16x16x32_bf16)
// NST - 1 tiles (96 KiB per CU at NST = 4, one workgroup per CU) are in flight under the MFMAs: enough bytes to hold HBM busy on the
// skinny shapes and most of the latency covered on the deep ones.  4 waves = 2 x 2, wave tile 64 x 64 (16 accumulator tiles of
// 16 x 16): 16 KiB of fragment reads per 32 MFMAs - the lowest LDS traffic a 128 x 128 tile allows.
// LDS images and fragment reads are those of gemm256p.hip (a "half tile" there = an operand tile here): K-contiguous operands
// [128 rows][128 B] with 16-byte chunk c of row r at c ^ ((r >> 1) & 7); K-strided operands (stored [K][cols]) [64 k][256 B] with chunk c
// of k-row r at c ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1)), read by ds_read_b64_tr_b16.  Rows / columns beyond the edge are clamped on
// the source side and never stored; K chunks beyond K read a 16-byte zero buffer.  Epilogue, split-K slices and grouped launches:
// as gemm_bf16_kernel (gemm128_epilogue.h).
#include <stdlib.h>

#include "gemm.h"
#include "gemm128_epilogue.h"

#define QT 128
#define QK 64
#define Q_OPER_BYTES (QT * QK * 2)          // 16 KiB
#define Q_STAGE_BYTES (2 * Q_OPER_BYTES)    // A | B
#define Q_SS 68                             // floats per row of a wave's fp32 epilogue stage (64 + 4: conflict-free 16-byte writes)
#define Q_EPI_BYTES (4 * 64 * Q_SS * 4)     // 68 KiB

typedef __attribute__((address_space(3))) void q_lvoid_t;
typedef __attribute__((ext_vector_type(4))) short q_s16x4_t;
typedef __attribute__((address_space(3))) q_s16x4_t q_lds_s16x4_t;

// one LDS-DMA wave instruction (64 lanes x 16 B -> LDS [m0 + lane * 16]); inline asm for the reason given in gemm256p.hip: the
// compiler must not see it, or it drains the DMA queue in front of every ds_read
__device__ __forceinline__ void q_dma16(const void* g, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void q_dma16_s(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// 64 lanes x 4 B -> LDS [m0 + lane * 4]
__device__ __forceinline__ void q_dma4(const void* g, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}
#define Q_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// fragment of v_mfma_f32_16x16x32_bf16 (16 rows x 32 k of slice s; lane l: row l & 15, k (l >> 4) * 8 .. + 8)
__device__ __forceinline__ bf16x8 q_frag_kc(const char* tile, int rbase, int s, int lane) {
    const int row = rbase + (lane & 15);
    const int chunk = s * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
__device__ __forceinline__ bf16x8 q_frag_ks(const char* tile, int cbase, int s, int lane) {
    const int g = lane >> 4, pq = lane & 15;
    const int krow = s * 32 + g * 8 + (pq >> 2);
    const int col = cbase + (pq & 3) * 4;
    const int swz = ((krow & 3) << 2) | (((krow >> 3) & 1) << 1);
    const int off = krow * 256 + (((col >> 3) ^ swz) << 4) + ((col >> 2) & 1) * 8;
    const q_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q_lds_s16x4_t*)(tile + off));
    const q_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q_lds_s16x4_t*)(tile + off + 4 * 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// MASK = 1 (NT only): the rows of the K-contiguous A operand are x [rows][mask_ld] under lora_dropout - the packed keep mask of the K tile
// (128 rows x 8 bytes, vlr_dropout_bits) rides the ring as a ninth DMA instruction per wave (global_load_lds_dword: 32 rows x 8 B each)
// and a fragment (8 consecutive k of one row = ONE mask byte) is masked when it is read from LDS: no dropped copy of x, no hash, and
// the operand still travels by LDS-DMA.  K % 64 == 0 (whole tiles), mask_ld % 64 == 0.
// MASK = 2 (TN only): the K-strided B operand is x stored [K = rows][mask_ld] (dA = v^T (mask . x)); a fragment is 8 consecutive ROWS of
// one column = one byte of the K-tile-blocked transposed masks of vlr_dropout_bits2 ([row / 64][col][8 B]); the 128 columns x 8 B
// of a K tile are 1 KiB contiguous: 256 B per wave by global_load_lds_dword.  N % 128 == 0, K slices start on whole tiles.
template <bool A_KS, bool B_KS, int MASK = 0>
__global__ __launch_bounds__(256) void gemm128p_kernel(GemmParams p, const bf16_t* __restrict__ zero16, int nst) {
    static_assert(MASK == 0 || (MASK == 1 && !A_KS && !B_KS) || (MASK == 2 && A_KS && B_KS), "masked operand: 1 = NT A rows, 2 = TN B rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // max(nst * 32 KiB, Q_EPI_BYTES)
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    if (p.groups > 1) {            // grouped launch: this workgroup's problem
        const int g = blockIdx.z;
        p.A += (size_t)g * p.gA;
        p.B += (size_t)g * p.gB;
        p.C = (char*)p.C + (size_t)g * p.gC * (p.out_f32 ? 4 : 2);
        if (p.splitk > 1) p.part += (size_t)g * p.splitk * p.M * p.N;
        if (MASK) p.mask_bits += (size_t)g * p.gMask;
    }
    int kabs0 = 0;                 // split-K slices move p.A: the mask index needs the absolute k
    // TN with a K-tile list (GemmParams::ktlist): the K loop walks list entries [kt0, kt0 + K / 64) instead of consecutive tiles
    const int* ktl = nullptr;
    int kt0 = 0;
    if constexpr (A_KS && B_KS) ktl = p.ktlist;
    if (ktl) {
        const int nall = ktl[0];
        int n = nall;
        if (p.splitk > 1) {
            const int per = (nall + p.splitk - 1) / p.splitk;
            kt0 = (int)blockIdx.y * per;
            n = min(per, nall - kt0);
            n = n > 0 ? n : 0;      // (an empty slice still writes its zero partial)
        }
        p.K = n * QK;
    }
    if (p.splitk > 1) {            // split-K slice z: raw alpha * acc -> its own fp32 partial
        const int z = blockIdx.y, k0 = z * p.kchunk;
        if (!ktl) {
            kabs0 = k0;
            p.A += A_KS ? (size_t)k0 * p.lda : (size_t)k0;
            p.B += B_KS ? (size_t)k0 * p.ldb : (size_t)k0;
            p.K = min(p.kchunk, p.K - k0);
        }
        p.C = p.part + (size_t)z * p.M * p.N;
        p.ldc = p.N; p.out_f32 = 1; p.bias = nullptr; p.residual = nullptr; p.accumulate = 0; p.act = 0;
    }
    // ---- XCD-aware, grouped tile map (as gemm_bf16_kernel)
    const int tiles_m = (p.M + QT - 1) / QT, tiles_n = (p.N + QT - 1) / QT;
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        const int q = nwg >> 3, rem = nwg & 7;
        pid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int first_m = (pid / per_group) * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int m0 = (first_m + (pid % per_group) % gsz) * QT;
    const int n0 = ((pid % per_group) / gsz) * QT;
    if (!A_KS && p.rowskip) {      // no marked row in this tile's 128 rows: the caller zeroes them (GemmParams::rowskip)
        const int r0 = m0 + lane, r1 = m0 + 64 + lane;
        const bool any = (r0 < p.M && p.rowskip[r0] != 0) || (r1 < p.M && p.rowskip[r1] != 0);
        if (__ballot(any) == 0) return;
    }

    // ---- per-lane source offsets (bytes) of this wave's four DMA pieces of each operand tile, relative to the K tile's base
    uint32_t offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (A_KS) {
            const int r = (wave + 4 * i) * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1));
            int col = m0 + chunk * 8;
            col = col + 8 <= p.M ? col : p.M - 8;
            offA[i] = (uint32_t)(((size_t)r * p.lda + col) * 2);
        } else {
            const int r = (wave + 4 * i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int grow = m0 + r;
            grow = grow < p.M ? grow : p.M - 1;
            offA[i] = (uint32_t)(((size_t)(grow - m0) * p.lda + c * 8) * 2);
        }
        if constexpr (B_KS) {
            const int r = (wave + 4 * i) * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1));
            int col = n0 + chunk * 8;
            col = col + 8 <= p.N ? col : p.N - 8;
            offB[i] = (uint32_t)(((size_t)r * p.ldb + col) * 2);
        } else {
            const int r = (wave + 4 * i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int grow = n0 + r;
            grow = grow < p.N ? grow : p.N - 1;
            offB[i] = (uint32_t)(((size_t)(grow - n0) * p.ldb + c * 8) * 2);
        }
    }
    // uniform bases: K-contiguous operands start at the tile's first row, K-strided ones at the operand (their column rides in the offset)
    const char* baseA = reinterpret_cast<const char*>(A_KS ? p.A : p.A + (size_t)m0 * p.lda);
    const char* baseB = reinterpret_cast<const char*>(B_KS ? p.B : p.B + (size_t)n0 * p.ldb);
    const size_t stepA = A_KS ? (size_t)QK * p.lda * 2 : (size_t)QK * 2;
    const size_t stepB = B_KS ? (size_t)QK * p.ldb * 2 : (size_t)QK * 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(q_lvoid_t*)smem + wave * 1024;
    const uint32_t lbits = (uint32_t)(uintptr_t)(q_lvoid_t*)smem + nst * Q_STAGE_BYTES;      // mask tiles behind the operand stages
    const int nt = (p.K + QK - 1) / QK;

    auto stage_bits = [&](int kt, int slot, int ka) {
        if constexpr (MASK == 1) {      // lane -> (row wave*32 + lane/2, 4-byte half lane&1), LDS [128 rows][8 B]
            int grow = m0 + wave * 32 + (lane >> 1);
            grow = grow < p.M ? grow : p.M - 1;
            const unsigned char* g = p.mask_bits + (((size_t)grow * p.mask_ld + kabs0 + kt * QK) >> 3) + (lane & 1) * 4;
            q_dma4(g, lbits + slot * 1024 + wave * 256);
        } else if constexpr (MASK == 2) {   // [128 columns][8 B] of K tile (kabs0 / 64 + kt): 1 KiB contiguous
            const unsigned char* g = p.mask_bits + ((size_t)(ktl ? ka : (kabs0 >> 6) + kt) * p.mask_ld + n0) * 8 + wave * 256 + lane * 4;
            q_dma4(g, lbits + slot * 1024 + wave * 256);
        }
    };
    auto stage_tile = [&](int kt, int slot) {
        const uint32_t la = lds0 + slot * Q_STAGE_BYTES, lb = la + Q_OPER_BYTES;
        const int ka = ktl ? __builtin_amdgcn_readfirstlane(ktl[1 + kt0 + kt]) : kt;      // the K tile this ring entry holds
        if ((kt + 1) * QK <= p.K) {
            const char* ba = baseA + (size_t)ka * stepA;
            const char* bb = baseB + (size_t)ka * stepB;
#pragma unroll
            for (int i = 0; i < 4; ++i) q_dma16_s(ba, offA[i], la + i * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i) q_dma16_s(bb, offB[i], lb + i * 4096);
            if constexpr (MASK) stage_bits(kt, slot, ka);
        } else {
            // the last, partial K tile: chunks / k-rows beyond K come from the zero buffer
            const int k0 = kt * QK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* g;
                if constexpr (A_KS) {
                    const int r = (wave + 4 * i) * 4 + (lane >> 4);
                    g = (k0 + r < p.K) ? baseA + (size_t)kt * stepA + offA[i] : reinterpret_cast<const char*>(zero16);
                } else {
                    const int r = (wave + 4 * i) * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((r >> 1) & 7);
                    g = (k0 + c * 8 + 8 <= p.K) ? baseA + (size_t)kt * stepA + offA[i] : reinterpret_cast<const char*>(zero16);
                }
                q_dma16(g, la + i * 4096);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* g;
                if constexpr (B_KS) {
                    const int r = (wave + 4 * i) * 4 + (lane >> 4);
                    g = (k0 + r < p.K) ? baseB + (size_t)kt * stepB + offB[i] : reinterpret_cast<const char*>(zero16);
                } else {
                    const int r = (wave + 4 * i) * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((r >> 1) & 7);
                    g = (k0 + c * 8 + 8 <= p.K) ? baseB + (size_t)kt * stepB + offB[i] : reinterpret_cast<const char*>(zero16);
                }
                q_dma16(g, lb + i * 4096);
            }
            if constexpr (MASK) stage_bits(kt, slot, ka);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    // ---- prologue: the first nst - 1 K tiles in flight
    for (int s = 0; s < nst - 1 && s < nt; ++s) stage_tile(s, s);
    int slot = 0;                       // stage of K tile kt
    int fill = nst - 1;                 // stage K tile kt + nst - 1 goes to
    for (int kt = 0; kt < nt; ++kt) {
        const int younger = min(nst - 2, nt - 1 - kt);     // tiles issued after kt and still allowed in flight
        if constexpr (MASK) {          // nine DMA instructions per tile and wave
            if (younger >= 2) Q_WAIT_VM(18);
            else if (younger == 1) Q_WAIT_VM(9);
            else Q_WAIT_VM(0);
        } else {
            if (younger >= 2) Q_WAIT_VM(16);
            else if (younger == 1) Q_WAIT_VM(8);
            else Q_WAIT_VM(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt + nst - 1 < nt) stage_tile(kt + nst - 1, fill);
        const char* ta = smem + slot * Q_STAGE_BYTES;
        const char* tb = ta + Q_OPER_BYTES;
        const unsigned char* tbits = reinterpret_cast<const unsigned char*>(smem) + nst * Q_STAGE_BYTES + slot * 1024;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (A_KS) fa[i] = q_frag_ks(ta, wr * 64 + i * 16, ks, lane);
                else fa[i] = q_frag_kc(ta, wr * 64 + i * 16, ks, lane);
                if constexpr (MASK == 1) {
                    const uint32_t keep = tbits[(wr * 64 + i * 16 + (lane & 15)) * 8 + ks * 4 + (lane >> 4)];
                    u32x4 w = __builtin_bit_cast(u32x4, fa[i]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] &= ((keep >> (2 * e)) & 1 ? 0x0000ffffu : 0u) | ((keep >> (2 * e + 1)) & 1 ? 0xffff0000u : 0u);
                    fa[i] = __builtin_bit_cast(bf16x8, w);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (B_KS) fb[j] = q_frag_ks(tb, wc * 64 + j * 16, ks, lane);
                else fb[j] = q_frag_kc(tb, wc * 64 + j * 16, ks, lane);
                if constexpr (MASK == 2) {      // lane: column wc*64 + j*16 + (lane & 15), rows ks*32 + (lane >> 4)*8 .. + 7
                    const uint32_t keep = tbits[(wc * 64 + j * 16 + (lane & 15)) * 8 + ks * 4 + (lane >> 4)];
                    u32x4 w = __builtin_bit_cast(u32x4, fb[j]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] &= ((keep >> (2 * e)) & 1 ? 0x0000ffffu : 0u) | ((keep >> (2 * e + 1)) & 1 ? 0xffff0000u : 0u);
                    fb[j] = __builtin_bit_cast(bf16x8, w);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        slot = slot + 1 == nst ? 0 : slot + 1;
        fill = fill + 1 == nst ? 0 : fill + 1;
    }
    // ---- epilogue through LDS: per-wave 64 x 64 fp32 stage [64][Q_SS]; lane (lm, lq) of accumulator tile (i, j) holds row i*16 + lm,
    // columns j*16 + 4*lq .. + 3 (operands swapped in the MFMA: the accumulator's register index runs along the columns)
    __syncthreads();
    float* stage = reinterpret_cast<float*>(smem) + wave * 64 * Q_SS;
    const int lm = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(stage + (i * 16 + lm) * Q_SS + j * 16 + 4 * lq) = acc[i][j];
    __syncthreads();
    gemm128_copy_out<Q_SS>(p, stage, m0 + wr * 64, n0 + wc * 64, lane);
}

static bf16_t* q_zero16() {
    static bf16_t* z = nullptr;
    if (!z) {
        if (hipMalloc((void**)&z, 256) != hipSuccess) return nullptr;
        hipMemset(z, 0, 256);
        hipDeviceSynchronize();      // one-time: the first launch may be on a non-blocking stream that does not order behind the null stream
    }
    return z;
}

// NST (ring depth) from VLR_GEMM128P: 0 = off (the register-staged kernel runs), 2..4.  Default 2: 70 KiB of LDS, TWO workgroups per
// CU - measured faster than one workgroup with a 4-deep ring on every shape of tools/gemm128_bench.py (504 x 4096 x 22016 split 4 ways:
// 108 us against 134-140 us; the register-staged kernel: 161 us): the second workgroup's MFMAs run under the first one's DMA issue
static int q_stages() {
    static int nst = -1;
    if (nst < 0) {
        const char* e = getenv("VLR_GEMM128P");
        nst = e ? atoi(e) : 2;
        if (nst == 1 || nst > 4 || nst < 0) nst = 2;
    }
    return nst;
}

// grid = (tiles, split-K slices, groups) exactly as for gemm_bf16_kernel.  false: operands the DMA path does not take (the caller runs
// the register-staged kernel): 16-byte alignment of every row piece, K % 8, >= 8 columns on a K-strided operand
bool vlr_gemm128p_try_launch(int layout, const GemmParams& p, dim3 grid, hipStream_t stream) {
    const int nst = q_stages();
    if (!nst || p.fuse == 6) return false;
    const bool masked = p.mask_on != 0;
    if (masked) {     // packed masks only (the hashing forms stay on the register-staged kernel): 1 = NT, row-major bits; 3 = TN, K-tile-blocked transposed bits
        if (!p.mask_bits || ((uintptr_t)p.mask_bits & 3) || p.gMask % 4 != 0 || (p.splitk > 1 && p.kchunk % QK != 0)) return false;
        if (p.mask_on == 1) { if (layout != 0 || p.K % QK != 0 || p.mask_ld % QK != 0) return false; }
        else if (p.mask_on == 3) { if (layout != 2 || p.N % QT != 0 || p.mask_ld != p.N) return false; }
        else return false;
    }
    const bool a_ks = layout == 2, b_ks = layout != 0;
    if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return false;
    if (p.lda % 8 != 0 || p.ldb % 8 != 0 || (p.groups > 1 && ((p.gA | p.gB) % 8 != 0))) return false;
    if (a_ks ? (p.M % 8 != 0 || p.M < 8) : (p.K % 8 != 0)) return false;
    if (b_ks ? (p.N % 8 != 0 || p.N < 8) : (p.K % 8 != 0)) return false;
    if (p.splitk > 1 && p.kchunk % 8 != 0) return false;
    bf16_t* zero16 = q_zero16();
    if (!zero16) return false;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)gemm128p_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * Q_STAGE_BYTES);
        hipFuncSetAttribute((const void*)gemm128p_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * Q_STAGE_BYTES);
        hipFuncSetAttribute((const void*)gemm128p_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * Q_STAGE_BYTES);
        hipFuncSetAttribute((const void*)gemm128p_kernel<false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (Q_STAGE_BYTES + 1024));
        hipFuncSetAttribute((const void*)gemm128p_kernel<true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (Q_STAGE_BYTES + 1024));
        attr = true;
    }
    int lds = nst * Q_STAGE_BYTES > Q_EPI_BYTES ? nst * Q_STAGE_BYTES : Q_EPI_BYTES;
    if (masked) {
        const int need = nst * (Q_STAGE_BYTES + 1024);
        lds = need > Q_EPI_BYTES ? need : Q_EPI_BYTES;
        if (p.mask_on == 1) hipLaunchKernelGGL((gemm128p_kernel<false, false, 1>), grid, dim3(256), lds, stream, p, (const bf16_t*)zero16, nst);
        else hipLaunchKernelGGL((gemm128p_kernel<true, true, 2>), grid, dim3(256), lds, stream, p, (const bf16_t*)zero16, nst);
        return true;
    }
    if (layout == 0) hipLaunchKernelGGL((gemm128p_kernel<false, false>), grid, dim3(256), lds, stream, p, (const bf16_t*)zero16, nst);
    else if (layout == 1) hipLaunchKernelGGL((gemm128p_kernel<false, true>), grid, dim3(256), lds, stream, p, (const bf16_t*)zero16, nst);
    else hipLaunchKernelGGL((gemm128p_kernel<true, true>), grid, dim3(256), lds, stream, p, (const bf16_t*)zero16, nst);
    return true;
}
