// 256x256x64-tile bf16 MFMA GEMM, eight-phase schedule: the large-shape path of vlr_gemm_bf16 for all three layouts.
//
// Its predecessor, a two-phase staggered kernel (r01 history), was bound by the ISSUE of the LDS-DMA: one global_load_lds blocks its in-order
// wave for ~100 cycles, and a LOAD phase that carries 4-8 of them next to 12 fragment reads takes longer than the 16 MFMAs
// of its partner wave.  Here the same work is cut finer, after the guide's 256^2 8-phase template:
//   * K tile 64; LDS = 2 buffers x {A-lo, A-hi, B-lo, B-hi} half tiles of 128 rows x 64 k (16 KiB each) = 128 KiB;
//   * 8 waves = 2 (wr) x 4 (wc).  A wave owns the rows {wr*64..+64} of BOTH A halves and the columns {wc*32..+32} of BOTH
//     B halves, i.e. four 64x32 quadrants (A0|A1) x (B0|B1) of the 256x256 tile, 32 accumulator tiles of 16x16;
//   * per K tile four phases, each = {fragment ds_reads; ONE half tile of LDS-DMA (2 instructions per wave); barrier;
//     16 v_mfma_f32_16x16x32_bf16 (one quadrant x K=64, 8 independent accumulators: a 32x32x16 quadrant has only two and
//     its dependent chain left the matrix pipe idle - 1.31 PF with everything else ablated); barrier}.  Fragment reads per
//     phase: 12 (B0, A0) / 4 (B1) / 8 (A1) / 0;
//   * half tiles become free in the order B-lo, A-lo (last read in phase 1), B-hi (phase 2), A-hi (phase 3) and are re-staged
//     for tile t+2 in phases 2, 3, 4 and phase 1 of tile t+1; the only vector-memory wait is a COUNTED vmcnt(6) in phase 4:
//     three half tiles stay in flight across barriers, everything tile t+1 needs has landed;
//   * waves 4-7 (wr = 1) run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section while its partner
//     issues reads and DMA; s_setprio(1) around the MFMAs lets the arbiter prefer the former.
// Ordering rules (guide, "Read a staged buffer one phase AFTER the wait that retires it"): RAW - the phase-4 wait precedes that
// phase's first barrier, the reads start in the next phase; WAR - a half is re-staged two phases after its last read, or one
// phase after when an lgkmcnt before the reading phase's first barrier retired the reads (B-lo: the four B reads are issued
// first in phase 1 and retired by lgkmcnt(8)).
// K-contiguous operands: LDS image [128 rows][128 B], 16-byte chunk c of row r stored at c ^ ((r>>1)&7) (XOR on the per-lane
// SOURCE address of the DMA, same XOR on the ds_read_b128).  K-strided operands (stored [K][cols]): LDS image [64 k][256 B],
// chunk c of k-row r stored at c ^ (((r&3)<<2) | (((r>>3)&1)<<1)), fragments by two ds_read_b64_tr_b16.  M/N edges: rows/columns clamped on the
// source side, never stored.  K tail: chunks beyond K read a 16-byte zero buffer.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "gemm.h"
#include "gemm_tilemap.h"

#ifndef VLR_KLOOP_BAL
#define VLR_KLOOP_BAL 1        // balanced fragment reads per phase (6 / 6 / 6 / 6 instead of the template's 12 / 4 / 8 / 0); 0: the round-3 K loop (A/B builds)
#endif
#ifndef VLR_EPI_NT
#define VLR_EPI_NT 0           // 1: the continuous-pipeline epilogues store with the non-temporal hint (A/B build, `build_hip.py --define VLR_EPI_NT=1 --tag _nt`)
#endif
#if VLR_EPI_NT
#define EPI_GST(T_, addr_, val_) __builtin_nontemporal_store((T_)(val_), reinterpret_cast<T_*>(addr_))
#else
#define EPI_GST(T_, addr_, val_) (*reinterpret_cast<T_*>(addr_) = (val_))
#endif
#ifndef VLR_EPI_FAST
#define VLR_EPI_FAST 1         // the SwiGLU-backward epilogue as straight-line code on buffer addressing (exact vmcnt bookkeeping, loads a ring of chunks ahead of the stores); 0: the round-5 epilogue (A/B builds)
#endif
#ifndef VLR_KLOOP_NT6
#define VLR_KLOOP_NT6 1       // NT launches: the 18-fragment split of the TN loop, 6 / 6 / 6 / 6 ds_read_b128 per phase (0: 8 / 4 / 8 / 4)
#endif
#ifndef VLR_KLOOP_NN8
#define VLR_KLOOP_NN8 1        // NN launches: the whole next A0 read in phase 4 (8 / 8 / 8 / 8 LDS instructions per phase; 0: 12 / 8 / 8 / 4)
#endif
#ifndef VLR_KLOOP_TN12
#define VLR_KLOOP_TN12 1       // TN launches: 12 / 12 / 12 / 12 transposing reads per phase (0: the 16 / 8 / 16 / 8 of the NT split; A/B builds)
#endif
#define PT 256
#define PK 64
#define HALF_BYTES (128 * PK * 2)    // 16 KiB
#define BUF_BYTES (4 * HALF_BYTES)   // A-lo | A-hi | B-lo | B-hi
#define C_STRIDE 528                 // bytes per row of the bf16 C image (256 cols + 16 B pad: conflict-free 8-byte writes)
#define P_LDS_BYTES (256 * C_STRIDE)  // 132 KiB: the two K-tile buffers (128 KiB) / the epilogue's C image
#define PTAB_PIECES 384                // capacity of the piece table behind them (8 ints per piece); persist_grid() keeps launches below it
#define PTAB_BYTES (PTAB_PIECES * 32)
#define EPATCH_STRIDE 144              // bytes per row of a wave's epilogue patch: 32 fp32 columns + 16 B (conflict-free 16-byte row-per-lane writes)
#define EPATCH_BF16 1280               // image of a packed bf16 chunk: 16 rows x (64 B + 16 B)
#define EPATCH_BYTES (2 * EPATCH_BF16)                   // per wave: one 16-row x 32-column fp32 chunk (16 * EPATCH_STRIDE = 2304 B) or TWO bf16 chunk images
#define CONT_LDS_BYTES (2 * BUF_BYTES + PTAB_BYTES + 8 * EPATCH_BYTES)      // dynamic LDS of the continuous-pipeline kernels (all 160 KiB)
#define TILE_LDS_BYTES (P_LDS_BYTES + PTAB_BYTES)        // ... of the per-tile kernels

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// The lane id, computed where it is needed (two VALU instructions) instead of carried from threadIdx.x: a copy that lives across the K
// loop is one VGPR of a kernel that has none to spare (round 6: it was the first value hipcc spilled), and an asm result is opaque -
// nothing derived from it is hoisted out of the tile loop.
#define LANE_FRESH(v_) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v_))
// compile-time loop: f(std::integral_constant<int, B>{}) ... f(<E - 1>) - software-pipelined epilogues index their register windows statically
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// One LDS-DMA wave instruction (64 lanes x 16 B -> LDS [m0 + lane*16]) issued through inline asm ON PURPOSE: when hipcc sees
// the builtin it tracks an outstanding "LDS store through VMEM" and puts s_waitcnt vmcnt(0) in front of every
// ds_read_b64_tr_b16 that follows (observed in the ISA of the builtin version: the DMA queue was drained every phase).
// Hidden from the compiler, the only vector-memory waits are the counted ones written in the K loop.
__device__ __forceinline__ void lds_dma16(const bf16_t* g, char* lds) {
    const uint32_t l = (uint32_t)(uintptr_t)(lvoid_t*)lds;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}
// fast form: uniform 64-bit base in SGPRs + per-lane 32-bit byte offset (loop invariant), no per-piece vector arithmetic
__device__ __forceinline__ void lds_dma16_s(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// per-lane byte offsets of the two pieces this wave stages for one half tile, relative to the K tile's base address
__device__ __forceinline__ void piece_off_kc(int ld, int row0, int nrows, int wave, int lane, uint32_t (&off)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave + 8 * i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int grow = row0 + r;
        grow = grow < nrows ? grow : nrows - 1;
        off[i] = (uint32_t)(((size_t)grow * ld + c * 8) * 2);
    }
}
// B-operand row of local row r (0..127) of half tile h for the fused-epilogue variants (GemmParams::fuse); FUSE 0 = plain
template <int FUSE>
__device__ __forceinline__ int b_src_row(int n0, int h, int r, int N) {
    if constexpr (FUSE == 1) {            // SwiGLU: lo = gate rows [n0, n0+128), hi = up rows I + [n0, n0+128)
        const int I = N >> 1;
        int c = n0 + r;
        c = c < I ? c : I - 1;
        return h * I + c;
    } else if constexpr (FUSE == 2) {     // RoPE: lo = features 0..63 of the tile's two heads, hi = features 64..127
        return n0 + (r >> 6) * 128 + h * 64 + (r & 63);
    } else {
        const int g = n0 + h * 128 + r;
        return g < N ? g : N - 1;
    }
}
template <int FUSE>
__device__ __forceinline__ void piece_off_kc_b(int ld, int n0, int h, int N, int wave, int lane, uint32_t (&off)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave + 8 * i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        off[i] = (uint32_t)(((size_t)b_src_row<FUSE>(n0, h, r, N) * ld + c * 8) * 2);
    }
}
template <int FUSE>
__device__ __forceinline__ void stage_kc_b(const bf16_t* __restrict__ P, int ld, int n0, int h, int N, int k0, int K,
                                           const bf16_t* __restrict__ zero16, char* half, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = (wave + 8 * i) * 8;
        const int r = R + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int k = k0 + c * 8;
        const bf16_t* g = (k + 8 <= K) ? P + (size_t)b_src_row<FUSE>(n0, h, r, N) * ld + k : zero16;
        lds_dma16(g, half + R * 128);
    }
}
// adapter segment, B operand (lora_B rows [N][ld], K2 columns): half h of a SwiGLU tile (gate / up rows) owns the K range
// [h*K2, (h+1)*K2) of the tile's 2*K2-wide segment and reads zeros elsewhere; plain / RoPE tiles own [0, K2)
template <int FUSE>
__device__ __forceinline__ void stage_seg_b(const bf16_t* __restrict__ P, int ld, int n0, int h, int N, int k0, int K2,
                                            const bf16_t* __restrict__ zero16, char* half, int wave, int lane) {
    const int koff = FUSE == 1 ? h * K2 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = (wave + 8 * i) * 8;
        const int r = R + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int k = k0 + c * 8 - koff;
        const bf16_t* g = (k >= 0 && k + 8 <= K2) ? P + (size_t)b_src_row<FUSE>(n0, h, r, N) * ld + k : zero16;
        lds_dma16(g, half + R * 128);
    }
}
__device__ __forceinline__ void piece_off_ks(int ld, int col0, int ncols, int wave, int lane, uint32_t (&off)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave + 8 * i) * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1));
        int col = col0 + chunk * 8;
        col = col + 8 <= ncols ? col : ncols - 8;
        off[i] = (uint32_t)(((size_t)r * ld + col) * 2);
    }
}
// ---- LDS-DMA of one half tile, 2 wave instructions (1 KiB each) per wave (general form: K tail -> zeros)
// k-contiguous operand P[rows][ld]: instruction = 8 rows x 128 B
__device__ __forceinline__ void stage_kc(const bf16_t* __restrict__ P, int ld, int row0, int nrows, int k0, int K,
                                         const bf16_t* __restrict__ zero16, char* half, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = (wave + 8 * i) * 8;
        const int r = R + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int grow = row0 + r;
        grow = grow < nrows ? grow : nrows - 1;
        const int k = k0 + c * 8;
        const bf16_t* g = (k + 8 <= K) ? P + (size_t)grow * ld + k : zero16;
        lds_dma16(g, half + R * 128);
    }
}
// k-strided operand P[K][ld] (columns contiguous): instruction = 4 k-rows x 256 B
__device__ __forceinline__ void stage_ks(const bf16_t* __restrict__ P, int ld, int col0, int ncols, int k0, int K,
                                         const bf16_t* __restrict__ zero16, char* half, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = (wave + 8 * i) * 4;
        const int r = R + (lane >> 4);
        const int chunk = (lane & 15) ^ (((r & 3) << 2) | (((r >> 3) & 1) << 1));
        int col = col0 + chunk * 8;
        col = col + 8 <= ncols ? col : ncols - 8;
        const int k = k0 + r;
        const bf16_t* g = (k < K) ? P + (size_t)k * ld + col : zero16;
        lds_dma16(g, half + R * 256);
    }
}
// ---- MFMA operand fragments for v_mfma_f32_16x16x32_bf16 (16 rows x 32 k of slice s; lane l: row l&15, k (l>>4)*8..+8)
__device__ __forceinline__ bf16x8 pfrag_kc(const char* half, int rbase, int s, int lane) {
    const int row = rbase + (lane & 15);
    const int chunk = s * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(half + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
// k-strided half [64 k][128 cols]: within a 16-lane group g lane p supplies the address of (k-row p/4, 4 columns at (p%4)*4)
// and receives column p of that 4 x 16 block; two reads (k-rows +0..3, +4..7 of the group's 8).  A 32-lane half of the
// instruction covers k-rows {b..b+3} and {b+8..b+11} x 32 B: the swizzle spreads them over 8 distinct 32-byte segments.
__device__ __forceinline__ bf16x8 pfrag_ks(const char* half, int cbase, int s, int lane) {
    const int g = lane >> 4, pq = lane & 15;
    const int krow = s * 32 + g * 8 + (pq >> 2);
    const int col = cbase + (pq & 3) * 4;
    const int swz = ((krow & 3) << 2) | (((krow >> 3) & 1) << 1);
    const int off = krow * 256 + (((col >> 3) ^ swz) << 4) + ((col >> 2) & 1) * 8;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(half + off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(half + off + 4 * 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Widen the register-direct epilogue stores (guide T21).  A lane (lm, lq) holds columns 4 lq .. 4 lq + 3 of one row of the two
// 16-column tiles j = 0, 1 (packed bf16: two dwords each).  v_permlane16_swap_b32 exchanges the odd 16-lane rows of its first
// operand with the even rows of the second, so afterwards the lane holds EIGHT consecutive columns of ONE tile:
//   lq 0: tile 0 cols 0-7 | lq 1: tile 1 cols 0-7 | lq 2: tile 0 cols 8-15 | lq 3: tile 1 cols 8-15
// -> one 16-byte store per lane instead of two 8-byte ones (the store tail is issue-bound: half the instructions).
__device__ __forceinline__ u32x4 widen_pair(const f32x4& t0, const f32x4& t1) {
    const uint32_t a0 = pack_bf16(t0[0], t0[1]), a1 = pack_bf16(t0[2], t0[3]);
    const uint32_t b0 = pack_bf16(t1[0], t1[1]), b1 = pack_bf16(t1[2], t1[3]);
    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
    return w;
}
// column (inside the wave's 32-column strip) of the 8 values widen_pair leaves in lane group lq
__device__ __forceinline__ int widen_col(int lq) { return (lq & 1) * 16 + (lq >> 1) * 8; }

#define PFENCE() __builtin_amdgcn_sched_barrier(0)
#define PBAR()                               \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)
// waves 4-7 of the workgroup?  Recomputed from threadIdx at every use: as a value carried through the tile loop hipcc kept it in a VGPR,
// spilled that to scratch and reloaded it - behind an s_waitcnt vmcnt(0) that drained the LDS-DMA queue - at the top of every tile.
#define WAVES_HI() (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 256)
#define PWAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define PWAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// Diagnostics build only (-DVLR_GEMM_TRACE, build_hip.py --trace -> libvlr_hip_trace.so; tools/gemm_tile_trace.py): wave 0 of every
// workgroup stamps s_memrealtime (100 MHz) after the first K tile, after the K loop and after the epilogue's last store has been
// ISSUED of every piece into the free words of its piece-table entry and copies them out at the end - the timeline of the tiles of a
// persistent launch (are the epilogues of the 256 workgroups in phase?  what does an epilogue cost?  does the next tile's first
// counted wait stall behind the stores?).  SMEM returns out of order with LDS reads on lgkmcnt, hence the full wait after it.
#ifdef VLR_GEMM_TRACE
static uint32_t* g_trace = nullptr;
#define TSTAMP(slot_)                                                                  \
    do {                                                                               \
        if (p.trace && wave == 0) {                                                    \
            const uint64_t t__ = p.trace_clk ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memrealtime(); \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         \
            if (lane0 == 0) ptab[(slot_)] = (int)(uint32_t)t__;                        \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         \
        }                                                                              \
    } while (0)
// VLR_GEMM_DEPHASE="P,D" (trace build only): workgroup b idles ((b * 7) % P) * D / P microseconds before its first tile - an experiment
// on whether the epilogue bursts of workgroups that run in phase are what an epilogue costs
static int g_dephase_p = -1, g_dephase_ticks = 0;
static void trace_set(GemmParams& p) {
    if (g_dephase_p < 0) {
        g_dephase_p = 0;
        const char* e = getenv("VLR_GEMM_DEPHASE");
        int a = 0, b = 0;
        if (e && sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) { g_dephase_p = a; g_dephase_ticks = b * 100; }
    }
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("VLR_EPI_ABLATE"); abl = e ? atoi(e) : 0; }      // timing only (wrong results): 1 no epilogue loads, 2 no epilogue stores
    static int clk = -1;
    if (clk < 0) { const char* e = getenv("VLR_GEMM_TRACE_CLK"); clk = (e && e[0] == '1') ? 1 : 0; }      // 1: stamps of the SHADER clock (s_memtime) - cycles instead of 10 ns ticks
    p.trace = g_trace; p.dephase_p = g_dephase_p; p.dephase_ticks = g_dephase_ticks; p.epi_abl = abl; p.trace_clk = clk;
}
#define TRACE_SET(p_) trace_set(p_)
#define EPI_LD_ON (!(p.epi_abl & 1))
#define EPI_ST_ON (!(p.epi_abl & 2))
#else
#define TSTAMP(slot_) do { } while (0)
#define TRACE_SET(p_) do { } while (0)
#define EPI_LD_ON true
#define EPI_ST_ON true
#endif

// ---- epilogue staging of the continuous-pipeline kernels: a wave-private LDS patch (EPATCH_BYTES behind the piece table) turns the
// accumulator layout (lane (lm, lq): row lm, 4 columns at 4 lq of a 16 x 16 tile) into row-contiguous registers.  LDS instructions of
// one wave execute in order, so write -> read -> next write need no wait, only a compiler barrier (`epatch`, `lane` of the kernel).
#define EPI_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
// An epilogue load whose value is only used under a predicate (an edge-tile store) is waited for inside that branch; on the other path
// it stays "pending" for hipcc, which then guards the next write of that register - a fragment read at the TOP OF THE K LOOP - with
// s_waitcnt vmcnt(0..1) and drains the LDS-DMA queue every K tile (the TN kernel lost 7 % that way; build_hip.py now checks the ISA of
// the K loops for stray vmcnt waits).  EPI_USE consumes the value unconditionally, i.e. puts the wait where the value arrives.
#define EPI_USE(x_) asm volatile("" ::"v"(x_))
// fp32: the tiles j = 0, 1 (16 rows x 32 columns) -> o_[k] = row (lane >> 3) + 8 k, columns (lane & 7) * 4 .. + 3
#define EPI_XPOSE_F32(t0_, t1_, o_)                                                                                   \
    do {                                                                                                              \
        EPI_SYNC();                                                                                                   \
        *reinterpret_cast<f32x4*>(epatch + (lane & 15) * EPATCH_STRIDE + (lane >> 4) * 16) = (t0_);                   \
        *reinterpret_cast<f32x4*>(epatch + (lane & 15) * EPATCH_STRIDE + 64 + (lane >> 4) * 16) = (t1_);              \
        EPI_SYNC();                                                                                                   \
        (o_)[0] = *reinterpret_cast<const f32x4*>(epatch + (lane >> 3) * EPATCH_STRIDE + (lane & 7) * 16);            \
        (o_)[1] = *reinterpret_cast<const f32x4*>(epatch + ((lane >> 3) + 8) * EPATCH_STRIDE + (lane & 7) * 16);      \
    } while (0)
// fp32, 8 columns per lane: o_[0], o_[1] = row lane >> 2, columns (lane & 3) * 8 .. + 3 / + 4 .. + 7 (for epilogues whose global accesses are
// bf16: 8 columns = 16 bytes per lane, 4 lanes = 64 B per row, 16 rows per instruction - half the instructions of the 4-column form)
#define EPI_XPOSE_F32_8(t0_, t1_, o_)                                                                                 \
    do {                                                                                                              \
        EPI_SYNC();                                                                                                   \
        *reinterpret_cast<f32x4*>(epatch + (lane & 15) * EPATCH_STRIDE + (lane >> 4) * 16) = (t0_);                   \
        *reinterpret_cast<f32x4*>(epatch + (lane & 15) * EPATCH_STRIDE + 64 + (lane >> 4) * 16) = (t1_);              \
        EPI_SYNC();                                                                                                   \
        (o_)[0] = *reinterpret_cast<const f32x4*>(epatch + (lane >> 2) * EPATCH_STRIDE + (lane & 3) * 32);            \
        (o_)[1] = *reinterpret_cast<const f32x4*>(epatch + (lane >> 2) * EPATCH_STRIDE + (lane & 3) * 32 + 16);       \
    } while (0)
// bf16: w_in_ = widen_pair(tile 0, tile 1) (row lane & 15, 8 columns at widen_col(lane >> 4)) -> w_out_ = row lane >> 2, columns
// (lane & 3) * 8 .. + 7 of the 32-column strip
#define EPI_XPOSE_BF16(w_in_, w_out_)                                                                                 \
    do {                                                                                                              \
        const u32x4 w__ = (w_in_);                                                                                    \
        EPI_SYNC();                                                                                                   \
        *reinterpret_cast<u32x4*>(epatch + (lane & 15) * 80 + widen_col(lane >> 4) * 2) = w__;                        \
        EPI_SYNC();                                                                                                   \
        (w_out_) = *reinterpret_cast<const u32x4*>(epatch + (lane >> 2) * 80 + (lane & 3) * 16);                      \
    } while (0)

// pipelined form for runs of bf16 chunks: EPI_BF16_PUT writes chunk image `buf_` (0 | 1: the patch holds two) and requests it back
// row-contiguous into w_out_ WITHOUT waiting; the caller stores chunk n - 1 after putting chunk n, so that the LDS round trip of a
// chunk hides under the packing of the next one (hipcc waits with a counted lgkmcnt at the store).
#define EPI_BF16_PUT(buf_, w_in_, w_out_)                                                                                                \
    do {                                                                                                                                 \
        const u32x4 w__ = (w_in_);                                                                                                       \
        EPI_SYNC();                                                                                                                      \
        *reinterpret_cast<u32x4*>(epatch + (buf_) * EPATCH_BF16 + (lane & 15) * 80 + widen_col(lane >> 4) * 2) = w__;                    \
        EPI_SYNC();                                                                                                                      \
        (w_out_) = *reinterpret_cast<const u32x4*>(epatch + (buf_) * EPATCH_BF16 + (lane >> 2) * 80 + (lane & 3) * 16);                  \
    } while (0)

// CONT = true: continuous pipeline across the output tiles of a persistent workgroup (plain bf16 epilogue only): the last K tiles
// of tile i stage the first K tiles of tile i+1 (same slots, same counted wait), the epilogue stores straight from the
// accumulator registers (8 bytes per lane, no LDS, no barrier) and the K loop of tile i+1 starts with its data resident.
struct TileProb { const bf16_t* A; const bf16_t* B; void* C; int M, N, lda, ldb, ldc; };
// GRP = true (TN continuous pipeline only): the launch covers TWO problems of equal K - (A, B, C, M, N, ld*) and (A1, ... ) of GemmParams -
// as one list of output tiles, the first problem's tiles first: the weight gradients of gate|up and down_proj are 5.375 + 2.69 rounds
// of 256 tiles as two launches (6 + 3) and 8.06 as one (the host peels one tile row to the 128x128 kernel: 8).
template <bool A_KS, bool B_KS, int ABL = 0, bool CONT = false, int FUSE = 0, bool SEG = false, bool GRP = false>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmParams p, const bf16_t* __restrict__ zero16) {
    static_assert(!GRP || (A_KS && B_KS && CONT && FUSE == 0 && !SEG), "grouped launches: TN, plain epilogue");
    static_assert(FUSE == 0 || (FUSE == 6 && !CONT && !A_KS && B_KS) || (FUSE != 6 && CONT && !A_KS && (FUSE == 3 ? B_KS : !B_KS)),
                  "fused epilogues: continuous pipeline; 1, 2, 4, 5 NT, 3 NN; 6 (dropout-accumulate) NN on the LDS-image epilogue");
    static_assert(!SEG || (!A_KS && !B_KS && FUSE <= 2), "adapter segment: NT, plain / SwiGLU / RoPE epilogues");
    constexpr int NW = FUSE == 1 ? 128 : PT;      // output columns (of the gate half, for SwiGLU) per workgroup tile
    extern __shared__ __attribute__((aligned(16))) char smem[];   // P_LDS_BYTES
    const int t = threadIdx.x;
    const int lane0 = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // consume the kernel arguments here: otherwise the s_load of `zero16` is first waited for (lgkmcnt(0)) inside the K loop,
    // in front of the phase-1 DMA, and drains the 12 fragment reads every iteration
    asm volatile("" ::"s"(zero16), "s"(p.A), "s"(p.B), "s"(p.K), "s"(p.M), "s"(p.N), "s"(p.lda), "s"(p.ldb));

    // ---- persistent workgroups: the grid is min(#tiles, #CUs) and every workgroup walks the tiles of ITS XCD (block b sits on
    // XCD b % 8) in steps of gridDim/8.  A finished tile's global stores drain while the next tile's first K tiles are
    // fetched; a workgroup that ends instead holds its CU (LDS) until the stores have landed and the next one starts cold:
    // 10-12 us per tile, measured (K sweep: 0.23 ms of fixed cost per 2304-tile launch).
    const int tiles_m = (p.M + PT - 1) / PT, tiles_n = FUSE == 1 ? ((p.N >> 1) + NW - 1) / NW : (p.N + PT - 1) / PT;
    const int tiles_m1 = GRP ? (p.M1 + PT - 1) / PT : 0, tiles_n1 = GRP ? (p.N1 + PT - 1) / PT : 0;
    const int nwg0 = tiles_m * tiles_n;
    const int nwg = nwg0 + tiles_m1 * tiles_n1;
    // SEG: K tiles [0, nt1) come from (A, B), [nt1, nt) from the adapter pair (A2 columns of this tile's output block, B2)
    const int nt1 = (p.K + PK - 1) / PK;
    const int k2t = !SEG ? 0 : (FUSE == 1 ? 2 * p.K2 : p.K2);
    const int nt = nt1 + (k2t + PK - 1) / PK;
    // ---- the workgroup's list of PIECES (a piece = one output tile, whole K): piece i = tile (j + i * G8) of this XCD's tiles
    // (j = blockIdx / 8, G8 = gridDim / 8), written ONCE into LDS behind the K-tile buffers (lane l of wave 0 computes pieces l, l + 64,
    // ...): an iteration reads its own and the next piece with two broadcast ds_reads - the scalars of the enumeration, kept live
    // across the K loop, pushed the kernel into scratch.  (Round 3 also cut pieces along K across workgroups - a stream-K tail, an
    // XCD rotation, an XCD round barrier, accumulators handed over through fp32 slabs: all measured slower or neutral, DESIGN.md
    // section 4, and removed in round 4.)
    const int G8 = (int)gridDim.x >> 3, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const bool persistent = (int)gridDim.x < nwg;                  // else: one tile per workgroup, gridDim == nwg (any count)
    const int n_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);       // tiles of this XCD
    // sched bit 5 (VLR_SCHED_SHARED, gemm_tilemap.h): round i of the launch is tiles [i * gridDim, (i + 1) * gridDim) of a list in which the
    // XCD blocks of a round are stacked into one super-block (shared B panels, A panels re-read one round later: Infinity-Cache distances)
    // - XCD x takes the x-th run of gridDim / 8 tiles of the round; persistent launches with whole XCD octets only
    const bool shared_map = persistent && (p.sched & 32) && ((int)gridDim.x & 7) == 0;
    const int l_first = xcd * G8 + jx;
    const int npieces = __builtin_amdgcn_readfirstlane(!persistent ? 1 : shared_map ? (nwg - l_first + (int)gridDim.x - 1) / (int)gridDim.x
                                                                                    : (n_x - jx + G8 - 1) / G8);
    int* ptab = reinterpret_cast<int*>(smem + (CONT ? 2 * BUF_BYTES : P_LDS_BYTES));
    if (t < 64) {
        for (int i = t; i < npieces; i += 64) {
            int pid;
            if (shared_map) {
                pid = i * (int)gridDim.x + l_first;
            } else {
                const int idx = jx + i * G8;
                const int q = nwg >> 3, rem = nwg & 7;
                pid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;      // XCD x owns a contiguous range of tile ids (bijective)
            }
            const bool g1 = GRP && pid >= nwg0;
            const int pl = g1 ? pid - nwg0 : pid, tm = g1 ? tiles_m1 : tiles_m, tn_ = g1 ? tiles_n1 : tiles_n;
            int trow, tcol;
            if (shared_map) {
                vlr_tile_of_shared(pl, tm, tn_, &trow, &tcol, (SEG && p.seg_skip) ? 1 : 0);      // rows rotate only where tiles differ in length
            } else {
                const int GROUP = 8;
                const int per_group = GROUP * tn_;
                const int first_m = (pl / per_group) * GROUP;
                const int gsz = min(tm - first_m, GROUP);
                trow = first_m + (pl % per_group) % gsz;
                tcol = (pl % per_group) / gsz;
            }
            ptab[i * 8 + 0] = trow * PT;
            ptab[i * 8 + 1] = tcol * NW;
            ptab[i * 8 + 2] = g1 ? 1 : 0;
            if constexpr (SEG) ptab[i * 8 + 3] = (p.seg_skip && p.seg_skip[trow]) ? 1 : 0;      // this row tile holds no row of the segment's second adapter
        }
    }
    __syncthreads();
    if (npieces <= 0) return;
#ifdef VLR_GEMM_TRACE
    if (CONT && p.dephase_p > 0) {
        const uint64_t until = __builtin_amdgcn_s_memrealtime() + (uint64_t)(((int)blockIdx.x * 7) % p.dephase_p) * p.dephase_ticks / p.dephase_p;
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(16);
    }
#endif
    TSTAMP((PTAB_PIECES - 1) * 8 + 5);
    int parb = 0;                 // CONT: buffer parity of the current piece's K tile 0 (K tiles keep alternating across pieces)
    for (int titer = 0;; ++titer) {
    // per-tile opaque copy of the lane id: keeps hipcc from hoisting every lane-derived address out of the tile loop (it did,
    // and spilled 100-200 bytes per lane into the K loop)
    int lane;
    LANE_FRESH(lane);
#define RFL(x) __builtin_amdgcn_readfirstlane(x)
    const int m0 = RFL(ptab[titer * 8 + 0]), n0 = RFL(ptab[titer * 8 + 1]);
    // a piece is a whole tile: K tiles [0, nt) - or, SEG with the row tile's skip flag set, the base K tiles and the first seg_keep / 64 K tiles
    // of every sub-target's block of the segment (K tile s of the short list = K tile (s / kl) * (K2 / 64) + s % kl of the segment)
    const bool seg_short = SEG && RFL(ptab[titer * 8 + 3]) != 0;
    const int seg_kl = SEG ? p.seg_keep / PK : 0, seg_kf = SEG ? p.K2 / PK : 0;
    const int kb = 0, ntp = seg_short ? nt1 + (FUSE == 1 ? 2 : 1) * seg_kl : nt;
    const bool has_next = CONT && titer + 1 < npieces;
    const int tni = has_next ? titer + 1 : titer;
    const int m0n = RFL(ptab[tni * 8 + 0]), n0n = RFL(ptab[tni * 8 + 1]);
    const int kbn = 0;
    // operands of this piece and of the next one: the launch's (A, B, C), or - GRP, two problems of equal K in one launch - those of
    // the problem the piece belongs to (piece-table word 2)
    TileProb tp = {p.A, p.B, p.C, p.M, p.N, p.lda, p.ldb, p.ldc}, tnx = tp;
    if constexpr (GRP) {
        const TileProb t1 = {p.A1, p.B1, p.C1, p.M1, p.N1, p.lda1, p.ldb1, p.ldc1};
        if (RFL(ptab[titer * 8 + 2])) tp = t1;
        if (RFL(ptab[tni * 8 + 2])) tnx = t1;
    }
    const bool first = !CONT || titer == 0;
    const bool epi_par = CONT && !(p.sched & 16);      // both wave groups in the epilogue at once (sched bit 4 = the old serial order, A/B)
    parb = RFL(parb);

    f32x4 acc[2][4][2][2];   // [A half a][16-row tile i][B half b][16-col tile j]
    float zero_;             // an opaque zero per tile: the constant was kept as a 4-register tuple across the tile loop (and spilled, round 6)
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][i][b][j][r] = zero_;

    const int a2off = (!SEG || FUSE == 1) ? 0 : (n0 >= p.seg_b0 ? (n0 >= p.seg_b1 ? 2 : 1) : 0) * p.K2;
    // half h of K tile `tile`: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi
    // diagnostics (template ABL via VLR_GEMM_ABLATE, NT only, timing only - results are wrong): 1 no DMA in the K loop,
    // 2 no fragment reads after the first K tile, 4 no barriers in the K loop
    constexpr bool abl_dma = ABL & 1, abl_rd = ABL & 2, abl_bar = ABL & 4;
    bool in_loop = false;
    // loop-invariant per-lane source offsets (bytes) of this wave's two pieces of each half, and the per-K-tile base step
    uint32_t offA[2][2], offB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if constexpr (A_KS) piece_off_ks(tp.lda, m0 + h * 128, tp.M, wave, lane, offA[h]);
        else piece_off_kc(tp.lda, m0 + h * 128, tp.M, wave, lane, offA[h]);
        if constexpr (B_KS) piece_off_ks(tp.ldb, n0 + h * 128, tp.N, wave, lane, offB[h]);
        else piece_off_kc_b<FUSE>(tp.ldb, n0, h, tp.N, wave, lane, offB[h]);
    }
    const size_t stepA = A_KS ? (size_t)PK * tp.lda * 2 : (size_t)PK * 2;
    const size_t stepB = B_KS ? (size_t)PK * tp.ldb * 2 : (size_t)PK * 2;
    const uint32_t lds_wave = (uint32_t)(uintptr_t)(lvoid_t*)smem + wave * 1024;
    // fastc = true: the caller guarantees tile < nt and that the tile is a full one (steady-state loop): no checks at all
    // `tile` is relative to the piece: K tile kb + tile of this output tile; tile >= ntp: K tile kbn + (tile - ntp) of the NEXT piece
    auto stage = [&](int tile, auto hc, auto fastc) {
        constexpr int h = decltype(hc)::value;
        constexpr bool fast = decltype(fastc)::value;
        if constexpr (abl_dma) { if (in_loop) return; }
        if constexpr (!fast) {
            if (tile >= ntp) {
                if constexpr (CONT) {      // the K loop runs on into the next piece: its first K tiles, general path
                    if (!has_next) return;
                    int lane;                            // opaque per call (shadows the tile's copy): hipcc hoisted the general path's 64-bit
                    LANE_FRESH(lane);                    // source addresses out of the slow K loop and spilled them around it
                    char* dst = smem + ((tile + parb) & 1) * BUF_BYTES + h * HALF_BYTES;
                    const int k0 = (kbn + tile - ntp) * PK;
                    if constexpr (h < 2) {
                        if constexpr (A_KS) stage_ks(tnx.A, tnx.lda, m0n + h * 128, tnx.M, k0, p.K, zero16, dst, wave, lane);
                        else stage_kc(tnx.A, tnx.lda, m0n + h * 128, tnx.M, k0, p.K, zero16, dst, wave, lane);
                    } else {
                        if constexpr (B_KS) stage_ks(tnx.B, tnx.ldb, n0n + (h - 2) * 128, tnx.N, k0, p.K, zero16, dst, wave, lane);
                        else stage_kc_b<FUSE>(tnx.B, tnx.ldb, n0n, h - 2, tnx.N, k0, p.K, zero16, dst, wave, lane);
                    }
                }
                return;
            }
        }
        if constexpr (SEG && !fast) {
            if (kb + tile >= nt1) {
                int ln = lane;                       // opaque per call: keeps the adapter-segment addresses out of the loop-invariant
                asm volatile("" : "+v"(ln));         // set hipcc would otherwise carry (and spill) through the whole K loop
                char* dst = smem + ((tile + parb) & 1) * BUF_BYTES + h * HALF_BYTES;
                int sx = kb + tile - nt1;
                if (seg_short) sx = (sx / seg_kl) * seg_kf + sx % seg_kl;
                const int k0 = sx * PK;
                if constexpr (h < 2) stage_kc(p.A2 + a2off, p.lda2, m0 + h * 128, tp.M, k0, k2t, zero16, dst, wave, ln);
                else stage_seg_b<FUSE>(p.B2, p.ldb2, n0, h - 2, tp.N, k0, p.K2, zero16, dst, wave, ln);
                return;
            }
        }
        if constexpr (SEG && fast) {
            // whole adapter tiles (K2 % 64 == 0) on the DMA path: uniform base of the segment's K tile + per-lane offsets recomputed here
            // (a handful of VALU ops per piece; kept out of the loop-carried registers like the general form above).  SwiGLU: the half
            // that does not own this K range streams the 16 zero bytes (every lane the same address).
            if (kb + tile >= nt1) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const uint32_t dst = lds_wave + ((tile + parb) & 1) * BUF_BYTES + h * HALF_BYTES;
                int s_ = kb + tile - nt1;
                if (seg_short) s_ = (s_ / seg_kl) * seg_kf + s_ % seg_kl;
                uint32_t o2[2];
                if constexpr (h < 2) {
                    piece_off_kc(p.lda2, m0 + h * 128, tp.M, wave, ln, o2);
                    const char* base = (const char*)(p.A2 + a2off) + (size_t)s_ * (PK * 2);
                    lds_dma16_s(base, o2[0], dst);
                    lds_dma16_s(base, o2[1], dst + 8192);
                } else {
                    const int k0 = s_ * PK - (FUSE == 1 ? (h - 2) * p.K2 : 0);
                    if (k0 >= 0 && k0 < p.K2) {
                        piece_off_kc_b<FUSE>(p.ldb2, n0, h - 2, tp.N, wave, ln, o2);
                        const char* base = (const char*)p.B2 + (ptrdiff_t)k0 * 2;
                        lds_dma16_s(base, o2[0], dst);
                        lds_dma16_s(base, o2[1], dst + 8192);
                    } else {
                        uint32_t z = 0;
                        asm volatile("" : "+v"(z));
                        lds_dma16_s((const char*)zero16, z, dst);
                        lds_dma16_s((const char*)zero16, z, dst + 8192);
                    }
                }
                return;
            }
        }
        if (!fast && (p.K & (PK - 1)) != 0 && kb + tile == nt - 1) {      // only the last K tile can be partial: it takes the general (zero-filling) path
            int lane;
            LANE_FRESH(lane);
            char* dst = smem + ((tile + parb) & 1) * BUF_BYTES + h * HALF_BYTES;
            const int k0 = (kb + tile) * PK;
            if constexpr (h < 2) {
                if constexpr (A_KS) stage_ks(tp.A, tp.lda, m0 + h * 128, tp.M, k0, p.K, zero16, dst, wave, lane);
                else stage_kc(tp.A, tp.lda, m0 + h * 128, tp.M, k0, p.K, zero16, dst, wave, lane);
            } else {
                if constexpr (B_KS) stage_ks(tp.B, tp.ldb, n0 + (h - 2) * 128, tp.N, k0, p.K, zero16, dst, wave, lane);
                else stage_kc_b<FUSE>(tp.B, tp.ldb, n0, h - 2, tp.N, k0, p.K, zero16, dst, wave, lane);
            }
            return;
        }
        const uint32_t dst = lds_wave + ((tile + parb) & 1) * BUF_BYTES + h * HALF_BYTES;
        if constexpr (h < 2) {
            const char* base = (const char*)tp.A + (size_t)(kb + tile) * stepA;
            lds_dma16_s(base, offA[h][0], dst);
            lds_dma16_s(base, offA[h][1], dst + 8192);
        } else {
            const char* base = (const char*)tp.B + (size_t)(kb + tile) * stepB;
            lds_dma16_s(base, offB[h - 2][0], dst);
            lds_dma16_s(base, offB[h - 2][1], dst + 8192);
        }
    };
    using H_ALO = std::integral_constant<int, 0>;
    using H_AHI = std::integral_constant<int, 1>;
    using H_BLO = std::integral_constant<int, 2>;
    using H_BHI = std::integral_constant<int, 3>;

    // ---- prologue: K tile 0 resident, the first three halves of tile 1 in flight
    using SLOW = std::false_type;
    using FAST = std::true_type;
    if (first) {
        // (continuous pipeline: K >= 256 is a launch condition, so K tiles 0 and 1 are whole tiles of (A, B) - the unchecked path)
        using PRO = std::integral_constant<bool, CONT>;
        stage(0, H_BLO{}, PRO{}); stage(0, H_ALO{}, PRO{}); stage(0, H_BHI{}, PRO{}); stage(0, H_AHI{}, PRO{});
        stage(1, H_BLO{}, PRO{}); stage(1, H_ALO{}, PRO{}); stage(1, H_BHI{}, PRO{});
        // (from the second tile on the previous tile's stores are still in flight and vmcnt counts them too: wait for everything)
        if (titer == 0 && ntp >= 2) PWAIT_VM(6); else PWAIT_VM(0);
        PBAR();
        if (WAVES_HI()) PBAR();      // waves 4-7 run one barrier behind waves 0-3
    } else if (epi_par) {
        if (WAVES_HI()) PBAR();      // ... again after an epilogue that both wave groups ran side by side (below)
    }

    bf16x8 fa0[4][2], fa1[4][2], fb0[2][2], fb1[2][2];   // [16-row/col tile][k slice of 32]
    // Per-lane LDS read pointers into the CURRENT buffer's A-lo / B-lo half, loop carried and flipped by +-BUF_BYTES per K tile,
    // so that every fragment read is `pointer + compile-time offset` (half, tile, slice): no address arithmetic in the phases.
    //   k-contiguous: one pointer per k slice (the swizzled chunk differs by an XOR, the 16-row tiles by +2048 B);
    //   k-strided:    one pointer per 16-column tile (XOR in the chunk index), the k slices / row quads by +8192 / +1024 B.
    constexpr int NPA = A_KS ? 4 : 2, NPB = 2;
    const char* pa[NPA];
    const char* pb[NPB];
    {
        const int g = lane >> 4, pq = lane & 15;
        if constexpr (A_KS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int krow = g * 8 + (pq >> 2), col = wr * 64 + i * 16 + (pq & 3) * 4;
                const int swz = ((krow & 3) << 2) | (((krow >> 3) & 1) << 1);
                pa[i] = smem + krow * 256 + (((col >> 3) ^ swz) << 4) + ((col >> 2) & 1) * 8;
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = wr * 64 + pq;
                pa[ks] = smem + row * 128 + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 4);
            }
        }
        if constexpr (B_KS) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int krow = g * 8 + (pq >> 2), col = wc * 32 + j * 16 + (pq & 3) * 4;
                const int swz = ((krow & 3) << 2) | (((krow >> 3) & 1) << 1);
                pb[j] = smem + 2 * HALF_BYTES + krow * 256 + (((col >> 3) ^ swz) << 4) + ((col >> 2) & 1) * 8;
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = wc * 32 + pq;
                pb[ks] = smem + 2 * HALF_BYTES + row * 128 + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 4);
            }
        }
    }
    int flip = BUF_BYTES;
    if (CONT && parb) {           // this tile's K tile 0 sits in buffer 1
#pragma unroll
        for (int q = 0; q < NPA; ++q) pa[q] += BUF_BYTES;
#pragma unroll
        for (int q = 0; q < NPB; ++q) pb[q] += BUF_BYTES;
        flip = -BUF_BYTES;
    }
    auto tr2 = [&](const char* q) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)q);
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(q + 4 * 256));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // hi = 0: the -lo half, 1: the -hi half of the current buffer
    auto rdA = [&](auto hic, bf16x8 (&f)[4][2]) {
        constexpr int ho = decltype(hic)::value * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (A_KS) f[i][ks] = tr2(pa[i] + ho + ks * 8192);
                else f[i][ks] = *reinterpret_cast<const bf16x8*>(pa[ks] + ho + i * 2048);
            }
    };
    auto rdB = [&](auto hic, bf16x8 (&f)[2][2]) {
        constexpr int ho = decltype(hic)::value * HALF_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (B_KS) f[j][ks] = tr2(pb[j] + ho + ks * 8192);
                else f[j][ks] = *reinterpret_cast<const bf16x8*>(pb[ks] + ho + j * 2048);
            }
    };
    using LO = std::integral_constant<int, 0>;
    using HI = std::integral_constant<int, 1>;
#if VLR_KLOOP_BAL
    // 16-row tiles [I0, I1) of an A half; boff = 0: the current K-tile buffer, `flip`: the other one (the NEXT K tile's)
    auto rdAr = [&](auto hic, auto i0c, auto i1c, bf16x8 (&f)[4][2], int boff) {
        constexpr int ho = decltype(hic)::value * HALF_BYTES, I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
#pragma unroll
        for (int i = I0; i < I1; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (A_KS) f[i][ks] = tr2(pa[i] + boff + ho + ks * 8192);
                else f[i][ks] = *reinterpret_cast<const bf16x8*>(pa[ks] + boff + ho + i * 2048);
            }
    };
    using I0_ = std::integral_constant<int, 0>;
    using I1_ = std::integral_constant<int, 1>;
    using I2_ = std::integral_constant<int, 2>;
    using I3_ = std::integral_constant<int, 3>;
    using I4_ = std::integral_constant<int, 4>;
#endif
#define PMFMA(A_, B_, a_, b_)                                                                                            \
    do {                                                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
            acc[a_][i][b_][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B_[j][ks], A_[i][ks], acc[a_][i][b_][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
    } while (0)

#define LBAR() do { if (!abl_bar) PBAR(); } while (0)
    // steady state: absolute K tiles kb + kt + 2 below `lim` are whole tiles of (A, B); SEG: K % 64 == 0, the adapter tiles too when K2 % 64 == 0
    const int lim = ((SEG && ((p.K2 % PK) != 0 || (p.sched & 8))) ? nt1 : nt) - ((p.K & (PK - 1)) != 0 ? 1 : 0) - kb;
    const int n_fast = (ntp < lim ? ntp : lim) - 2;
#if VLR_KLOOP_BAL
    // (TN - both operands K-strided, two transposing reads per fragment - takes the 12 / 12 / 12 / 12 split of ktile_bal below: A0[0..2] ahead)
    if (n_fast > 0) {
        if constexpr ((A_KS && VLR_KLOOP_TN12) || (!A_KS && !B_KS && VLR_KLOOP_NT6)) rdAr(LO{}, I0_{}, I3_{}, fa0, 0);
        else if constexpr (!A_KS && B_KS && VLR_KLOOP_NN8) rdAr(LO{}, I0_{}, I4_{}, fa0, 0);
        else rdAr(LO{}, I0_{}, I2_{}, fa0, 0);
    }      // first part of A-lo of K tile 0 (inside the steady state phase 4 reads the next K tile's)
#endif
    in_loop = true;
#if VLR_KLOOP_BAL
    // Balanced phases (round 4).  The template's 12 / 4 / 8 / 0 fragment reads per phase make phase 1 the long pole: its load section
    // (12 reads + one half tile of LDS-DMA) outlasts the 16 MFMAs of the partner wave group, and every phase lasts max(load section,
    // MFMA section) because the two groups swap roles at the barriers (SQ_VALU_MFMA_BUSY_CYCLES: 84 % of the CU-busy cycles in the
    // longest-K launches).  Here the first two 16-row tiles of the NEXT K tile's A-lo are read in phase 4 (which read nothing), into the
    // registers phase 2 released: 8 / 4 / 8 / 4.  No more than 16 fragments are ever live (as before) - reading further ahead (6 / 6 / 6 / 6)
    // needs 32 more registers and spilled inside the K loop.
    //   phase 1  B0 (4, retired first), A0[2..3] (4) | stage A-hi(kt+1)                 | MFMA A0 x B0
    //   phase 2  B1 (4)                              | stage B-lo(kt+2)                 | MFMA A0 x B1
    //   phase 3  A1 (8)                              | stage A-lo(kt+2); vmcnt(8)        | MFMA A1 x B1
    //   phase 4  A0(kt+1)[0..1] (4)                  | stage B-hi(kt+2); vmcnt(6)        | MFMA A1 x B0
    // RAW: A-lo of tile kt+1 was issued in phase 3 of tile kt-1; the phase-3 wait (newest 8 = B-hi(kt+1), A-hi(kt+1), B-lo(kt+2),
    // A-lo(kt+2) may stay in flight) precedes phase 3's first barrier, the reads start in phase 4.  WAR: A-lo(kt) is re-staged in
    // phase 3 of tile kt: its last reads are now in phase 1 of tile kt (retired before that phase's MFMAs, two barriers earlier).
    // Steady-state iterations only (every staged tile exists and is whole); the last two K tiles of a tile and ragged cases take the
    // classic 12 / 4 / 8 / 0 body below, which reads all of A0 itself - so nothing read ahead is live when the slow path (or the
    // epilogue) needs registers.
    auto ktile_bal = [&](int kt, auto prec) {      // prec: read the next K tile's A0[0..1] in phase 4 (all but the last steady-state iteration)
        using fastc_t = FAST;
        const fastc_t fastc{};
        const bool rd = !abl_rd || kt == 0;
        constexpr bool more = true;
        constexpr bool bal18 = (A_KS && VLR_KLOOP_TN12) || (!A_KS && !B_KS && VLR_KLOOP_NT6);
        if constexpr (bal18) {
            // TN (round 5): every fragment is TWO ds_read_b64_tr_b16, so the split above is 16 / 8 / 16 / 8 LDS instructions per phase and the
            // load sections of phases 1 and 3 (16 reads + one half tile of LDS-DMA) outlast the partner group's 16 MFMAs - 2570 cycles per
            // K tile against 2200 for NT (tools/gemm_tile_trace.py with VLR_GEMM_TRACE_CLK=1).  Here 12 / 12 / 12 / 12: A1[0] moves into phase
            // 2 (its registers are free since phase 4 of the previous K tile) and the next K tile's A0[2] into phase 4; 18 fragments live
            // instead of 16 (+8 registers of the 15 this instantiation has left).
            //   phase 1  B0 (4 fragments, retired first), A0[3] (2)      | stage A-hi(kt+1)            | MFMA A0 x B0
            //   phase 2  B1 (4), A1[0] (2)                               | stage B-lo(kt+2)            | MFMA A0 x B1
            //   phase 3  A1[1..3] (6)                                    | stage A-lo(kt+2); vmcnt(8)  | MFMA A1 x B1
            //   phase 4  A0(kt+1)[0..2] (6)                              | stage B-hi(kt+2); vmcnt(6)  | MFMA A1 x B0
            // RAW / WAR as above: A-hi(kt) was waited for in phase 4 of kt-1 and is re-staged in phase 1 of kt+1; A-lo(kt+1) is waited for in
            // phase 3 (before its first barrier) and read in phase 4.
            if (rd) rdB(LO{}, fb0);
            PFENCE();
            if (rd) rdAr(LO{}, I3_{}, I4_{}, fa0, 0);
            PFENCE();
            stage(kt + 1, H_AHI{}, fastc);
            PFENCE();
            if constexpr (A_KS) PWAIT_LGKM(4); else PWAIT_LGKM(2);      // the B reads are retired (B-lo is re-staged in phase 2); the A reads (TN: 4, NT: 2 instructions) may be in flight
            LBAR();
            PMFMA(fa0, fb0, 0, 0);
            LBAR();
            if (rd) rdB(HI{}, fb1);
            PFENCE();
            if (rd) rdAr(HI{}, I0_{}, I1_{}, fa1, 0);
            PFENCE();
            stage(kt + 2, H_BLO{}, fastc);
            LBAR();
            PMFMA(fa0, fb1, 0, 1);
            LBAR();
            if (rd) rdAr(HI{}, I1_{}, I4_{}, fa1, 0);
            PFENCE();
            stage(kt + 2, H_ALO{}, fastc);
            PFENCE();
            PWAIT_VM(8);
            LBAR();
            PMFMA(fa1, fb1, 1, 1);
            LBAR();
            if constexpr (decltype(prec)::value) { if (rd) rdAr(LO{}, I0_{}, I3_{}, fa0, flip); }
            PFENCE();
            stage(kt + 2, H_BHI{}, fastc);
            PFENCE();
            PWAIT_VM(6);
            LBAR();
            PMFMA(fa1, fb0, 1, 0);
            LBAR();
        } else {
        // ---------------- phase 1
        // (NN, VLR_KLOOP_NN8: B is K-strided - 8 transposing reads per B half - so the whole next A0 is read in phase 4 and phase 1 keeps only
        // B0: 8 / 8 / 8 / 8 LDS instructions instead of 12 / 8 / 8 / 4; 20 fragments live in phase 4)
        constexpr bool nn8 = !A_KS && B_KS && VLR_KLOOP_NN8;
        if (rd) rdB(LO{}, fb0);
        PFENCE();
        if constexpr (!nn8) { if (rd) rdAr(LO{}, I2_{}, I4_{}, fa0, 0); }
        PFENCE();
        stage(kt + 1, H_AHI{}, fastc);
        PFENCE();
        if constexpr (nn8) PWAIT_LGKM(0); else if constexpr (A_KS) PWAIT_LGKM(8); else PWAIT_LGKM(4);      // the B reads are retired (B-lo is re-staged in phase 2); the A reads may be in flight
        LBAR();
        PMFMA(fa0, fb0, 0, 0);
        LBAR();
        // ---------------- phase 2
        if (rd) rdB(HI{}, fb1);
        PFENCE();
        stage(kt + 2, H_BLO{}, fastc);
        LBAR();
        PMFMA(fa0, fb1, 0, 1);
        LBAR();
        // ---------------- phase 3
        if (rd) rdA(HI{}, fa1);
        PFENCE();
        stage(kt + 2, H_ALO{}, fastc);
        PFENCE();
        if (more) PWAIT_VM(8); else PWAIT_VM(0);                    // A-lo (and B-lo) of tile kt+1 have landed
        LBAR();
        PMFMA(fa1, fb1, 1, 1);
        LBAR();
        // ---------------- phase 4
        if constexpr (decltype(prec)::value) {      // (not in the last steady-state iteration: the classic body that follows reads A0 itself)
            if constexpr (nn8) { if (rd) rdAr(LO{}, I0_{}, I4_{}, fa0, flip); } else { if (rd) rdAr(LO{}, I0_{}, I2_{}, fa0, flip); }
        }
        PFENCE();
        stage(kt + 2, H_BHI{}, fastc);
        PFENCE();
        if (more) PWAIT_VM(6); else PWAIT_VM(0);
        LBAR();
        PMFMA(fa1, fb0, 1, 0);
        LBAR();
        }
#pragma unroll
        for (int q = 0; q < NPA; ++q) pa[q] += flip;
#pragma unroll
        for (int q = 0; q < NPB; ++q) pb[q] += flip;
        flip = -flip;
    };
#endif
    auto ktile = [&](int kt, auto fastc) {
        const bool rd = !abl_rd || kt == 0;
        // ---------------- phase 1: B0 (4 reads, retired first), A0 (8 reads); stage A-hi of tile kt+1
        if (rd) rdB(LO{}, fb0);
        PFENCE();
        if (rd) rdA(LO{}, fa0);
        PFENCE();
        stage(kt + 1, H_AHI{}, fastc);
        PFENCE();
        if constexpr (A_KS) PWAIT_LGKM(15);   // 16 A reads (two per fragment); the counter holds 15: the B reads are retired
        else PWAIT_LGKM(8);
        LBAR();
        PMFMA(fa0, fb0, 0, 0);
        LBAR();
        // ---------------- phase 2: B1; stage B-lo of tile kt+2
        if (rd) rdB(HI{}, fb1);
        PFENCE();
        stage(kt + 2, H_BLO{}, fastc);
        LBAR();
        PMFMA(fa0, fb1, 0, 1);
        LBAR();
        // ---------------- phase 3: A1; stage A-lo of tile kt+2
        if (rd) rdA(HI{}, fa1);
        PFENCE();
        stage(kt + 2, H_ALO{}, fastc);
        LBAR();
        PMFMA(fa1, fb1, 1, 1);
        LBAR();
        // ---------------- phase 4: no reads; stage B-hi of tile kt+2; the counted wait that makes tile kt+1 resident
        stage(kt + 2, H_BHI{}, fastc);
        PFENCE();
        if (decltype(fastc)::value || kt + 2 < ntp || (CONT && has_next)) PWAIT_VM(6); else PWAIT_VM(0);
        LBAR();
        PMFMA(fa1, fb0, 1, 0);
        LBAR();
#pragma unroll
        for (int q = 0; q < NPA; ++q) pa[q] += flip;
#pragma unroll
        for (int q = 0; q < NPB; ++q) pb[q] += flip;
        flip = -flip;
    };
    // steady state: every staged tile (kt+1, kt+2) exists and is full -> no checks, no K-tail path in the hot loop
    // (absolute K tiles kb + kt + 2 below `lim` are whole tiles of (A, B); SEG: K % 64 == 0, the adapter tiles too when K2 % 64 == 0)
    int kt = 0;
#if VLR_KLOOP_BAL
#ifdef VLR_GEMM_TRACE
    if (n_fast > 1) { ktile_bal(0, std::true_type{}); kt = 1; TSTAMP(titer * 8 + 5); }
#endif
    for (; kt < n_fast - 1; ++kt) ktile_bal(kt, std::true_type{});
    if (kt < n_fast) { ktile_bal(kt, std::false_type{}); ++kt; }
#else
#ifdef VLR_GEMM_TRACE
    if (n_fast > 0) { ktile(0, FAST{}); kt = 1; TSTAMP(titer * 8 + 5); }
#endif
    for (; kt < n_fast; ++kt) ktile(kt, FAST{});
#endif
    for (; kt < ntp; ++kt) ktile(kt, SLOW{});
    TSTAMP(titer * 8 + 6);
    // Waves 4-7 are one barrier behind: when waves 0-3 leave the K loop, waves 4-7 still owe their last barrier - and found its partner
    // only in waves 0-3's FIRST barrier of the next tile, i.e. after waves 0-3's epilogue; then waves 0-3 waited in their second barrier
    // for the epilogue of waves 4-7.  The two halves of the workgroup ran their epilogues one after the other, each with the other half
    // idle (tile timeline, tools/gemm_tile_trace.py: "first K tile of the next tile" = one epilogue longer than a K tile, for every
    // epilogue; 22 - 30 us per tile for the fp32-residual, RoPE and SwiGLU-backward ones).  So: waves 0-3 pay the barrier here, both
    // halves run their epilogues side by side, and waves 4-7 drop back by one barrier before the next tile's first phase.
    if (epi_par) { if (!WAVES_HI()) PBAR(); }
    char* const epatch = smem + 2 * BUF_BYTES + PTAB_BYTES + wave * EPATCH_BYTES;      // CONT only (the per-tile kernels own the K-tile buffers here)
#undef PMFMA
    if constexpr (CONT && FUSE == 1) {
        // SwiGLU epilogue: acc[a][i][0][j] = gate, acc[a][i][1][j] = the matching up columns; act = silu(gate) * up from the fp32
        // accumulators (one rounding), gate | up stored for the backward only when asked.  The arithmetic runs in the accumulator
        // layout (gate and up of an element sit in one lane); the rounded 16 x 32 chunks go through the patch and leave as 16-byte
        // stores, 4 lanes = 64 B per row (see the plain epilogue below).
        bf16_t* C = reinterpret_cast<bf16_t*>(tp.C);
        bf16_t* C2 = reinterpret_cast<bf16_t*>(p.C2);
        const int I = tp.N >> 1;
        // chunk n = (row block r = (a, i), kind): kind 0 act, 1 gate, 2 up (NK = 1: act only - the no-grad passes); chunk n is put while
        // chunk n - 1 comes back from the patch (EPI_BF16_PUT)
        auto body = [&](auto nkc) {
            constexpr int NK = decltype(nkc)::value;
            u32x4 pend[2];
            static_for<0, 8 * NK + 1>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                if constexpr (n < 8 * NK) {
                    constexpr int r = n / NK, kind = n % NK, a = r >> 2, i = r & 3;
                    if constexpr (kind == 0) {
                        f32x4 h[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 g = acc[a][i][0][j], u = acc[a][i][1][j];
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[j][e] = g[e] * fast_sigmoid(g[e]) * u[e];
                        }
                        EPI_BF16_PUT(n & 1, widen_pair(h[0], h[1]), pend[n & 1]);
                    } else {
                        EPI_BF16_PUT(n & 1, widen_pair(acc[a][i][kind - 1][0], acc[a][i][kind - 1][1]), pend[n & 1]);
                    }
                }
                if constexpr (n >= 1) {
                    constexpr int c = n - 1, r = c / NK, kind = c % NK, a = r >> 2, i = r & 3;
                    int ln = lane;                       // opaque per chunk: keeps hipcc from carrying 24 hoisted store addresses (it spilled)
                    asm volatile("" : "+v"(ln));
                    const int gm = m0 + a * 128 + wr * 64 + i * 16 + (ln >> 2);
                    const int gn = n0 + wc * 32 + (ln & 3) * 8;
                    if (gm < tp.M && gn + 8 <= I) {
                        if constexpr (kind == 0) EPI_GST(u32x4, C2 + (size_t)gm * p.ldc2 + gn, pend[c & 1]);
                        else EPI_GST(u32x4, C + (size_t)gm * tp.ldc + (kind - 1) * I + gn, pend[c & 1]);
                    }
                }
            });
        };
        if (p.store_c) body(std::integral_constant<int, 3>{}); else body(std::integral_constant<int, 1>{});
    } else if constexpr (CONT && FUSE == 4) {
        // lm-head forward: per row, this wave's 64 columns (wc*32..+32 of both B halves) -> (max, sum exp) partial + the target logit
        const int lm_ = lane & 15, lq_ = lane >> 4;
        const int nparts = tiles_n * 4;
        const int part = (n0 / PT) * 4 + wc;
        float* __restrict__ parts = reinterpret_cast<float*>(p.C2);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gm = m0 + a * 128 + wr * 64 + i * 16 + lm_;
                const int tg = gm < tp.M ? p.pos[gm] : -1;
                float mx = -INFINITY;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int gn = n0 + b * 128 + wc * 32 + j * 16 + 4 * lq_;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[a][i][b][j][e];
                            if (gn + e < tp.N) mx = fmaxf(mx, v);
                            if (gn + e == tg) p.f1[gm] = v;               // exactly one lane of one workgroup owns the target column
                        }
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float sm = 0.f;
                if (mx > -INFINITY) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int gn = n0 + b * 128 + wc * 32 + j * 16 + 4 * lq_;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (gn + e < tp.N) sm += __expf(acc[a][i][b][j][e] - mx);
                        }
                }
                sm += __shfl_xor(sm, 16);
                sm += __shfl_xor(sm, 32);
                if (lq_ == 0 && gm < tp.M) {
                    f32x2 o = {mx, sm};
                    *reinterpret_cast<f32x2*>(parts + ((size_t)gm * nparts + part) * 2) = o;
                }
            }
    } else if constexpr (CONT && FUSE == 5) {
        // lm-head backward: d logits = g_row * ([col == target] - exp(logit - lse_row)), rounded once, 16-byte stores
        bf16_t* C = reinterpret_cast<bf16_t*>(tp.C);
        const int lm_ = lane & 15, lq_ = lane >> 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gm = m0 + a * 128 + wr * 64 + i * 16 + lm_;
                const int gmc = gm < tp.M ? gm : tp.M - 1;
                const int tg = p.pos[gmc];
                const float z = p.f0[gmc], g = p.f1[gmc];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 d[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int gn0 = n0 + b * 128 + wc * 32 + j * 16 + 4 * lq_;
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[j][e] = g * ((gn0 + e == tg ? 1.f : 0.f) - __expf(acc[a][i][b][j][e] - z));
                    }
                    const int gn = n0 + b * 128 + wc * 32 + widen_col(lq_);
                    const u32x4 w = widen_pair(d[0], d[1]);
                    if (gm < tp.M && gn + 8 <= tp.N) EPI_GST(u32x4, C + (size_t)gm * tp.ldc + gn, w);
                }
            }
    } else if constexpr (CONT && FUSE == 3) {
        // SwiGLU backward epilogue (dgrad of the down projection, NN): acc = d act [rows][cols of I]; gate | up of the forward sit in
        // C2 [M][2I] and are overwritten IN PLACE with d gate | d up (each element is read and written by the same lane).  The fp32
        // d act chunks go through the patch (row-contiguous: lane -> rows er, er + 8, 4 columns), gate | up are read and d gate | d up
        // written as 8-byte accesses, 8 lanes = 64 B per row; the loads of an A half are issued together before its arithmetic.
        // (History: the first version regrouped d act with v_permlane16_swap to the layout of 16-byte row-per-lane loads - hipcc 7.2
        // folded the four swaps on one accumulator into one)
        bf16_t* GU = reinterpret_cast<bf16_t*>(p.C2);
        const int I = tp.N;
        // per A half: the gate | up loads of its 8 chunks together (16 bytes = 8 columns per lane: row lane >> 2), then transposition,
        // arithmetic and two 16-byte stores chunk by chunk
        const int ec8 = (lane & 3) * 8, er8 = lane >> 2;
        // Round 6: every tile as straight-line code with BUFFER addressing (VLR_EPI_FAST; 0 = the round-5 body below).  A predicated store
        // is a branch, and where it joins hipcc has to assume the store was not issued: the counted waits of the round-5 body (vmcnt(15),
        // (14), (13), ...) reach into the stores of earlier chunks, and the second A half's loads were only issued behind all sixteen
        // stores of the first (two exposed round trips per tile, ISA of round 5).  Here the edges cost no predicate: the tile's
        // descriptor ends behind the last valid row (rows past M: loads return 0, stores are dropped by the range check) and a lane
        // whose 8 columns lie past I carries an offset beyond every descriptor.  So the bookkeeping is exact, and the loads run as a
        // RING D chunks ahead - chunk n + D is requested into the registers chunk n has just given up, in front of chunk n's stores:
        // no wait of the tile ever covers a store.  D = 7 (gate | up: 56 registers; 8 spilled one chunk) without an addend, 5 (gate | up | addend: 60) with
        // the LoRA term of down_proj.  In place: a chunk's loads precede its own stores in program order.
        if (VLR_EPI_FAST) {
            auto ring = [&](auto addc) {
                constexpr bool ADD = decltype(addc)::value;
                constexpr int D = ADD ? 5 : 7;
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int rows_v = min(tp.M - m0, PT);                               // >= 1: valid rows of this tile
                const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(GU + (size_t)m0 * p.ldc2, 0, rows_v * p.ldc2 * 2, 0x00020000);
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ADD ? p.residual + (size_t)m0 * p.ldr : GU), 0,
                                                                                    ADD ? rows_v * p.ldr * 2 : 0, 0x00020000);
                const int row_l = wr * 64 + (ln >> 2), col_l = n0 + wc * 32 + (ln & 3) * 8;
                // per-lane byte offsets: gate / up of the two B halves (+ addend); 0x80000000 = past every descriptor, no wrap with soffset
                uint32_t vg[2], vu[2], va[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bool okc = col_l + b * 128 + 8 <= I;
                    vg[b] = okc ? (uint32_t)((row_l * p.ldc2 + col_l + b * 128) * 2) : 0x80000000u;
                    vu[b] = okc ? vg[b] + (uint32_t)(I * 2) : 0x80000000u;
                    va[b] = okc && ADD ? (uint32_t)((row_l * p.ldr + col_l + b * 128) * 2) : 0x80000000u;
                }
                // the chunk's row offset rides soffset (uniform), clamped to num_records: the range check compares the VGPR part with
                // num_records - soffset, UNSIGNED - a chunk wholly past the last valid row must not wrap it
                const uint32_t nrg = (uint32_t)(rows_v * p.ldc2 * 2), nra = ADD ? (uint32_t)(rows_v * p.ldr * 2) : 0u;
                u32x4 gq[D], uq[D], aq[ADD ? D : 1];
                auto request = [&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    constexpr int a = n >> 3, c = n & 7, i = c >> 1, b = c & 1, sl = n % D;
                    const uint32_t sg_ = min((uint32_t)((a * 128 + i * 16) * p.ldc2 * 2), nrg);
                    gq[sl] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                    uq[sl] = gq[sl];
                    if (EPI_LD_ON) {
                        gq[sl] = __builtin_amdgcn_raw_buffer_load_b128(rg, vg[b], sg_, 0);
                        uq[sl] = __builtin_amdgcn_raw_buffer_load_b128(rg, vu[b], sg_, 0);
                    }
                    if constexpr (ADD) aq[sl] = __builtin_amdgcn_raw_buffer_load_b128(ra, va[b], min((uint32_t)((a * 128 + i * 16) * p.ldr * 2), nra), 0);
                };
                static_for<0, D>(request);
                static_for<0, 16>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    constexpr int a = n >> 3, c = n & 7, i = c >> 1, b = c & 1, sl = n % D;
                    f32x4 o[2];
                    EPI_XPOSE_F32_8(acc[a][i][b][0], acc[a][i][b][1], o);
                    float d[8] = {o[0][0], o[0][1], o[0][2], o[0][3], o[1][0], o[1][1], o[1][2], o[1][3]};
                    if constexpr (ADD) {      // + addend on d act (the LoRA term of down_proj)
                        float ad[8];
                        unpack8(aq[sl], ad);
#pragma unroll
                        for (int e = 0; e < 8; ++e) d[e] += ad[e];
                    }
                    float g[8], u[8], dg[8], du[8];
                    unpack8(gq[sl], g);
                    unpack8(uq[sl], u);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sg = fast_sigmoid(g[e]);
                        du[e] = d[e] * g[e] * sg;
                        dg[e] = d[e] * u[e] * sg * (1.f + g[e] * (1.f - sg));
                    }
                    const u32x4 wg = pack8(dg), wu = pack8(du);
                    if constexpr (n + D < 16) request(std::integral_constant<int, n + D>{});
                    if (EPI_ST_ON) {
                        const uint32_t sg_ = min((uint32_t)((a * 128 + i * 16) * p.ldc2 * 2), nrg);
                        __builtin_amdgcn_raw_buffer_store_b128(wg, rg, vg[b], sg_, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(wu, rg, vu[b], sg_, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            if (p.residual) ring(std::true_type{}); else ring(std::false_type{});
        } else
        static_for<0, 2>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            u32x4 gq[8], uq[8];
            static_for<0, 8>([&](auto cc) {      // unpredicated, clamped at the edges
                constexpr int c = decltype(cc)::value;
                constexpr int i = c >> 1, b = c & 1;
                int gm = m0 + a * 128 + wr * 64 + i * 16 + er8;
                int gn = n0 + b * 128 + wc * 32 + ec8;
                gm = gm < tp.M ? gm : tp.M - 1;
                gn = gn + 8 <= I ? gn : I - 8;
                gq[c] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                uq[c] = gq[c];
                if (EPI_LD_ON) {
                    gq[c] = *reinterpret_cast<const u32x4*>(GU + (size_t)gm * p.ldc2 + gn);
                    uq[c] = *reinterpret_cast<const u32x4*>(GU + (size_t)gm * p.ldc2 + I + gn);
                }
            });
            static_for<0, 8>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int i = c >> 1, b = c & 1;
                f32x4 o[2];
                EPI_XPOSE_F32_8(acc[a][i][b][0], acc[a][i][b][1], o);
                EPI_USE(gq[c]); EPI_USE(uq[c]);
                int lc = lane;                        // opaque per chunk: the chunks' addresses are recomputed, not carried (they spilled beside the interior body)
                asm volatile("" : "+v"(lc));
                const int gm = m0 + a * 128 + wr * 64 + i * 16 + (lc >> 2);
                const int gn = n0 + b * 128 + wc * 32 + (lc & 3) * 8;
                const bool ok = gm < tp.M && gn + 8 <= I;
                float d[8] = {o[0][0], o[0][1], o[0][2], o[0][3], o[1][0], o[1][1], o[1][2], o[1][3]};
                if (p.residual && ok) {     // + addend on d act (the LoRA term of down_proj)
                    float ad[8];
                    unpack8(*reinterpret_cast<const u32x4*>(p.residual + (size_t)gm * p.ldr + gn), ad);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] += ad[e];
                }
                float g[8], u[8], dg[8], du[8];
                unpack8(gq[c], g);
                unpack8(uq[c], u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sg = fast_sigmoid(g[e]);
                    du[e] = d[e] * g[e] * sg;
                    dg[e] = d[e] * u[e] * sg * (1.f + g[e] * (1.f - sg));
                }
                if (EPI_ST_ON && ok) {
                    EPI_GST(u32x4, GU + (size_t)gm * p.ldc2 + gn, pack8(dg));
                    EPI_GST(u32x4, GU + (size_t)gm * p.ldc2 + I + gn, pack8(du));
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    } else if constexpr (CONT && FUSE == 2) {
        // RoPE epilogue: acc[a][i][0][j] / acc[a][i][1][j] = features d / d + 64 of one head (rotate-half partners).  Both fp32 chunks go
        // through the patch (row-contiguous: lane -> rows er, er + 8; features fc .. fc + 3 of the head's first / second 64), so that the
        // cos / sin rows of the positions are read as whole 128-byte lines (8 lanes x 16 B) and the rotated values leave as 8-byte
        // stores, 64 B per row; the positions of all 16 rows of the lane first, then the table rows of one (a, i) ahead of its arithmetic.
        bf16_t* C = reinterpret_cast<bf16_t*>(tp.C);
        const bool rot = n0 < p.rope_cols;            // q and k tiles; v tiles pass through
        const int ec8 = (lane & 3) * 8, er8 = lane >> 2;
        const int fc = (wc & 1) * 32 + ec8;           // feature inside the 64-wide half (8 of them per lane)
        const int hc = n0 + (wc >> 1) * 128 + fc;     // output column of the first-half features (second half: + 64)
        float b1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias) {                                  // biased fused projection (Qwen c_attn): added before the rotation
            const u32x4 w1 = *reinterpret_cast<const u32x4*>(p.bias + hc), w2 = *reinterpret_cast<const u32x4*>(p.bias + hc + 64);
            EPI_USE(w1); EPI_USE(w2);
            unpack8(w1, b1);
            unpack8(w2, b2);
        }
        // per A half: the positions of the lane's 4 rows (one per row block i), the cos / sin rows of those blocks (two 16-byte loads
        // each: 4 lanes read the 128-byte line of a position's features), then transposition (8 columns per lane), rotation and the two
        // 16-byte stores block by block
        auto body = [&](auto rotc) {
            constexpr bool ROT = decltype(rotc)::value;
            static_for<0, 2>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                f32x4 cs[4][2], sn[4][2];
                if constexpr (ROT) {
                    int ps[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int gm = m0 + a * 128 + wr * 64 + i * 16 + er8;
                        gm = gm < tp.M ? gm : tp.M - 1;          // rows past the edge take the last row's position, and are not stored
                        int q = p.pos[gm];
                        q = q < 0 ? 0 : (q >= p.max_pos ? p.max_pos - 1 : q);
                        ps[i] = q;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            cs[i][k] = *reinterpret_cast<const f32x4*>(p.rope_cos + (size_t)(ps[i] * 64 + fc + 4 * k));
                            sn[i][k] = *reinterpret_cast<const f32x4*>(p.rope_sin + (size_t)(ps[i] * 64 + fc + 4 * k));
                        }
                }
                static_for<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    f32x4 o1[2], o2[2];
                    EPI_XPOSE_F32_8(acc[a][i][0][0], acc[a][i][0][1], o1);
                    EPI_XPOSE_F32_8(acc[a][i][1][0], acc[a][i][1][1], o2);
                    if constexpr (ROT) { EPI_USE(cs[i][0]); EPI_USE(cs[i][1]); EPI_USE(sn[i][0]); EPI_USE(sn[i][1]); }
                    const int gm = m0 + a * 128 + wr * 64 + i * 16 + er8;
                    float y1[8], y2[8];
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x1 = o1[k][e] + b1[4 * k + e], x2 = o2[k][e] + b2[4 * k + e];
                            if constexpr (ROT) {
                                y1[4 * k + e] = x1 * cs[i][k][e] - x2 * sn[i][k][e];
                                y2[4 * k + e] = x2 * cs[i][k][e] + x1 * sn[i][k][e];
                            } else {
                                y1[4 * k + e] = x1;
                                y2[4 * k + e] = x2;
                            }
                        }
                    if (gm < tp.M) {
                        EPI_GST(u32x4, C + (size_t)gm * tp.ldc + hc, pack8(y1));
                        EPI_GST(u32x4, C + (size_t)gm * tp.ldc + hc + 64, pack8(y2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        };
        if (rot) body(std::true_type{}); else body(std::false_type{});        // (N % 256 == 0 is a launch condition of this epilogue)
    } else if constexpr (CONT) {
      // Staged through a wave-private LDS patch (EPI_* above): a lane of the accumulator layout holds 4 columns of ONE row per 16-lane
      // group, so a 16-byte access per lane straight from that layout is 16 different cache lines per group - 64 requests of 16 bytes per
      // wave instruction, and the texture path takes about one request per clock (tile timeline: 23 us per tile for the fp32-residual
      // epilogue, the same whether the two wave groups run it one after the other or side by side, and whether or not the other CUs
      // are in their epilogues at that moment).  A 16-row x 32-column chunk (the two accumulator tiles j = 0, 1) goes through the patch
      // and comes back row-contiguous: 8 lanes x 16 B = one 128-byte line per row (fp32), 4 lanes x 16 B = 64 B per row (bf16) -
      // 8 / 16 requests per instruction.  The patch lives behind the piece table, outside the K-tile buffers: the next tile's first K
      // tiles keep arriving by LDS-DMA underneath.
      if (p.out_f32) {
        // fp32 output (fp32 residual stream: x_new = x + attn Wo^T / x + act Wdown^T), nothing is rounded
        float* C = reinterpret_cast<float*>(tp.C);
        const float* R = reinterpret_cast<const float*>(p.residual);
        const int ec = (lane & 7) * 4, er = lane >> 3;
        // Per A half: its 16 residual loads (rows er, er + 8 of the 8 chunks (i, b): whole 128-byte lines) are issued together - the
        // fragment registers of the K loop are dead here - then each chunk is transposed, added and stored.  (Loads of later chunks
        // cannot run ahead under stores: with loads AND stores pending hipcc treats vmcnt as unordered and waits vmcnt(0) for every
        // value; parking the sums in the accumulator registers until all loads are in - two passes - made hipcc spill them.)
        auto body = [&]() {
            static_for<0, 2>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                f32x4 rv[8][2];
                static_for<0, 8>([&](auto cc) {      // unpredicated (rows / columns past the edge re-read the last valid ones)
                    constexpr int c = decltype(cc)::value;
                    constexpr int i = c >> 1, b = c & 1;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        int gm = m0 + a * 128 + wr * 64 + i * 16 + er + 8 * k;
                        int gn = n0 + b * 128 + wc * 32 + ec;
                        gm = gm < tp.M ? gm : tp.M - 1;
                        gn = gn + 4 <= tp.N ? gn : tp.N - 4;
                        rv[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (EPI_LD_ON) rv[c][k] = *reinterpret_cast<const f32x4*>(R + (size_t)gm * p.ldr + gn);
                    }
                });
                static_for<0, 8>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    constexpr int i = c >> 1, b = c & 1;
                    f32x4 o[2];
                    EPI_XPOSE_F32(acc[a][i][b][0], acc[a][i][b][1], o);
                    EPI_USE(rv[c][0]); EPI_USE(rv[c][1]);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int gm = m0 + a * 128 + wr * 64 + i * 16 + er + 8 * k;
                        const int gn = n0 + b * 128 + wc * 32 + ec;
                        const f32x4 v = p.alpha * o[k] + rv[c][k];
                        if (EPI_ST_ON && gm < tp.M && gn + 4 <= tp.N) EPI_GST(f32x4, C + (size_t)gm * tp.ldc + gn, v);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // one chunk's addresses at a time
                });
            });
        };
        auto body_nores = [&]() {      // no residual (not on the 7B path): transposition and store per chunk
            static_for<0, 16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int a = c >> 3, i = (c >> 1) & 3, b = c & 1;
                f32x4 o[2];
                EPI_XPOSE_F32(acc[a][i][b][0], acc[a][i][b][1], o);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int gm = m0 + a * 128 + wr * 64 + i * 16 + er + 8 * k;
                    const int gn = n0 + b * 128 + wc * 32 + ec;
                    if (gm < tp.M && gn + 4 <= tp.N) EPI_GST(f32x4, C + (size_t)gm * tp.ldc + gn, p.alpha * o[k]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if (R) body(); else body_nores();
      } else if (p.residual) {
        // bf16 output with a bf16 residual (adapter-segment launches on the bf16 stream): fp32 chunks through the patch, residual added
        // in fp32 before the single rounding, 8-byte accesses (8 lanes = 64 B per row)
        bf16_t* C = reinterpret_cast<bf16_t*>(tp.C);
        const int ec = (lane & 7) * 4, er = lane >> 3;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 o[2];
                    EPI_XPOSE_F32(acc[a][i][b][0], acc[a][i][b][1], o);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int gm = m0 + a * 128 + wr * 64 + i * 16 + er + 8 * k;
                        const int gn = n0 + b * 128 + wc * 32 + ec;
                        if (gm < tp.M && gn + 4 <= tp.N) {
                            f32x4 v = p.alpha * o[k];
                            const u32x2 w = *reinterpret_cast<const u32x2*>(p.residual + (size_t)gm * p.ldr + gn);
                            v[0] += bf16lo(w[0]); v[1] += bf16hi(w[0]); v[2] += bf16lo(w[1]); v[3] += bf16hi(w[1]);
                            u32x2 q;
                            q[0] = pack_bf16(v[0], v[1]); q[1] = pack_bf16(v[2], v[3]);
                            *reinterpret_cast<u32x2*>(C + (size_t)gm * tp.ldc + gn) = q;
                        }
                    }
                }
      } else {
        // plain bf16 epilogue: rounded in the accumulator layout, the packed 16-row x 32-column chunks (1 KiB) through the patch, ONE
        // 16-byte store per lane and chunk (lane -> row lane >> 2, columns (lane & 3) * 8 .. + 7); chunk n is put while chunk n - 1 returns
        bf16_t* C = reinterpret_cast<bf16_t*>(tp.C);
        u32x4 pend[2];
        static_for<0, 17>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n < 16) {
                constexpr int a = n >> 3, i = (n >> 1) & 3, b = n & 1;
                EPI_BF16_PUT(n & 1, widen_pair(p.alpha * acc[a][i][b][0], p.alpha * acc[a][i][b][1]), pend[n & 1]);
            }
            if constexpr (n >= 1) {
                constexpr int c = n - 1, a = c >> 3, i = (c >> 1) & 3, b = c & 1;
                const int gm = m0 + a * 128 + wr * 64 + i * 16 + (lane >> 2);
                const int gn = n0 + b * 128 + wc * 32 + (lane & 3) * 8;
                if (gm < tp.M && gn + 8 <= tp.N) EPI_GST(u32x4, C + (size_t)gm * tp.ldc + gn, pend[c & 1]);
            }
        });
      }
    }
    if constexpr (CONT) {
        TSTAMP(titer * 8 + 7);
        parb = (parb + ntp) & 1;
        if (!has_next) {
            if (!WAVES_HI() && !epi_par) PBAR();  // balance the extra barrier of waves 4-7
            break;
        }
        continue;
    }
    if (!WAVES_HI()) PBAR();          // balance the extra barrier of waves 4-7
    __syncthreads();

    // ---- epilogue.  The MFMAs were issued as (B fragment, A fragment), i.e. they accumulated C^T tiles: lane l of a 16x16
    // tile holds C[m = l&15][n = 4*(l>>4) .. +3] - four CONSECUTIVE columns of one row.
    const int lm = lane & 15, lq = lane >> 4;
    const bool skip_epilogue = (ABL & 8) && p.alpha != 12345.f;   // diagnostics: no epilogue (keeps the accumulators live)
    if (skip_epilogue) {
    } else if (!p.out_f32) {
        // bf16 output: every wave applies the fp32 epilogue to its quadrants (alpha, bias, activation, residual, accumulate:
        // 8-byte global reads), rounds ONCE and writes 8-byte pieces of a bf16 image of the whole 256x256 C tile in LDS; after a
        // barrier the block copies the image out as full 512-byte rows (two rows per wave instruction, 16 B per lane).
        // (The per-quadrant version wrote 64-byte row pieces straight from each wave: 12 us per tile, measured by ablation.)
        const bf16_t* Cold = reinterpret_cast<const bf16_t*>(tp.C);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tn_ = b * 128 + wc * 32 + j * 16 + 4 * lq;      // column inside the tile
                    const int gn = n0 + tn_;
                    const bool nok = gn + 4 <= tp.N;
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias && nok) {
                        const u32x2 w = *reinterpret_cast<const u32x2*>(p.bias + gn);
                        bv[0] = bf16lo(w[0]); bv[1] = bf16hi(w[0]); bv[2] = bf16lo(w[1]); bv[3] = bf16hi(w[1]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int tm_ = a * 128 + wr * 64 + i * 16 + lm;      // row inside the tile
                        const int gm = m0 + tm_;
                        f32x4 v = acc[a][i][b][j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(p.alpha * v[e] + bv[e], p.act);
                        if constexpr (FUSE == 6) {
                            // keep mask of elements (gm, gn .. gn+3): half a hash group (gn % 4 == 0, drop_ld % 8 == 0)
                            const long idx = (long)gm * p.drop_ld + gn;
                            const uint64_t rr = vlr_mix64(p.drop_key ^ (uint64_t)(2 * (idx >> 3) + ((idx >> 2) & 1)));
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if ((uint32_t)((rr >> (16 * e)) & 0xffffu) < p.drop_thr) v[e] = 0.f;
                        }
                        if (gm < tp.M && nok) {
                            if (p.residual) {
                                const u32x2 w = *reinterpret_cast<const u32x2*>(p.residual + (size_t)gm * p.ldr + gn);
                                v[0] += bf16lo(w[0]); v[1] += bf16hi(w[0]); v[2] += bf16lo(w[1]); v[3] += bf16hi(w[1]);
                            }
                            if (FUSE != 6 && p.accumulate) {      // fuse 6 accumulates in the coalesced copy-out below
                                const u32x2 w = *reinterpret_cast<const u32x2*>(Cold + (size_t)gm * tp.ldc + gn);
                                v[0] += bf16lo(w[0]); v[1] += bf16hi(w[0]); v[2] += bf16lo(w[1]); v[3] += bf16hi(w[1]);
                            }
                        }
                        u32x2 o;
                        o[0] = pack_bf16(v[0], v[1]);
                        o[1] = pack_bf16(v[2], v[3]);
                        *reinterpret_cast<u32x2*>(smem + tm_ * C_STRIDE + tn_ * 2) = o;
                    }
                }
            }
        __syncthreads();
        bf16_t* C = reinterpret_cast<bf16_t*>(tp.C);
        const int cch = lane & 31;                     // 16-byte chunk of the 512-byte row
        const int gn = n0 + cch * 8;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = it * 16 + wave * 2 + (lane >> 5);
            const int gm = m0 + row;
            if (gm < tp.M && gn + 8 <= tp.N) {
                u32x4 w = *reinterpret_cast<const u32x4*>(smem + row * C_STRIDE + cch * 16);
                if constexpr (FUSE == 6) {
                    // C += (rounded masked product): full 512-byte rows read and written once, 16 B per lane (the K loop is two
                    // tiles long - this read-modify-write IS the kernel); same two roundings as product -> scratch -> add
                    float o[8], d[8];
                    unpack8(*reinterpret_cast<const u32x4*>(C + (size_t)gm * tp.ldc + gn), o);
                    unpack8(w, d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += d[e];
                    w = pack8(o);
                }
                *reinterpret_cast<u32x4*>(C + (size_t)gm * tp.ldc + gn) = w;
            }
        }
    } else {
    // fp32 output (lm-head logits): one 64x32 quadrant at a time through a wave-private fp32 patch, 16-byte global accesses
    constexpr int PS = 36;   // patch row stride in floats
    float* patch = reinterpret_cast<float*>(smem) + wave * (64 * PS);
    auto quadrant = [&](auto ac, auto bc) {
        constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(patch + (i * 16 + lm) * PS + j * 16 + 4 * lq) = acc[a][i][b][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int gm0 = m0 + a * 128 + wr * 64, gn0 = n0 + b * 128 + wc * 32;
        float* C = reinterpret_cast<float*>(tp.C);
        const int cq = (lane & 7) * 4;
        const int gn = gn0 + cq;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && gn + 4 <= tp.N) {
            const u32x2 w = *reinterpret_cast<const u32x2*>(p.bias + gn);
            bv[0] = bf16lo(w[0]); bv[1] = bf16hi(w[0]); bv[2] = bf16lo(w[1]); bv[3] = bf16hi(w[1]);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int gm = gm0 + row;
            if (gm < tp.M && gn + 4 <= tp.N) {
                f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * PS + cq);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(p.alpha * v[e] + bv[e], p.act);
                if (p.residual) {
                    if (p.res_f32) {
                        v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.residual) + (size_t)gm * p.ldr + gn);
                    } else {
                        const u32x2 w = *reinterpret_cast<const u32x2*>(p.residual + (size_t)gm * p.ldr + gn);
                        v[0] += bf16lo(w[0]); v[1] += bf16hi(w[0]); v[2] += bf16lo(w[1]); v[3] += bf16hi(w[1]);
                    }
                }
                float* dst = C + (size_t)gm * tp.ldc + gn;
                if (p.accumulate) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += o[e];
                }
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    quadrant(I0{}, I0{});
    quadrant(I0{}, I1{});
    quadrant(I1{}, I0{});
    quadrant(I1{}, I1{});
    }
    __syncthreads();              // every wave is done with the LDS image / patches before the next tile's DMA lands
    if (titer + 1 >= npieces) break;            // (per-tile kernels: whole tiles only)
    }   // persistent tile loop
#ifdef VLR_GEMM_TRACE
    if (CONT && p.trace && wave == 0) {          // [block][64 pieces][4] = {m0 / 256 << 16 | n0 / 128, first K tile, K loop, epilogue}; [block][63] = {npieces, start, nt, 0}
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint32_t* out = p.trace + (size_t)blockIdx.x * 256;
        for (int i = lane0; i < npieces && i < 63; i += 64) {
            out[i * 4 + 0] = ((uint32_t)(ptab[i * 8 + 0] / PT) << 16) | (uint32_t)(ptab[i * 8 + 1] / 128);
            out[i * 4 + 1] = (uint32_t)ptab[i * 8 + 5];
            out[i * 4 + 2] = (uint32_t)ptab[i * 8 + 6];
            out[i * 4 + 3] = (uint32_t)ptab[i * 8 + 7];
        }
        if (lane0 == 0) { out[252] = (uint32_t)npieces; out[253] = (uint32_t)ptab[(PTAB_PIECES - 1) * 8 + 5]; out[254] = (uint32_t)nt; out[255] = 0; }
    }
#endif
}

static int gemm256p_n_cu() { return vlr_compute_cus(); }      // grid of the persistent launches (whole XCD octets; 256 on MI355X)
// grid of a launch over `ntiles` output tiles: the CUs (persistent workgroups, a piece list each) unless the list would not fit the table
static int persist_grid(int ntiles, int n_cu) { return (ntiles > n_cu && ntiles / n_cu + 8 <= PTAB_PIECES) ? n_cu : ntiles; }
static bf16_t* gemm256p_zero16() {
    static bf16_t* z = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        if (hipMalloc((void**)&z, 256) != hipSuccess || hipMemset(z, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) z = nullptr;
    }
    return z;
}

// GemmParams::sched of a launch: bit 3 = adapter K tiles on the general staging path, bit 4 = serial epilogue order (both A/B switches),
// bit 5 = the shared-panel tile map (vlr_gemm_set_sched / VLR_GEMM_SCHED; 32 in production)
static void sched_prepare(GemmParams& p) { p.sched = vlr_gemm_sched_mode() & (24 | 32); }

bool vlr_gemm256p_fused_try_launch(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_GEMM_FUSE");
        on = e ? atoi(e) : 3;               // bit 0 SwiGLU, bit 1 RoPE
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, 0, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, 0, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
    }
    if (p.fuse == 3) return false;       // NN: vlr_gemm256p_swiglu_bwd_try_launch
    if (p.fuse < 1 || p.fuse > 2 || !((on >> (p.fuse - 1)) & 1)) return false;
    bf16_t* zero16 = gemm256p_zero16();
    if (!zero16) return false;
    const int n_cu = gemm256p_n_cu();
    const int tiles_m = (p.M + PT - 1) / PT;
    const int tiles_n = p.fuse == 1 ? ((p.N >> 1) + 127) / 128 : p.N / PT;
    const int ntiles = tiles_m * tiles_n;
    if (p.K < 4 * PK) return false;
    const int grid = persist_grid(ntiles, n_cu);       // few tiles (small batches, the peeled last tile rows): one tile per workgroup
    if (p.fuse == 2 && (p.N % PT != 0 || p.rope_cols % PT != 0)) return false;
    if (p.fuse == 1 && ((p.N >> 1) % 8 != 0 || p.ldc2 % 8 != 0 || ((uintptr_t)p.C2 & 15))) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.ldc % 8 != 0) return false;
    sched_prepare(p);
    TRACE_SET(p);
    const int pi = vlr_prof_begin(VLR_K_GEMM256P, 2.0 * p.M * p.N * p.K, stream);
    if (p.fuse == 1) hipLaunchKernelGGL((gemm256p_kernel<false, false, 0, true, 1>), dim3(grid), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    else hipLaunchKernelGGL((gemm256p_kernel<false, false, 0, true, 2>), dim3(grid), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    vlr_prof_end(pi, stream);
    return true;
}

// C = A B^T + A2 B2^T (adapter segment, GemmParams::A2...), NT, persistent continuous pipeline only: fuse 0 (plain, optional
// residual), 1 (SwiGLU), 2 (RoPE).  false: the caller runs the base GEMM and the adapter GEMMs separately.
bool vlr_gemm256p_seg_try_launch(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    sched_prepare(p);
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_GEMM_SEG");
        on = (e && e[0] == '0') ? 0 : 1;
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, 0, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, 0, true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, 0, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
    }
    if (!on || p.fuse < 0 || p.fuse > 2 || p.K2 <= 0 || !p.A2 || !p.B2) return false;
    bf16_t* zero16 = gemm256p_zero16();
    if (!zero16) return false;
    const int n_cu = gemm256p_n_cu();
    const int tiles_m = (p.M + PT - 1) / PT;
    const int tiles_n = p.fuse == 1 ? ((p.N >> 1) + 127) / 128 : (p.N + PT - 1) / PT;
    if (p.K < 4 * PK || p.K % PK != 0) return false;
    const int grid = persist_grid(tiles_m * tiles_n, n_cu);
    if (p.K2 % 8 != 0 || p.lda2 % 8 != 0 || p.ldb2 % 8 != 0 || (((uintptr_t)p.A2 | (uintptr_t)p.B2) & 15)) return false;
    if (p.fuse != 1 && ((p.seg_b0 < p.N && p.seg_b0 % PT != 0) || (p.seg_b1 < p.N && p.seg_b1 % PT != 0))) return false;   // a tile lies in one block
    if (p.fuse == 2 && (p.N % PT != 0 || p.rope_cols % PT != 0)) return false;
    if (p.fuse == 1 && ((p.N >> 1) % 8 != 0 || p.ldc2 % 8 != 0 || ((uintptr_t)p.C2 & 15))) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.ldc % 8 != 0 || p.N % 8 != 0) return false;
    if ((p.bias && p.fuse != 2) || p.accumulate || p.act != ACT_NONE) return false;
    if (p.bias && ((uintptr_t)p.bias & 7)) return false;
    if (p.out_f32) {          // fp32 residual stream: y fp32 = x W^T + u Bl^T + residual fp32 (register-direct 16-byte accesses)
        if (p.fuse != 0 || p.ldc % 4 != 0 || (p.residual && (!p.res_f32 || p.ldr % 4 != 0 || ((uintptr_t)p.residual & 15)))) return false;
    } else if (p.residual && (p.res_f32 || p.fuse != 0 || p.ldr % 4 != 0 || ((uintptr_t)p.residual & 7) || (const void*)p.residual == (const void*)p.C)) return false;
    TRACE_SET(p);
    const int pi = vlr_prof_begin(VLR_K_GEMM256P, 2.0 * p.M * p.N * (p.K + p.K2), stream);
    if (p.fuse == 0) hipLaunchKernelGGL((gemm256p_kernel<false, false, 0, true, 0, true>), dim3(grid), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    else if (p.fuse == 1) hipLaunchKernelGGL((gemm256p_kernel<false, false, 0, true, 1, true>), dim3(grid), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    else hipLaunchKernelGGL((gemm256p_kernel<false, false, 0, true, 2, true>), dim3(grid), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    vlr_prof_end(pi, stream);
    return true;
}

// dx [M][N] += keep ? alpha * (A [M][K] . B [K][N]) : 0 (NN; fuse 6).  Any tile count >= 192 (K is the adapter rank: the launch is
// store-bound, no peeling); false -> the caller materialises the product and runs the dropout-accumulate kernel.
bool vlr_gemm256p_dropacc_try_launch(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    TRACE_SET(p);
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_GEMM_DROPACC");
        on = (e && e[0] == '0') ? 0 : 1;
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, true, 0, false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_LDS_BYTES);
    }
    if (!on || p.fuse != 6) return false;
    bf16_t* zero16 = gemm256p_zero16();
    if (!zero16) return false;
    const int n_cu = gemm256p_n_cu();
    const int ntiles = ((p.M + PT - 1) / PT) * ((p.N + PT - 1) / PT);
    if (ntiles < 192) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15) || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.N % 8 != 0 || p.ldc % 8 != 0 || p.drop_ld % 8 != 0) return false;
    const int pi = vlr_prof_begin(VLR_K_GEMM256P, 2.0 * p.M * p.N * p.K, stream);
    hipLaunchKernelGGL((gemm256p_kernel<false, true, 0, false, 6>), dim3(ntiles < n_cu ? ntiles : n_cu), dim3(512), TILE_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    vlr_prof_end(pi, stream);
    return true;
}

// d act = dy . Wdown (NN) with the SwiGLU backward in the epilogue: p.C2 = gate | up [M][2I] (in/out), p.N = I, p.C unused
bool vlr_gemm256p_swiglu_bwd_try_launch(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_GEMM_FUSE");
        on = e ? ((atoi(e) >> 2) & 1) : 1;               // bit 2
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, true, 0, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
    }
    if (!on) return false;
    bf16_t* zero16 = gemm256p_zero16();
    if (!zero16) return false;
    const int n_cu = gemm256p_n_cu();
    const int ntiles = ((p.M + PT - 1) / PT) * ((p.N + PT - 1) / PT);
    if (p.K < 4 * PK) return false;
    const int grid = persist_grid(ntiles, n_cu);
    if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C2) & 15) || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.N % 8 != 0 || p.ldc2 % 8 != 0) return false;
    sched_prepare(p);
    TRACE_SET(p);
    const int pi = vlr_prof_begin(VLR_K_GEMM256P, 2.0 * p.M * p.N * p.K, stream);
    hipLaunchKernelGGL((gemm256p_kernel<false, true, 0, true, 3>), dim3(grid), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    vlr_prof_end(pi, stream);
    return true;
}

// diagnostics (include/vlr.h): tile timeline of the persistent launches; only the -DVLR_GEMM_TRACE build records anything
extern "C" int vlr_gemm_set_trace(void* buf, long bytes) {
#ifdef VLR_GEMM_TRACE
    VLR_REQUIRE(!buf || bytes >= 256L * 1024, "vlr_gemm_set_trace: the buffer holds 256 words for each of 256 workgroups (256 KiB), got %ld bytes", bytes);
    g_trace = (uint32_t*)buf;
    return VLR_OK;
#else
    (void)buf; (void)bytes;
    VLR_REQUIRE(false, "vlr_gemm_set_trace: this library was built without -DVLR_GEMM_TRACE (python vl-rlhf_amd/build_hip.py --trace)");
#endif
}

// Two TN problems of equal K as ONE persistent launch of the grouped continuous-pipeline kernel (template GRP): C0 = A0^T B0 and
// C1 = A1^T B1, plain bf16 outputs.  false: the caller launches them one by one.
bool vlr_gemm256p_tn_pair_try_launch(const GemmParams& p0, const GemmParams& p1, hipStream_t stream) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_GEMM_PAIR");
        on = (e && e[0] == '0') ? 0 : 1;
        hipFuncSetAttribute((const void*)gemm256p_kernel<true, true, 0, true, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
    }
    if (!on || p0.K != p1.K || p0.K < 4 * PK) return false;
    bf16_t* zero16 = gemm256p_zero16();
    if (!zero16) return false;
    for (const GemmParams* q : {&p0, &p1}) {
        if (q->bias || q->residual || q->accumulate || q->act != ACT_NONE || q->out_f32 || q->alpha != 1.f || q->fuse) return false;
        if ((((uintptr_t)q->A | (uintptr_t)q->B | (uintptr_t)q->C) & 15) || q->lda % 8 != 0 || q->M % 8 != 0 || q->ldb % 8 != 0 || q->N % 8 != 0 || q->ldc % 8 != 0) return false;
    }
    const int n_cu = gemm256p_n_cu();
    const int ntiles = ((p0.M + PT - 1) / PT) * ((p0.N + PT - 1) / PT) + ((p1.M + PT - 1) / PT) * ((p1.N + PT - 1) / PT);
    if (ntiles <= n_cu || persist_grid(ntiles, n_cu) != n_cu) return false;
    GemmParams p = p0;
    p.A1 = p1.A; p.B1 = p1.B; p.C1 = p1.C; p.M1 = p1.M; p.N1 = p1.N; p.lda1 = p1.lda; p.ldb1 = p1.ldb; p.ldc1 = p1.ldc;
    sched_prepare(p);
    TRACE_SET(p);
    const int pi = vlr_prof_begin(VLR_K_GEMM256P, 2.0 * p.K * ((double)p0.M * p0.N + (double)p1.M * p1.N), stream);
    hipLaunchKernelGGL((gemm256p_kernel<true, true, 0, true, 0, false, true>), dim3(n_cu), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    vlr_prof_end(pi, stream);
    return true;
}

int vlr_gemm256p_lmhead_parts(int V) { return ((V + PT - 1) / PT) * 4; }

bool vlr_gemm256p_lmhead_try_launch(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLR_GEMM_FUSE");
        on = e ? ((atoi(e) >> 3) & 1) : 1;               // bit 3
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, 0, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
        hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, 0, true, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
    }
    if (!on || (p.fuse != 4 && p.fuse != 5)) return false;
    bf16_t* zero16 = gemm256p_zero16();
    if (!zero16) return false;
    const int n_cu = gemm256p_n_cu();
    const int ntiles = ((p.M + PT - 1) / PT) * ((p.N + PT - 1) / PT);
    if (ntiles <= n_cu || persist_grid(ntiles, n_cu) != n_cu || p.K < 4 * PK) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.B) & 15) || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.N % 8 != 0) return false;
    if (p.fuse == 5 && (p.ldc % 8 != 0 || ((uintptr_t)p.C & 15))) return false;
    sched_prepare(p);
    TRACE_SET(p);
    const int pi = vlr_prof_begin(VLR_K_GEMM256P, 2.0 * p.M * p.N * p.K, stream);
    if (p.fuse == 4) hipLaunchKernelGGL((gemm256p_kernel<false, false, 0, true, 4>), dim3(n_cu), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    else hipLaunchKernelGGL((gemm256p_kernel<false, false, 0, true, 5>), dim3(n_cu), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    vlr_prof_end(pi, stream);
    return true;
}

// returns false when the problem does not qualify (caller falls back to the staggered / 128x128 kernels)
bool vlr_gemm256p_try_launch(int layout, const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    sched_prepare(p);
    TRACE_SET(p);
    static int mode = -1;
    static bf16_t* zero16 = nullptr;
    if (mode < 0) {
        const char* e = getenv("VLR_GEMM_8PHASE");
        mode = e ? atoi(e) : 7;            // bit 0 NT, bit 1 NN, bit 2 TN
        if (mode) {
            hipFuncSetAttribute((const void*)gemm256p_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_LDS_BYTES);
            hipFuncSetAttribute((const void*)gemm256p_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_LDS_BYTES);
            hipFuncSetAttribute((const void*)gemm256p_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_LDS_BYTES);
            hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
            hipFuncSetAttribute((const void*)gemm256p_kernel<false, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
            hipFuncSetAttribute((const void*)gemm256p_kernel<true, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CONT_LDS_BYTES);
            if (hipMalloc((void**)&zero16, 256) != hipSuccess || hipMemset(zero16, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) mode = 0;
        }
    }
    if (!((mode >> layout) & 1)) return false;
    const int ntiles = ((p.M + PT - 1) / PT) * ((p.N + PT - 1) / PT);
    if (ntiles < 192) return false;
    static int persist = -1;
    if (persist < 0) { const char* e = getenv("VLR_GEMM_PERSIST"); persist = (e && e[0] == '0') ? 0 : 1; }
    const int n_cu = persist ? gemm256p_n_cu() : (1 << 30);
    const int tiles = persist_grid(ntiles, n_cu);   // grid size: persistent workgroups when there are more tiles than CUs
    // 16-byte DMA source alignment: k-contiguous operands need ld % 8 and K % 8 (checked by the caller), k-strided
    // operands ld % 8 and at least 8 columns; pointers 16-byte aligned
    const bool a_ks = layout == 2, b_ks = layout != 0;
    if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return false;
    if (a_ks && (p.lda % 8 != 0 || p.M % 8 != 0)) return false;
    if (b_ks && (p.ldb % 8 != 0 || p.N % 8 != 0)) return false;
    if (!a_ks && p.lda % 8 != 0) return false;
    if (!b_ks && p.ldb % 8 != 0) return false;
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("VLR_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    if (layout == 0 && abl) {
#define PABL(n) case n: hipFuncSetAttribute((const void*)gemm256p_kernel<false, false, n>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_LDS_BYTES); \
        hipLaunchKernelGGL((gemm256p_kernel<false, false, n>), dim3(tiles), dim3(512), TILE_LDS_BYTES, stream, p, (const bf16_t*)zero16); return true;
        switch (abl) { PABL(1) PABL(4) PABL(5) PABL(8) default: break; }   // 2, 3, 6, 7 (no fragment reads) spill; 8 = no epilogue
#undef PABL
    }
    // timed as its own kernel id by the in-library profiler (bench.py quotes the roofline of THIS kernel)
    const int pi = vlr_prof_begin(VLR_K_GEMM256P, 2.0 * p.M * p.N * p.K, stream);
    // continuous pipeline across tiles: persistent launch, plain bf16 epilogue (C = alpha * A B), at least 4 K tiles
    static int cont = -1;
    if (cont < 0) { const char* e = getenv("VLR_GEMM_CONT"); cont = (e && e[0] == '0') ? 0 : 1; }
    static int cont_res = -1;
    // residual epilogue on the continuous pipeline: OFF by default - measured 1.4 ms per step SLOWER than the LDS-image epilogue of
    // the per-tile kernel (635.8 vs 634.4 ms, same box, bit-identical losses): its 8-byte residual reads sit between the K loop and
    // the stores and are not hidden.  (The adapter-segment kernels use it: they exist only as continuous pipelines.)
    if (cont_res < 0) { const char* e = getenv("VLR_GEMM_CONT_RES"); cont_res = (e && e[0] == '1') ? 1 : 0; }
    // (an in-place residual, C == residual, stays on the LDS-image epilogue: the widened store of one lane covers columns another
    // lane still has to read)
    const bool res_ok = !p.residual || (cont_res && p.ldr % 4 == 0 && !((uintptr_t)p.residual & 7) && (const void*)p.residual != (const void*)p.C);
    // fp32 residual stream: C fp32 = acc + residual fp32, 16-byte accesses in the accumulator layout (in place is fine: a lane reads
    // exactly the bytes it writes).  VLR_GEMM_F32RES_CONT=0 sends these launches to the per-tile kernel's fp32 patches instead.
    static int f32_cont = -1;
    if (f32_cont < 0) { const char* e = getenv("VLR_GEMM_F32RES_CONT"); f32_cont = (e && e[0] == '0') ? 0 : 1; }
    const bool f32_ok = p.out_f32 && f32_cont && (!p.residual || (p.res_f32 && p.ldr % 4 == 0 && !((uintptr_t)p.residual & 15))) && p.ldc % 4 == 0 && p.N % 4 == 0;
    if (cont && ntiles > tiles && !p.bias && (p.out_f32 ? f32_ok : (res_ok && !p.res_f32)) && !p.accumulate && p.act == ACT_NONE && p.K >= 4 * PK &&
        (p.out_f32 || (p.ldc % 8 == 0 && p.N % 8 == 0)) && !((uintptr_t)p.C & 15)) {
        sched_prepare(p);
        if (layout == 0) hipLaunchKernelGGL((gemm256p_kernel<false, false, 0, true>), dim3(tiles), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
        else if (layout == 1) hipLaunchKernelGGL((gemm256p_kernel<false, true, 0, true>), dim3(tiles), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
        else hipLaunchKernelGGL((gemm256p_kernel<true, true, 0, true>), dim3(tiles), dim3(512), CONT_LDS_BYTES, stream, p, (const bf16_t*)zero16);
        vlr_prof_end(pi, stream);
        return true;
    }
    if (layout == 0) hipLaunchKernelGGL((gemm256p_kernel<false, false>), dim3(tiles), dim3(512), TILE_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    else if (layout == 1) hipLaunchKernelGGL((gemm256p_kernel<false, true>), dim3(tiles), dim3(512), TILE_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    else hipLaunchKernelGGL((gemm256p_kernel<true, true>), dim3(tiles), dim3(512), TILE_LDS_BYTES, stream, p, (const bf16_t*)zero16);
    vlr_prof_end(pi, stream);
    return true;
}
