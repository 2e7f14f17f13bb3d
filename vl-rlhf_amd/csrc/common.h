// Shared device/host helpers for the vlr HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint16_t bf16_t;  // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define VLR_OK 0
#define VLR_ERR_ARG 1
#define VLR_ERR_HIP 2

void vlr_set_error(const char* fmt, ...);
int vlr_check_launch(const char* what);
// kernel ids for the in-library profiler (api.cpp)
enum { VLR_K_GEMM_NT = 0, VLR_K_GEMM_NN = 1, VLR_K_GEMM_TN = 2, VLR_K_ATTN_FWD = 3, VLR_K_ATTN_BWD = 4, VLR_K_GEMM256P = 5, VLR_K_COUNT = 6 };
extern "C" int vlr_compute_cus(void);      // CUs the persistent kernels may fill (api.cpp: device CUs - vlr_set_comm_cus, whole XCD octets)
int vlr_prof_begin(int kernel, double work, hipStream_t st);
void vlr_prof_end(int idx, hipStream_t st);

// ---- internal (not part of the C ABI): the ragged last tile rows of a decoder GEMM ("peel": 504 of 12792 rows on the 128x128 kernel, split
// along K, + its reduction) leave most CUs idle for 50 - 125 us, and the kernel that follows on the stream is an HBM-bound RMSNorm pass over
// the same rows.  A caller that can consume the rows in two parts hands the next vlr_gemm_bf16* call a VlrGemmTail: the peeled rows are
// then launched on `side` (after the main part), `done` is recorded behind them and M1 = rows of the main part; the caller runs its
// consumer on rows [0, M1) at once, waits for `done`, and finishes rows [M1, M).  used = 0: the call did not peel (consume all rows).
struct VlrGemmTail { hipStream_t side; hipEvent_t fork, done; int M1; int used; };
void vlr_internal_set_gemm_tail(VlrGemmTail* t);      // applies to the NEXT vlr_gemm_bf16 / _f32res call of this thread only
int vlr_internal_rmsnorm_bwd_split(const void* dy, const void* x, int x_f32, const void* w, const float* rstd, const void* dres, void* dx,
                                   void* dw, int dw_accumulate, void* workspace, int M, int H, int M1, hipEvent_t tail_done, hipStream_t st);

#define VLR_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            vlr_set_error(__VA_ARGS__);        \
            return VLR_ERR_ARG;                \
        }                                      \
    } while (0)

// counter-based dropout (elementwise.hip, the GEMM epilogue of gemm256p.hip): splitmix64 of (key ^ counter); a group of 8
// consecutive elements g takes its eight 16-bit lanes from mix64(key ^ 2g) (elements 0-3) and mix64(key ^ (2g + 1)) (4-7); an
// element is kept when its lane >= thr = round(p * 65536).  oracle/llava_dpo_oracle.py:dropout_mask restates it.
__host__ __device__ inline uint64_t vlr_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint32_t vlr_dropout_thr(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ float bf16lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// round-to-nearest-even pack of two floats (lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16(f, 0.f) & 0xffffu); }

// sigmoid with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (v_div_scale / v_rcp / 4 v_fma / v_div_fmas /
// v_div_fixup: ~10 VALU instructions per element - a third of the SwiGLU epilogues' arithmetic, which runs with the matrix pipe idle);
// the SAME function in the fused GEMM epilogues (gemm256p.hip) and in the elementwise kernels, so every row sees the same arithmetic
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ void unpack8(const u32x4& w, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16lo(w[i]);
        f[2 * i + 1] = bf16hi(w[i]);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
    return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// block-wide sum for blockDim.x multiple of 64 (<= 1024); `red` = 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}
