// Shared definitions of the bf16 GEMM kernels (gemm.hip: 128x128 tile, gemm256p.hip: 256x256 tile).
#pragma once
#include "common.h"

enum { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU = 2 };

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const bf16_t* bias;      // [N] or null
    const bf16_t* residual;  // [M,N] ld = ldr, or null
    int M, N, K;
    int lda, ldb, ldc, ldr;
    int act;
    int accumulate;  // C += result (C read in its own dtype)
    int out_f32;
    int flags;       // tuning experiments (VLR_GEMM_FLAGS), 0 in production
    float alpha;     // v = act(alpha * acc + bias) + residual (+ C)
    int splitk;      // > 1: blockIdx.y = K-slice z of kchunk elements, raw alpha*acc -> part[z][M][N] fp32 (128x128 kernel only)
    int kchunk;
    float* part;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_QUICK_GELU) return v / (1.f + __expf(-1.702f * v));
    if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}


// 256x256 tile, eight-phase schedule (gemm256p.hip); returns false when the problem does not qualify (fewer than 192 tiles,
// unaligned operands, VLR_GEMM_8PHASE=0): the caller falls back to the 128x128 kernel
bool vlr_gemm256p_try_launch(int layout, const GemmParams& p, hipStream_t stream);
