// Sustained MFMA rate of the chip with NO memory traffic: every wave issues v_mfma_f32_16x16x32_bf16 on eight independent
// accumulators from register operands (random or zero bf16 bits), 512 threads per workgroup, 2 workgroups per CU.
// The number this prints is the clock/power-limited ceiling any bf16 GEMM instruction stream can reach on this box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, src[(t * 6 + i) & 65535]);
    for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(bf16x8, src[(t * 6 + 4 + i) & 65535]);
    f32x4 acc[4][2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) out[t] = s;
}

// the same with v_mfma_f32_32x32x16_bf16 (8 passes, twice the FLOP per instruction, half the operand-register reads per FLOP): does the
// power-limited clock differ between the two shapes?
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void mfma_loop32(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, src[(t * 6 + i) & 65535]);
    for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(bf16x8, src[(t * 6 + 4 + i) & 65535]);
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.678f) out[t] = s;
}

int main() {
    const int n = 65536;
    uint4* h = (uint4*)malloc(n * sizeof(uint4));
    uint4* d;
    float* o;
    hipMalloc(&d, n * sizeof(uint4));
    hipMalloc(&o, 1 << 22);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * 2, iters = 20000;
    for (int mode = 0; mode < 4; ++mode) {
        const int shape32 = mode >> 1;
        srand(1);
        for (int i = 0; i < n; ++i) {
            unsigned v[4];
            for (int k = 0; k < 4; ++k) {      // two bf16 in [-1, 1) per word (random sign, exponent 2^-8..2^-1, random mantissa)
                unsigned lo = (mode & 1) ? ((rand() & 0x8000) | ((119 + rand() % 8) << 7) | (rand() & 0x7f)) : 0;
                unsigned hi = (mode & 1) ? ((rand() & 0x8000) | ((119 + rand() % 8) << 7) | (rand() & 0x7f)) : 0;
                v[k] = lo | (hi << 16);
            }
            h[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
        hipMemcpy(d, h, n * sizeof(uint4), hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        // (32x32x16: 16 instructions of 32768 FLOP per iteration = the FLOP of 32 instructions of 16x16x32)
        if (shape32) hipLaunchKernelGGL(mfma_loop32, dim3(grid), dim3(512), 0, 0, d, o, 2000);
        else hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(512), 0, 0, d, o, 2000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (shape32) hipLaunchKernelGGL(mfma_loop32, dim3(grid), dim3(512), 0, 0, d, o, iters);
        else hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(512), 0, 0, d, o, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 8 * iters * 32.0 * 2.0 * 16 * 16 * 32;     // waves x MFMAs x FLOP per MFMA
        const double tf = flop / (ms * 1e-3) / 1e12;
        printf("mfma-only loop %s, %s operands: %.1f TF/s  (%.1f ms; = %.2f GHz x %d CUs x 4096 FLOP/clk)\n", shape32 ? "32x32x16" : "16x16x32", (mode & 1) ? "random bf16" : "zero", tf, ms,
               tf * 1e12 / (p.multiProcessorCount * 4096.0) / 1e9, p.multiProcessorCount);
    }
    return 0;
}
