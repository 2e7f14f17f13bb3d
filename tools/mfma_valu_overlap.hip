// Do the MFMAs of one wave and the VALU work of ANOTHER wave of the same SIMD run side by side on gfx950?  (Round 6: the attention forward's
// counters show matrix-pipe time and VALU time adding up - profiles/r06_attention_pmc.txt - and an 8-wave kernel that forces the two waves of
// a SIMD into opposite MFMA / softmax phases is slower than two unsynchronised workgroups.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
// One 512-thread workgroup per CU = two waves per SIMD (waves w and w + 4 share SIMD w & 3).  Per iteration an "M" wave issues 16
// v_mfma_f32_32x32x16_bf16 on four independent accumulators (512 matrix-pipe cycles), a "V" wave NV v_fma_f32 / v_exp_f32 on independent
// registers (4 issue cycles each; exp: quarter rate), an "I" wave both, interleaved by hand (FILL VALU instructions behind every MFMA).
// Modes: MM (both waves M), VV, MV (waves 0-3 M, waves 4-7 V), II (both interleaved), M- / V- (one wave per SIMD, the other exits).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int FILL, bool EXP>
__device__ __forceinline__ void valu_block(float (&x)[8], float c) {      // FILL independent VALU instructions
#pragma unroll
    for (int i = 0; i < FILL; ++i) {
        if (EXP && (i & 3) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i & 7]) : "v"(c));
    }
}

// role: 0 M, 1 V, 2 I (interleaved), 3 exit
template <int FILL, bool EXP>
__global__ __launch_bounds__(512) void overlap_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters, int role_lo, int role_hi) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_lo : role_hi;
    if (role == 3) return;
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, src[(t * 4 + i) & 65535]);
    for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(bf16x8, src[(t * 4 + 2 + i) & 65535]);
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
    const float c = 0.999f;
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) valu_block<FILL, EXP>(x, c);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        valu_block<FILL, EXP>(x, c);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) out[t] = s;
}

template <int FILL, bool EXP>
static void run(const char* name, const uint4* d, float* o, int grid, int iters, int lo, int hi, double clk_ghz) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((overlap_kernel<FILL, EXP>), dim3(grid), dim3(512), 0, 0, d, o, iters / 10, lo, hi);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((overlap_kernel<FILL, EXP>), dim3(grid), dim3(512), 0, 0, d, o, iters, lo, hi);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s FILL %2d %s  %8.3f ms   %7.1f ns per iteration (16 MFMAs = 512 matrix-pipe cycles and / or %d VALU per wave)\n", name, FILL,
           EXP ? "fma+exp" : "fma    ", ms, ms * 1e6 / iters, 16 * FILL);
    (void)clk_ghz;
}

int main() {
    const int n = 65536;
    uint4* h = (uint4*)malloc(n * sizeof(uint4));
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned v[4];
        for (int k = 0; k < 4; ++k) {
            unsigned lo = (rand() & 0x8000) | ((119 + rand() % 8) << 7) | (rand() & 0x7f), hi = (rand() & 0x8000) | ((119 + rand() % 8) << 7) | (rand() & 0x7f);
            v[k] = lo | (hi << 16);
        }
        h[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    uint4* d;
    float* o;
    hipMalloc(&d, n * sizeof(uint4));
    hipMalloc(&o, 1 << 24);
    hipMemcpy(d, h, n * sizeof(uint4), hipMemcpyHostToDevice);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount, iters = 20000;
#define SET(FILL_, EXP_)                                                                          \
    run<FILL_, EXP_>("M-  one MFMA wave per SIMD", d, o, grid, iters, 0, 3, 0);                   \
    run<FILL_, EXP_>("V-  one VALU wave per SIMD", d, o, grid, iters, 1, 3, 0);                   \
    run<FILL_, EXP_>("MM  two MFMA waves per SIMD", d, o, grid, iters, 0, 0, 0);                  \
    run<FILL_, EXP_>("VV  two VALU waves per SIMD", d, o, grid, iters, 1, 1, 0);                  \
    run<FILL_, EXP_>("MV  an MFMA wave and a VALU wave per SIMD", d, o, grid, iters, 0, 1, 0);    \
    run<FILL_, EXP_>("I-  one wave, VALU behind every MFMA", d, o, grid, iters, 2, 3, 0);         \
    run<FILL_, EXP_>("II  two such waves per SIMD", d, o, grid, iters, 2, 2, 0);                  \
    printf("\n");
    SET(4, false)
    SET(7, false)
    SET(8, true)
    return 0;
}
