// what v_permlane32_swap returns through the builtin when both operands are the SAME value (blocks 0-2: hipcc yields two equal results)
// and with distinct operands (block 3):  hipcc --offload-arch=gfx950 -O3 -o probe tools/probe_permlane32_swap.hip && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float x = (float)lane;                       // lane value
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out[lane] = __builtin_bit_cast(float, r[0]);
    out[64 + lane] = __builtin_bit_cast(float, r[1]);
    out[128 + lane] = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
    // inline asm (the form attn_fwd3.h uses) - expected: every lane holds max(lane & 31, (lane & 31) + 32) = (lane & 31) + 32
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    out[192 + lane] = fmaxf(a, b);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) { printf("blk %d: lanes 0,1,31,32,33,63 -> %g %g %g %g %g %g\n", b, h[b*64], h[b*64+1], h[b*64+31], h[b*64+32], h[b*64+33], h[b*64+63]); }
    return 0;
}
