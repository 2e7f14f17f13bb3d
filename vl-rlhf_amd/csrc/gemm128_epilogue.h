// Copy-out of the 128x128-tile GEMM kernels (gemm.hip: register-staged kernel, gemm128p.hip: LDS-DMA ring kernel): a wave's 64x64
// fp32 results sit in its LDS stage [64][SS]; bias / activation / dropout-accumulate mask / residual / accumulate are applied in
// fp32 and the global accesses are 16 B per lane along rows.
#pragma once
#include "gemm.h"

template <int SS>
__device__ __forceinline__ void gemm128_copy_out(const GemmParams& p, const float* stage, int gm0, int gn0, int lane) {
    if (!p.out_f32) {
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
        const int cq = (lane & 7) * 8;
        const int gn = gn0 + cq;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
        if (p.bias && gn + 8 <= p.N) unpack8(*reinterpret_cast<const u32x4*>(p.bias + gn), bv);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int gm = gm0 + row;
            if (gm < p.M && gn + 8 <= p.N) {
                float v[8];
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(stage + row * SS + cq);
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(stage + row * SS + cq + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = s0[e];
                    v[4 + e] = s1[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = apply_act(p.alpha * v[e] + bv[e], p.act);
                if (p.fuse == 6) {      // dropout-accumulate: the keep mask of elements (gm, gn .. gn+7) - one hash group of vlr_dropout
                    const long grp = ((long)gm * p.drop_ld + gn) >> 3;
                    if (p.drop_bits) {      // the packed mask drawn by vlr_dropout_bits
                        const uint32_t keep = p.drop_bits[grp];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (!((keep >> e) & 1)) v[e] = 0.f;
                    } else {
                        const uint64_t r0 = vlr_mix64(p.drop_key ^ (uint64_t)(2 * grp)), r1 = vlr_mix64(p.drop_key ^ (uint64_t)(2 * grp + 1));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if ((uint32_t)((r0 >> (16 * e)) & 0xffffu) < p.drop_thr) v[e] = 0.f;
                            if ((uint32_t)((r1 >> (16 * e)) & 0xffffu) < p.drop_thr) v[4 + e] = 0.f;
                        }
                    }
                }
                if (p.residual) {
                    float rv[8];
                    unpack8(*reinterpret_cast<const u32x4*>(p.residual + (size_t)gm * p.ldr + gn), rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                bf16_t* dst = C + (size_t)gm * p.ldc + gn;
                if (p.accumulate) {
                    float ov[8];
                    unpack8(*reinterpret_cast<const u32x4*>(dst), ov);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += ov[e];
                }
                *reinterpret_cast<u32x4*>(dst) = pack8(v);
            }
        }
    } else {
        float* C = reinterpret_cast<float*>(p.C);
        const int cq = (lane & 15) * 4;
        const int gn = gn0 + cq;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && gn + 4 <= p.N) {
            const u32x2 w = *reinterpret_cast<const u32x2*>(p.bias + gn);
            bv[0] = bf16lo(w[0]); bv[1] = bf16hi(w[0]); bv[2] = bf16lo(w[1]); bv[3] = bf16hi(w[1]);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + (lane >> 4);
            const int gm = gm0 + row;
            if (gm < p.M && gn + 4 <= p.N) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * SS + cq);
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
                float* dst = C + (size_t)gm * p.ldc + gn;
                if (p.accumulate) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += o[e];
                }
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
    }
}
