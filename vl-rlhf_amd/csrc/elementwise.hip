// HBM-bound row / elementwise kernels of the DPO step (gfx950).  All bf16 tensors are moved 16 B per lane
// (8 elements), arithmetic is fp32, reductions are wave shuffles + one LDS hop.  Algorithmic bytes per element are
// stated per kernel in DESIGN.md; none of these re-read a tensor.
#include "common.h"

// ------------------------------------------------------------------------------------------------------------
// RMSNorm forward: y = w * x * rsqrt(mean(x^2) + eps); one workgroup per row.  (LlamaRMSNorm, fp32 internal)
// ------------------------------------------------------------------------------------------------------------
// eight consecutive elements of a bf16 or an fp32 row (the residual stream is fp32 when vlr_llama_cfg::resid_f32 is set)
__device__ __forceinline__ void load8(const bf16_t* p, float* v) { unpack8(*reinterpret_cast<const u32x4*>(p), v); }
__device__ __forceinline__ void load8(const float* p, float* v) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
// REG (H <= 8192): the row stays in registers between the sum of squares and the scaling - x is read ONCE (the two-pass form reads it a
// second time out of the L2: the same values, the same order of additions, a longer dependent chain per workgroup)
template <typename XT, bool REG>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const XT* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out, int H,
                                                          float eps) {
    __shared__ float red[16];
    const size_t row = blockIdx.x;
    const XT* xr = x + row * H;
    float ss = 0.f;
    if constexpr (REG) {
        constexpr int MAXC = 4;
        float v[MAXC][8];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x * 8 + i * 2048;
            if (c < H) {
                load8(xr + c, v[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
            }
        }
        ss = block_sum(ss, red);
        const float rstd = rsqrtf(ss / (float)H + eps);
        if (threadIdx.x == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x * 8 + i * 2048;
            if (c < H) {
                float g[8], o[8];
                unpack8(*reinterpret_cast<const u32x4*>(w + c), g);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = g[e] * (v[i][e] * rstd);
                *reinterpret_cast<u32x4*>(y + row * H + c) = pack8(o);
            }
        }
        return;
    }
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        float v[8];
        load8(xr + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
    }
    ss = block_sum(ss, red);
    const float rstd = rsqrtf(ss / (float)H + eps);
    if (threadIdx.x == 0 && rstd_out) rstd_out[row] = rstd;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        float v[8], g[8];
        load8(xr + c, v);
        unpack8(*reinterpret_cast<const u32x4*>(w + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g[e] * (v[e] * rstd);
        *reinterpret_cast<u32x4*>(y + row * H + c) = pack8(v);
    }
}

// RMSNorm backward.  dx = rstd * (dy*w - xhat * mean(dy*w*xhat)) (+ dres), dw partial[j] += dy*xhat.
// Workgroup b walks rows b, b+G, ...; its dw partial goes to dw_part[b][H] (reduced by reduce_partials_kernel).
// EARLY: the residual-gradient addend of a row is loaded with dy / x, in front of the block reduction, instead of behind it (one exposed
// memory latency per row less; the same values in the same operations)
// MAXC: 8-column chunks per thread the register arrays are sized for (H <= 2048 MAXC): with 4 for every H the 7B width (2 used) cost 150
// registers = 3 waves per SIMD; MAXC = 2 leaves room for more rows in flight per CU (the same instructions on the same values)
template <typename XT, bool EARLY, int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const XT* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                          float* __restrict__ dw_part, int M, int H, int row0, int part0) {
    // rows [row0, M); the workgroup's dw partial goes to row part0 + blockIdx.x of dw_part (two launches over two row ranges share one reduction)
    __shared__ float red[16];
    float dwacc[MAXC][8];      // H <= 256*8*MAXC
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[i][e] = 0.f;
    for (int row = row0 + blockIdx.x; row < M; row += gridDim.x) {
        const size_t off = (size_t)row * H;
        const float rs = rstd[row];
        float dot = 0.f;
        float gx[MAXC][8], xh[MAXC][8];
        u32x4 rraw[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = (i * 256 + threadIdx.x) * 8;
            if (c < H) {
                float a[8], b[8], g[8];
                if constexpr (EARLY) { if (dres) rraw[i] = *reinterpret_cast<const u32x4*>(dres + off + c); }
                unpack8(*reinterpret_cast<const u32x4*>(dy + off + c), a);
                load8(x + off + c, b);
                unpack8(*reinterpret_cast<const u32x4*>(w + c), g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[i][e] = b[e] * rs;
                    gx[i][e] = a[e] * g[e];
                    dot += gx[i][e] * xh[i][e];
                    dwacc[i][e] += a[e] * xh[i][e];
                }
            }
        }
        dot = block_sum(dot, red) / (float)H;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = (i * 256 + threadIdx.x) * 8;
            if (c < H) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (gx[i][e] - xh[i][e] * dot);
                if (dres) {
                    float r[8];
                    if constexpr (EARLY) unpack8(rraw[i], r);
                    else unpack8(*reinterpret_cast<const u32x4*>(dres + off + c), r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += r[e];
                }
                *reinterpret_cast<u32x4*>(dx + off + c) = pack8(o);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < H) {
            float* d = dw_part + (size_t)(part0 + blockIdx.x) * H + c;
            *reinterpret_cast<f32x4*>(d) = f32x4{dwacc[i][0], dwacc[i][1], dwacc[i][2], dwacc[i][3]};
            *reinterpret_cast<f32x4*>(d + 4) = f32x4{dwacc[i][4], dwacc[i][5], dwacc[i][6], dwacc[i][7]};
        }
    }
}

// out[c] (bf16, optional +=) = sum_p part[p][c]
__global__ void reduce_partials_kernel(const float* __restrict__ part, int P, int C, bf16_t* __restrict__ out,
                                       int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += part[(size_t)p * C + c];
    if (accumulate) s += bf16_to_f32(out[c]);
    out[c] = f32_to_bf16(s);
}

// stage 1 of a two-stage reduction: out[y][c] = sum of part[p][c] for p = y, y + gridDim.y, ... in that order, SIXTEEN rows requested before the
// first add: with one load in flight per thread the 16 MiB of RMSNorm's 1024 partial rows moved at 0.9 TB/s (19 us per launch, 65 per step)
__global__ __launch_bounds__(256) void reduce_partials_stage16_kernel(const float* __restrict__ part, int P, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    const int G = gridDim.y;
    int p = blockIdx.y;
    for (; p + 15 * G < P; p += 16 * G) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(p + u * G) * C + c];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; p < P; p += G) s += part[(size_t)p * C + c];
    out[(size_t)blockIdx.y * C + c] = s;
}

__global__ void reduce_partials_f32_kernel(const float* __restrict__ part, int P, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += part[(size_t)p * C + c];
    out[c] = s;
}

// the same partials, 8 columns per thread (16-byte loads) and four rows requested before the first add - same rows in the same order per
// column.  The 4-byte, one-load-in-flight form below ran at 1.2 TB/s (Qwen-VL: 0.2 ms per biased c_attn gradient x 32 layers, 1 ms for the
// lm-head's column sums).  C % 8 == 0, ld % 8 == 0, X 16-byte aligned.
__global__ __launch_bounds__(64) void colsum_partial8_kernel(const bf16_t* __restrict__ X, int R, int C, int ld, float* __restrict__ part) {
    const int c = (blockIdx.x * 64 + threadIdx.x) * 8;
    if (c >= C) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int G = gridDim.y;
    int r = blockIdx.y;
    for (; r + 3 * G < R; r += 4 * G) {
        u32x4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const u32x4*>(X + (size_t)(r + u * G) * ld + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[8];
            unpack8(w[u], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
    }
    for (; r < R; r += G) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(X + (size_t)r * ld + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
    float* d = part + (size_t)blockIdx.y * C + c;
    *reinterpret_cast<f32x4*>(d) = f32x4{s[0], s[1], s[2], s[3]};
    *reinterpret_cast<f32x4*>(d + 4) = f32x4{s[4], s[5], s[6], s[7]};
}
// column-sum partials of a bf16 matrix X[R][C] (ld): workgroup (bx, by) sums rows by, by+Gy, ... of 512 columns
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ X, int R, int C, int ld,
                                                             float* __restrict__ part) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    for (int r = blockIdx.y; r < R; r += gridDim.y) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(X + (size_t)r * ld + c);
        s0 += bf16lo(w);
        s1 += bf16hi(w);
    }
    part[(size_t)blockIdx.y * C + c] = s0;
    part[(size_t)blockIdx.y * C + c + 1] = s1;
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm backward (Qwen-VL resampler: ln_q / ln_kv / ln_post are in front of or behind trainable weights).  mean / rstd are
// recomputed from x: dx = rstd * (g - mean(g) - xh * mean(g * xh)), g = dy * w, xh = (x - mean) * rstd;
// dw = sum_rows dy * xh, db = sum_rows dy -> per-block partials [G][D] | [G][D], reduced in a fixed order
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ w, float eps, bf16_t* __restrict__ dx,
                                                            float* __restrict__ part, int M, int D) {
    __shared__ float red[16];
    constexpr int MAXC = 2;  // D <= 4096
    float dwacc[MAXC][8], dbacc[MAXC][8];
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[i][e] = dbacc[i][e] = 0.f;
    for (int row = blockIdx.x; row < M; row += gridDim.x) {
        const size_t off = (size_t)row * D;
        float xv[MAXC][8], gv[MAXC][8], dv[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = (i * 256 + threadIdx.x) * 8;
            if (c < D) {
                unpack8(*reinterpret_cast<const u32x4*>(x + off + c), xv[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += xv[i][e];
            }
        }
        const float mean = block_sum(s, red) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = (i * 256 + threadIdx.x) * 8;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xv[i][e] -= mean;
                    q += xv[i][e] * xv[i][e];
                }
            }
        }
        const float rs = rsqrtf(block_sum(q, red) / (float)D + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = (i * 256 + threadIdx.x) * 8;
            if (c < D) {
                float wv[8];
                unpack8(*reinterpret_cast<const u32x4*>(dy + off + c), dv[i]);
                unpack8(*reinterpret_cast<const u32x4*>(w + c), wv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xv[i][e] *= rs;                       // xh
                    gv[i][e] = dv[i][e] * wv[e];
                    sg += gv[i][e];
                    sgx += gv[i][e] * xv[i][e];
                    dwacc[i][e] += dv[i][e] * xv[i][e];
                    dbacc[i][e] += dv[i][e];
                }
            }
        }
        sg = block_sum(sg, red) / (float)D;
        sgx = block_sum(sgx, red) / (float)D;
        if (dx) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = (i * 256 + threadIdx.x) * 8;
                if (c < D) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = rs * (gv[i][e] - sg - xv[i][e] * sgx);
                    *reinterpret_cast<u32x4*>(dx + off + c) = pack8(o);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < D) {
            float* d = part + (size_t)blockIdx.x * D + c;
            float* b = part + (size_t)(gridDim.x + blockIdx.x) * D + c;
#pragma unroll
            for (int e = 0; e < 8; ++e) { d[e] = dwacc[i][e]; b[e] = dbacc[i][e]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm forward (CLIP ViT; frozen tower, no backward).  Optional fused "embedding assemble": when pe != null the
// input row (b, t) is (t == 0 ? cls : pe[b, t-1]) + pos[t]  (CLIPVisionEmbeddings + pre_layrnorm).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int D,
                                                            float eps, const bf16_t* __restrict__ pe,
                                                            const bf16_t* __restrict__ cls,
                                                            const bf16_t* __restrict__ pos, int T) {
    __shared__ float red[16];
    const size_t row = blockIdx.x;
    constexpr int MAXC = 2;  // D <= 4096
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < D) {
            if (pe) {
                const int bi = (int)(row / T), ti = (int)(row % T);
                float a[8], p[8];
                if (ti == 0) unpack8(*reinterpret_cast<const u32x4*>(cls + c), a);
                else unpack8(*reinterpret_cast<const u32x4*>(pe + ((size_t)bi * (T - 1) + ti - 1) * D + c), a);
                unpack8(*reinterpret_cast<const u32x4*>(pos + (size_t)ti * D + c), p);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = a[e] + p[e];
            } else {
                unpack8(*reinterpret_cast<const u32x4*>(x + row * D + c), v[i]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    const float mean = block_sum(s, red) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(block_sum(q, red) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < D) {
            float g[8], bb[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(w + c), g);
            unpack8(*reinterpret_cast<const u32x4*>(b + c), bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
            *reinterpret_cast<u32x4*>(y + row * D + c) = pack8(o);
        }
    }
}

// raw-embedding variant used by tests / hidden_states parity: x_out = assemble (no norm)
// ------------------------------------------------------------------------------------------------------------
// RoPE (rotate-half), in place on the q and k thirds of the fused qkv buffer [M][3H].  cos/sin come from a fp32 table
// [max_pos][hd/2]; sign = +1 forward, -1 backward (transpose of the rotation).
// ------------------------------------------------------------------------------------------------------------
__global__ void rope_table_kernel(float* __restrict__ cos_t, float* __restrict__ sin_t, int max_pos, int half,
                                  float theta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= max_pos * half) return;
    const int p = i / half, f = i % half;
    const float inv = 1.0f / powf(theta, (float)(2 * f) / (float)(2 * half));
    const float a = (float)p * inv;
    cos_t[i] = cosf(a);
    sin_t[i] = sinf(a);
}

__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, const int* __restrict__ pos,
                                                   const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                   int M, int n_heads, int hd, int ld, float sign, int max_pos) {
    // one thread = 8 consecutive frequencies of one head of q or k; n_heads rotated heads per row (q heads then k heads),
    // hd/16 threads per head
    const int tph = hd / 16;
    const int per_row = n_heads * tph;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid / per_row;
    if (row >= M) return;
    const int r = (int)(gid % per_row);
    const int head = r / tph, f0 = (r % tph) * 8;   // head in [0, 2*nh): q heads then k heads
    const int half = hd / 2;
    int p = pos[row];
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    bf16_t* base = qkv + (size_t)row * ld + (size_t)head * hd + f0;   // k third follows q third: head*hd spans both
    float x1[8], x2[8], c[8], s[8];
    unpack8(*reinterpret_cast<const u32x4*>(base), x1);
    unpack8(*reinterpret_cast<const u32x4*>(base + half), x2);
    const float* cp = cos_t + (size_t)p * half + f0;
    const float* sp = sin_t + (size_t)p * half + f0;
    *reinterpret_cast<f32x4*>(c) = *reinterpret_cast<const f32x4*>(cp);
    *reinterpret_cast<f32x4*>(c + 4) = *reinterpret_cast<const f32x4*>(cp + 4);
    *reinterpret_cast<f32x4*>(s) = *reinterpret_cast<const f32x4*>(sp);
    *reinterpret_cast<f32x4*>(s + 4) = *reinterpret_cast<const f32x4*>(sp + 4);
    float o1[8], o2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float sn = sign * s[e];
        o1[e] = x1[e] * c[e] - x2[e] * sn;
        o2[e] = x2[e] * c[e] + x1[e] * sn;
    }
    *reinterpret_cast<u32x4*>(base) = pack8(o1);
    *reinterpret_cast<u32x4*>(base + half) = pack8(o2);
}

// ------------------------------------------------------------------------------------------------------------
// SwiGLU: gu = [gate | up] fused [M][2I]; act = silu(gate) * up.  Backward overwrites gu with [dgate | dup].
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act,
                                                         long M, int I) {
    const long chunks = (long)M * (I / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long)gridDim.x * 256) {
        const long row = i / (I / 8);
        const int c = (int)(i % (I / 8)) * 8;
        float g[8], u[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(gu + row * 2 * I + c), g);
        unpack8(*reinterpret_cast<const u32x4*>(gu + row * 2 * I + I + c), u);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = g[e] * fast_sigmoid(g[e]) * u[e];
        *reinterpret_cast<u32x4*>(act + row * I + c) = pack8(o);
    }
}
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(bf16_t* __restrict__ gu, const bf16_t* __restrict__ dact,
                                                         long M, int I) {
    const long chunks = (long)M * (I / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long)gridDim.x * 256) {
        const long row = i / (I / 8);
        const int c = (int)(i % (I / 8)) * 8;
        float g[8], u[8], d[8], dg[8], du[8];
        unpack8(*reinterpret_cast<const u32x4*>(gu + row * 2 * I + c), g);
        unpack8(*reinterpret_cast<const u32x4*>(gu + row * 2 * I + I + c), u);
        unpack8(*reinterpret_cast<const u32x4*>(dact + row * I + c), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = fast_sigmoid(g[e]);
            du[e] = d[e] * g[e] * sg;
            dg[e] = d[e] * u[e] * sg * (1.f + g[e] * (1.f - sg));
        }
        *reinterpret_cast<u32x4*>(gu + row * 2 * I + c) = pack8(dg);
        *reinterpret_cast<u32x4*>(gu + row * 2 * I + I + c) = pack8(du);
    }
}

// GELU (erf) forward / backward for the projector (z kept for backward)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ z, bf16_t* __restrict__ h, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(z + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f));
        *reinterpret_cast<u32x4*>(h + i * 8) = pack8(v);
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ dh,
                                                       bf16_t* __restrict__ dz, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8], d[8];
        unpack8(*reinterpret_cast<const u32x4*>(z + i * 8), v);
        unpack8(*reinterpret_cast<const u32x4*>(dh + i * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float cdf = 0.5f * (1.f + erff(v[e] * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * v[e] * v[e]);
            d[e] *= cdf + v[e] * pdf;
        }
        *reinterpret_cast<u32x4*>(dz + i * 8) = pack8(d);
    }
}

// ------------------------------------------------------------------------------------------------------------
// counter-based dropout (LoRA lora_dropout, peft lora.Linear: result += lora_B(lora_A(dropout(x))) * scaling).
// Group g of 8 consecutive elements draws two 64-bit words r_j = mix64(key ^ (2g+j)), key = mix64(seed); element e keeps
// iff its 16-bit lane (r_{e/4} >> 16*(e%4)) & 0xffff >= thr, thr = round(p * 65536).  Stateless, so the backward
// regenerates the mask from (seed, index) instead of storing it.  oracle/llava_dpo_oracle.py:dropout_mask restates it.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dropout_keep8(uint64_t key, long g, uint32_t thr) {
    const uint64_t r0 = vlr_mix64(key ^ (uint64_t)(2 * g)), r1 = vlr_mix64(key ^ (uint64_t)(2 * g + 1));
    uint32_t keep = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        keep |= (uint32_t)(((r0 >> (16 * e)) & 0xffffu) >= thr) << e;
        keep |= (uint32_t)(((r1 >> (16 * e)) & 0xffffu) >= thr) << (4 + e);
    }
    return keep;
}
// out = mask * x * scale ; or, with ADD, out += mask * x * scale
template <bool ADD>
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long n8,
                                                      uint64_t key, uint32_t thr, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), v);
        const uint32_t keep = dropout_keep8(key, i, thr);
        if (ADD) {
            float o[8];
            unpack8(*reinterpret_cast<const u32x4*>(out + i * 8), o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = o[e] + ((keep >> e) & 1 ? v[e] * scale : 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (keep >> e) & 1 ? v[e] * scale : 0.f;
        }
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8(v);
    }
}
// the same keep mask PACKED: bit e of byte g = element 8 g + e (one 32-bit store per four hash groups) - the form the LoRA adapter GEMMs read
// (GemmParams::mask_bits): the hash costs ~300-450 cycles per wave and group pair and the forward, the dA and the dx kernels of a target
// all need the same mask, so it is drawn ONCE per layer pass
__global__ __launch_bounds__(256) void dropout_bits_kernel(uint32_t* __restrict__ bits, long n32, uint64_t key, uint32_t thr) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n32; i += (long)gridDim.x * 256) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w |= dropout_keep8(key, 4 * i + j, thr) << (8 * j);
        bits[i] = w;
    }
}
// Both packed forms in one draw: row-major (bit e of byte (row * cols + col) / 8 = column col + e) and K-TILE-BLOCKED TRANSPOSED
// (byte ((row / 64) * cols + col) * 8 + (row % 64) / 8, bit e = row + e): the form the TN adapter product dA = v^T (mask . x) reads - x is
// its K-strided operand, a fragment is 8 consecutive ROWS of one column, i.e. one byte of this layout, and the 128 columns x 8 bytes
// of a K tile are 1 KiB contiguous (one LDS-DMA instruction per wave).  A workgroup owns 64 rows x 256 columns; a thread hashes an
// 8 x 8 block (row group j = t / 32, column block t % 32), stores its 8 row-major bytes (32 threads = 32 consecutive bytes of a row),
// transposes the block in a register (three masked swaps) and leaves its 8 column bytes in LDS, from where thread t writes the 8 bytes
// of column t as one 64-bit store.
__global__ __launch_bounds__(256) void dropout_bits2_kernel(unsigned char* __restrict__ bits, unsigned char* __restrict__ bits_kt, int rows, int cols,
                                                            uint64_t key, uint32_t thr) {
    __shared__ unsigned char colb[256][8];
    __shared__ __attribute__((aligned(16))) unsigned char rowb[64][32];       // the block's 64 rows x 256 columns, row-major bits
    const int t = threadIdx.x, j = t >> 5, cb = t & 31;
    const int cblocks = (cols + 255) >> 8;
    const int kt = blockIdx.x / cblocks, c0 = (blockIdx.x % cblocks) * 256;
    const int col = c0 + cb * 8;
    uint64_t x = 0;                                     // 8 rows x 8 columns: byte rr = row, bit e = column
    if (col < cols) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = kt * 64 + j * 8 + rr;
            if (row < rows) {
                const long g = ((long)row * cols + col) >> 3;
                const uint32_t keep = dropout_keep8(key, g, thr);
                rowb[j * 8 + rr][cb] = (unsigned char)keep;
                x |= (uint64_t)keep << (8 * rr);
            }
        }
    }
    // 8 x 8 bit-matrix transpose (bit 8 i + j <-> bit 8 j + i): byte e = column, bit rr = row
    uint64_t w = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;
    x = x ^ w ^ (w << 7);
    w = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull;
    x = x ^ w ^ (w << 14);
    w = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull;
    x = x ^ w ^ (w << 28);
#pragma unroll
    for (int e = 0; e < 8; ++e) colb[cb * 8 + e][j] = (unsigned char)(x >> (8 * e));
    __syncthreads();
    if (bits_kt && c0 + t < cols)
        *reinterpret_cast<uint64_t*>(bits_kt + ((long)kt * cols + c0 + t) * 8) = *reinterpret_cast<const uint64_t*>(&colb[t][0]);
    // row-major bits: 16 bytes per thread (a row of the block = 32 bytes = two stores) instead of one byte per thread and row -
    // the byte stores made this draw run at 300 GB/s (26 x its HBM floor, profiles/r03_lora_gemm_microbench_llava.txt)
    if (bits && t < 128) {
        const int r = t >> 1, h = t & 1;
        const int row = kt * 64 + r, cbyte = (c0 >> 3) + h * 16;       // byte column inside the row
        if (row < rows && c0 + h * 128 < cols) {
            unsigned char* dst = bits + (((long)row * cols) >> 3) + cbyte;
            const int nb = min(16, ((cols + 7) >> 3) - cbyte);
            if (nb == 16 && (((uintptr_t)dst) & 15) == 0) *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(&rowb[r][h * 16]);
            else for (int i = 0; i < nb; ++i) dst[i] = rowb[r][h * 16 + i];
        }
    }
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, long n8, uint64_t key, uint32_t thr) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const uint32_t keep = dropout_keep8(key, i, thr);
        uint64_t w = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) w |= (uint64_t)((keep >> e) & 1) << (8 * e);
        *reinterpret_cast<uint64_t*>(mask + i * 8) = w;
    }
}

// ------------------------------------------------------------------------------------------------------------
// im2col for the CLIP patch embedding: pixel_values fp32 [n][3][S][S] -> patches bf16 [n*g*g][Kp], column order
// (c, py, px) = Conv2d weight.reshape(D, 3*P*P); columns >= 3*P*P are zero padding up to Kp (multiple of 8).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ px, bf16_t* __restrict__ out, int n,
                                                     int S, int P, int Kp) {
    const int g = S / P;
    const long total = (long)n * g * g * Kp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        const long pr = i / Kp;
        float v = 0.f;
        if (k < 3 * P * P) {
            const int c = k / (P * P), py = (k / P) % P, pxx = k % P;
            const int gi = (int)(pr % (g * g));
            const long img = pr / (g * g);
            const int y = (gi / g) * P + py, x = (gi % g) * P + pxx;
            v = px[((img * 3 + c) * S + y) * S + x];
        }
        out[i] = f32_to_bf16(v);
    }
}

// generic helpers -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ rows,
                                                          bf16_t* __restrict__ dst, int R, int H) {
    const int r = blockIdx.x;
    const size_t s = (size_t)rows[r] * H;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8)
        *reinterpret_cast<u32x4*>(dst + (size_t)r * H + c) = *reinterpret_cast<const u32x4*>(src + s + c);
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ rows,
                                                           bf16_t* __restrict__ dst, int R, int H) {
    const int r = blockIdx.x;
    const size_t d = (size_t)rows[r] * H;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8)
        *reinterpret_cast<u32x4*>(dst + d + c) = *reinterpret_cast<const u32x4*>(src + (size_t)r * H + c);
}
// strided variants for sub-matrices (PLoRA of InternLM-XComposer2: the image rows of one column block of a fused activation)
__global__ __launch_bounds__(256) void rows_gather_kernel(const bf16_t* __restrict__ src, int lds, const int* __restrict__ rows,
                                                          bf16_t* __restrict__ dst, int W) {
    const int r = blockIdx.x;
    const size_t s = (size_t)rows[r] * lds;
    for (int c = threadIdx.x * 8; c < W; c += 256 * 8)
        *reinterpret_cast<u32x4*>(dst + (size_t)r * W + c) = *reinterpret_cast<const u32x4*>(src + s + c);
}
// dst[rows[r]][0:W] += src[r][0:W]  (fp32 add, one rounding; the row list has no duplicates: every destination row has one writer)
__global__ __launch_bounds__(256) void rows_add_kernel(const bf16_t* __restrict__ src, const int* __restrict__ rows,
                                                       bf16_t* __restrict__ dst, int ldd, int W) {
    const int r = blockIdx.x;
    bf16_t* d = dst + (size_t)rows[r] * ldd;
    for (int c = threadIdx.x * 8; c < W; c += 256 * 8) {
        float a[8], b[8];
        unpack8(*reinterpret_cast<const u32x4*>(src + (size_t)r * W + c), a);
        unpack8(*reinterpret_cast<const u32x4*>(d + c), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        *reinterpret_cast<u32x4*>(d + c) = pack8(a);
    }
}
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = f32_to_bf16(src[i]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = bf16_to_f32(src[i]);
}
// out[r] = sum_c X[r][c] * v[c]   (used for the logits-mean metric: mean_v(h . W_v) = h . mean_v W_v)
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16_t* __restrict__ X, const float* __restrict__ v,
                                                     float* __restrict__ out, int H) {
    __shared__ float red[16];
    const size_t row = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        float a[8];
        unpack8(*reinterpret_cast<const u32x4*>(X + row * H + c), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += a[e] * v[c + e];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[row] = s;
}

// ============================================================================================================
static inline int grid_for(long n, int per_block, int cap = 256 * 16) {
    long g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

static bool norm_fwd_reg() {      // VLR_NORM_FWD_REG=0: the two-pass kernel (A/B; bit-identical)
    static int on = -1;
    if (on < 0) { const char* e = getenv("VLR_NORM_FWD_REG"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}
extern "C" int vlr_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int H, float eps,
                               hipStream_t st) {
    VLR_REQUIRE(M > 0 && H > 0 && H % 8 == 0, "vlr_rmsnorm_fwd: bad shape M=%d H=%d", M, H);
    if (H <= 8192 && norm_fwd_reg())
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<bf16_t, true>), dim3(M), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, H, eps);
    else
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<bf16_t, false>), dim3(M), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, H, eps);
    return vlr_check_launch("vlr_rmsnorm_fwd");
}
// the same on an fp32 residual stream x [M][H]: y stays bf16 (the A operand of the projection that follows), rounded once
extern "C" int vlr_rmsnorm_fwd_f32(const float* x, const void* w, void* y, float* rstd, int M, int H, float eps, hipStream_t st) {
    VLR_REQUIRE(x && w && y, "vlr_rmsnorm_fwd_f32: null argument");
    VLR_REQUIRE(M > 0 && H > 0 && H % 8 == 0, "vlr_rmsnorm_fwd_f32: bad shape M=%d H=%d", M, H);
    if (H <= 8192 && norm_fwd_reg())
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<float, true>), dim3(M), dim3(256), 0, st, x, (const bf16_t*)w, (bf16_t*)y, rstd, H, eps);
    else
        hipLaunchKernelGGL((rmsnorm_fwd_kernel<float, false>), dim3(M), dim3(256), 0, st, x, (const bf16_t*)w, (bf16_t*)y, rstd, H, eps);
    return vlr_check_launch("vlr_rmsnorm_fwd_f32");
}

// 1024 workgroups (4 per CU) keep enough loads in flight for an HBM-bound pass; their dw partials are reduced in two
// deterministic stages (1024 -> 16 -> 1)
#define VLR_NORM_BWD_BLOCKS 1024
#define VLR_NORM_BWD_STAGE2 16
extern "C" int vlr_rmsnorm_bwd_workspace_bytes(int H) { return (VLR_NORM_BWD_BLOCKS + VLR_NORM_BWD_STAGE2) * H * 4; }

static int rmsnorm_bwd_impl(const void* dy, const void* x, int x_f32, const void* w, const float* rstd, const void* dres,
                            void* dx, void* dw, int dw_accumulate, void* workspace, int M, int H, hipStream_t st, int M1 = 0,
                            hipEvent_t tail_done = nullptr);
extern "C" int vlr_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres,
                               void* dx, void* dw, int dw_accumulate, void* workspace, int M, int H, hipStream_t st) {
    return rmsnorm_bwd_impl(dy, x, 0, w, rstd, dres, dx, dw, dw_accumulate, workspace, M, H, st);
}
// the same with the forward input x kept in fp32 (fp32 residual stream); the gradient stream dy / dres / dx stays bf16
extern "C" int vlr_rmsnorm_bwd_f32(const void* dy, const float* x, const void* w, const float* rstd, const void* dres,
                                   void* dx, void* dw, int dw_accumulate, void* workspace, int M, int H, hipStream_t st) {
    return rmsnorm_bwd_impl(dy, x, 1, w, rstd, dres, dx, dw, dw_accumulate, workspace, M, H, st);
}
// M1 > 0 (vlr_internal_rmsnorm_bwd_split, common.h: VlrGemmTail): rows [0, M1) now, rows [M1, M) once `tail_done` has fired on the
// stream - the producer's peeled rows were still being computed on a side stream - and ONE reduction of the dw partials of both launches
static int rmsnorm_bwd_impl(const void* dy, const void* x, int x_f32, const void* w, const float* rstd, const void* dres,
                            void* dx, void* dw, int dw_accumulate, void* workspace, int M, int H, hipStream_t st, int M1,
                            hipEvent_t tail_done) {
    VLR_REQUIRE(dy && x && w && rstd && dx, "vlr_rmsnorm_bwd: null argument");
    VLR_REQUIRE(M > 0 && H % 8 == 0 && H <= 8192, "vlr_rmsnorm_bwd: bad shape M=%d H=%d (H<=8192)", M, H);
    VLR_REQUIRE(workspace, "vlr_rmsnorm_bwd: workspace of vlr_rmsnorm_bwd_workspace_bytes(H) required");
    VLR_REQUIRE(M1 >= 0 && M1 < M, "vlr_rmsnorm_bwd: split row %d outside [0, %d)", M1, M);
    auto launch = [&](int row0, int rows_end, int G, int part0) {
        static int early = -1;      // VLR_NORM_BWD_EARLY=0: the addend loaded behind the reduction (A/B; bit-identical)
        if (early < 0) { const char* e = getenv("VLR_NORM_BWD_EARLY"); early = (e && e[0] == '0') ? 0 : 1; }
#define NORM_BWD_LAUNCH(XT_, E_)                                                                                                       \
    do {                                                                                                                               \
        if (H <= 4096) hipLaunchKernelGGL((rmsnorm_bwd_kernel<XT_, E_, 2>), dim3(G), dim3(256), 0, st, (const bf16_t*)dy, (const XT_*)x,  \
                                          (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, (float*)workspace, rows_end, H, row0, part0); \
        else hipLaunchKernelGGL((rmsnorm_bwd_kernel<XT_, E_, 4>), dim3(G), dim3(256), 0, st, (const bf16_t*)dy, (const XT_*)x,          \
                                (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, (float*)workspace, rows_end, H, row0, part0); \
    } while (0)
        if (x_f32) { if (early) NORM_BWD_LAUNCH(float, true); else NORM_BWD_LAUNCH(float, false); }
        else { if (early) NORM_BWD_LAUNCH(bf16_t, true); else NORM_BWD_LAUNCH(bf16_t, false); }
#undef NORM_BWD_LAUNCH
    };
    int G = M < VLR_NORM_BWD_BLOCKS ? M : VLR_NORM_BWD_BLOCKS;
    if (M1 > 0) {
        const int GB = (M - M1) < 64 ? (M - M1) : 64;
        const int GA = M1 < VLR_NORM_BWD_BLOCKS - GB ? M1 : VLR_NORM_BWD_BLOCKS - GB;
        launch(0, M1, GA, 0);
        if (tail_done) hipStreamWaitEvent(st, tail_done, 0);
        launch(M1, M, GB, GA);
        G = GA + GB;
    } else {
        launch(0, M, G, 0);
    }
    if (dw) {
        float* part2 = (float*)workspace + (size_t)VLR_NORM_BWD_BLOCKS * H;
        const int S2 = G < VLR_NORM_BWD_STAGE2 ? 1 : VLR_NORM_BWD_STAGE2;
        hipLaunchKernelGGL(reduce_partials_stage16_kernel, dim3((H + 255) / 256, S2), dim3(256), 0, st, (const float*)workspace, G, H, part2);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((H + 255) / 256), dim3(256), 0, st, (const float*)part2, S2, H,
                           (bf16_t*)dw, dw_accumulate);
    }
    return vlr_check_launch("vlr_rmsnorm_bwd");
}
int vlr_internal_rmsnorm_bwd_split(const void* dy, const void* x, int x_f32, const void* w, const float* rstd, const void* dres, void* dx,
                                   void* dw, int dw_accumulate, void* workspace, int M, int H, int M1, hipEvent_t tail_done, hipStream_t st) {
    return rmsnorm_bwd_impl(dy, x, x_f32, w, rstd, dres, dx, dw, dw_accumulate, workspace, M, H, st, M1, tail_done);
}

#define VLR_COLSUM_ROWS 64
extern "C" int vlr_colsum_workspace_bytes(int C) { return VLR_COLSUM_ROWS * C * 4; }
extern "C" int vlr_colsum(const void* X, int R, int C, int ld, void* out, int accumulate, void* workspace,
                          hipStream_t st) {
    VLR_REQUIRE(R > 0 && C > 0 && C % 2 == 0 && ld % 2 == 0 && workspace, "vlr_colsum: bad args R=%d C=%d", R, C);
    const int gy = R < VLR_COLSUM_ROWS ? R : VLR_COLSUM_ROWS;
    if (C % 8 == 0 && ld % 8 == 0 && !((uintptr_t)X & 15))
        hipLaunchKernelGGL(colsum_partial8_kernel, dim3((C / 8 + 63) / 64, gy), dim3(64), 0, st, (const bf16_t*)X, R, C, ld, (float*)workspace);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((C / 2 + 255) / 256, gy), dim3(256), 0, st, (const bf16_t*)X, R, C, ld,
                           (float*)workspace);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)workspace, gy, C,
                       (bf16_t*)out, accumulate);
    return vlr_check_launch("vlr_colsum");
}

extern "C" int vlr_colsum_f32(const void* X, int R, int C, int ld, float* out, void* workspace, hipStream_t st) {
    VLR_REQUIRE(R > 0 && C > 0 && C % 2 == 0 && ld % 2 == 0 && workspace, "vlr_colsum_f32: bad args R=%d C=%d", R, C);
    const int gy = R < VLR_COLSUM_ROWS ? R : VLR_COLSUM_ROWS;
    if (C % 8 == 0 && ld % 8 == 0 && !((uintptr_t)X & 15))
        hipLaunchKernelGGL(colsum_partial8_kernel, dim3((C / 8 + 63) / 64, gy), dim3(64), 0, st, (const bf16_t*)X, R, C, ld, (float*)workspace);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((C / 2 + 255) / 256, gy), dim3(256), 0, st, (const bf16_t*)X, R, C, ld,
                           (float*)workspace);
    hipLaunchKernelGGL(reduce_partials_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)workspace, gy, C, out);
    return vlr_check_launch("vlr_colsum_f32");
}

#define VLR_LN_BWD_BLOCKS 256
extern "C" int vlr_layernorm_bwd_workspace_bytes(int D) { return 2 * VLR_LN_BWD_BLOCKS * D * 4; }
// dx (may be NULL) / dw / db (may be NULL together) of y = LayerNorm(x) * w + b; dw, db bf16 [D] (accumulate: +=)
extern "C" int vlr_layernorm_bwd(const void* dy, const void* x, const void* w, float eps, void* dx, void* dw, void* db,
                                 int accumulate, void* workspace, int M, int D, hipStream_t st) {
    VLR_REQUIRE(M > 0 && D % 8 == 0 && D <= 4096 && workspace, "vlr_layernorm_bwd: bad shape M=%d D=%d", M, D);
    const int G = M < VLR_LN_BWD_BLOCKS ? M : VLR_LN_BWD_BLOCKS;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(G), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, eps,
                       (bf16_t*)dx, (float*)workspace, M, D);
    if (dw) hipLaunchKernelGGL(reduce_partials_kernel, dim3((D + 255) / 256), dim3(256), 0, st, (const float*)workspace, G, D, (bf16_t*)dw, accumulate);
    if (db) hipLaunchKernelGGL(reduce_partials_kernel, dim3((D + 255) / 256), dim3(256), 0, st, (const float*)workspace + (size_t)G * D, G, D,
                               (bf16_t*)db, accumulate);
    return vlr_check_launch("vlr_layernorm_bwd");
}

extern "C" int vlr_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int M, int D, float eps,
                                 hipStream_t st) {
    VLR_REQUIRE(M > 0 && D % 8 == 0 && D <= 4096, "vlr_layernorm_fwd: bad shape M=%d D=%d", M, D);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(M), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w,
                       (const bf16_t*)b, (bf16_t*)y, D, eps, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, 1);
    return vlr_check_launch("vlr_layernorm_fwd");
}
extern "C" int vlr_vit_embed_ln(const void* patch_embeds, const void* cls, const void* pos, const void* w, const void* b,
                                void* y, int n_img, int T, int D, float eps, hipStream_t st) {
    VLR_REQUIRE(n_img > 0 && T > 1 && D % 8 == 0 && D <= 4096, "vlr_vit_embed_ln: bad shape");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(n_img * T), dim3(256), 0, st, (const bf16_t*)nullptr, (const bf16_t*)w,
                       (const bf16_t*)b, (bf16_t*)y, D, eps, (const bf16_t*)patch_embeds, (const bf16_t*)cls,
                       (const bf16_t*)pos, T);
    return vlr_check_launch("vlr_vit_embed_ln");
}

extern "C" int vlr_rope_table(float* cos_t, float* sin_t, int max_pos, int head_dim, float theta, hipStream_t st) {
    VLR_REQUIRE(max_pos > 0 && head_dim % 16 == 0, "vlr_rope_table: bad args");
    const int n = max_pos * (head_dim / 2);
    hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cos_t, sin_t, max_pos, head_dim / 2, theta);
    return vlr_check_launch("vlr_rope_table");
}
// rotate the first n_heads heads (head_dim columns each: the q heads followed by the k heads) of every row of qkv [M][ld]
extern "C" int vlr_rope_heads(void* qkv, const int* pos, const float* cos_t, const float* sin_t, int M, int n_heads, int head_dim,
                              int ld, int max_pos, int backward, hipStream_t st) {
    VLR_REQUIRE(M > 0 && n_heads > 0 && head_dim % 16 == 0 && ld % 8 == 0 && n_heads * head_dim <= ld, "vlr_rope: bad shape");
    const long threads = (long)M * n_heads * (head_dim / 16);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (bf16_t*)qkv, pos, cos_t,
                       sin_t, M, n_heads, head_dim, ld, backward ? -1.f : 1.f, max_pos);
    return vlr_check_launch("vlr_rope");
}
extern "C" int vlr_rope(void* qkv, const int* pos, const float* cos_t, const float* sin_t, int M, int H, int head_dim,
                        int ld, int max_pos, int backward, hipStream_t st) {
    VLR_REQUIRE(head_dim > 0 && H % head_dim == 0, "vlr_rope: bad shape");
    return vlr_rope_heads(qkv, pos, cos_t, sin_t, M, 2 * (H / head_dim), head_dim, ld, max_pos, backward, st);
}

extern "C" int vlr_swiglu_fwd(const void* gu, void* act, int M, int I, hipStream_t st) {
    VLR_REQUIRE(M > 0 && I % 8 == 0, "vlr_swiglu_fwd: bad shape");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for((long)M * I / 8, 256)), dim3(256), 0, st, (const bf16_t*)gu,
                       (bf16_t*)act, (long)M, I);
    return vlr_check_launch("vlr_swiglu_fwd");
}
extern "C" int vlr_swiglu_bwd(void* gu_inout, const void* dact, int M, int I, hipStream_t st) {
    VLR_REQUIRE(M > 0 && I % 8 == 0, "vlr_swiglu_bwd: bad shape");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((long)M * I / 8, 256)), dim3(256), 0, st, (bf16_t*)gu_inout,
                       (const bf16_t*)dact, (long)M, I);
    return vlr_check_launch("vlr_swiglu_bwd");
}
extern "C" int vlr_gelu_fwd(const void* z, void* h, long n, hipStream_t st) {
    VLR_REQUIRE(n > 0 && n % 8 == 0, "vlr_gelu_fwd: n %% 8");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)z, (bf16_t*)h, n / 8);
    return vlr_check_launch("vlr_gelu_fwd");
}
extern "C" int vlr_gelu_bwd(const void* z, const void* dh, void* dz, long n, hipStream_t st) {
    VLR_REQUIRE(n > 0 && n % 8 == 0, "vlr_gelu_bwd: n %% 8");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)dh,
                       (bf16_t*)dz, n / 8);
    return vlr_check_launch("vlr_gelu_bwd");
}
extern "C" int vlr_dropout(const void* x, void* out, long n, float p, uint64_t seed, float alpha, int add, hipStream_t st) {
    VLR_REQUIRE(n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f, "vlr_dropout: n %% 8 == 0 and 0 <= p < 1 required (n=%ld p=%g)", n, (double)p);
    const float scale = alpha / (1.f - p);
    if (add)
        hipLaunchKernelGGL(dropout_kernel<true>, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, n / 8,
                           vlr_mix64(seed), vlr_dropout_thr(p), scale);
    else
        hipLaunchKernelGGL(dropout_kernel<false>, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, n / 8,
                           vlr_mix64(seed), vlr_dropout_thr(p), scale);
    return vlr_check_launch("vlr_dropout");
}
extern "C" int vlr_dropout_mask(void* mask_u8, long n, float p, uint64_t seed, hipStream_t st) {
    VLR_REQUIRE(n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f, "vlr_dropout_mask: n %% 8 == 0 and 0 <= p < 1 required");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (uint8_t*)mask_u8, n / 8, vlr_mix64(seed),
                       vlr_dropout_thr(p));
    return vlr_check_launch("vlr_dropout_mask");
}
extern "C" int vlr_dropout_bits(void* bits_u8, long n, float p, uint64_t seed, hipStream_t st) {
    VLR_REQUIRE(bits_u8 && n > 0 && n % 32 == 0 && p >= 0.f && p < 1.f && !((uintptr_t)bits_u8 & 3),
                "vlr_dropout_bits: n %% 32 == 0, 0 <= p < 1 and a 4-byte aligned buffer required (n=%ld p=%g)", n, (double)p);
    hipLaunchKernelGGL(dropout_bits_kernel, dim3(grid_for(n / 32, 256)), dim3(256), 0, st, (uint32_t*)bits_u8, n / 32, vlr_mix64(seed),
                       vlr_dropout_thr(p));
    return vlr_check_launch("vlr_dropout_bits");
}
extern "C" long vlr_dropout_bits_kt_bytes(int rows, int cols) { return (long)((rows + 63) / 64) * cols * 8; }
extern "C" int vlr_dropout_bits2(void* bits_u8, void* bits_kt_u8, int rows, int cols, float p, uint64_t seed, hipStream_t st) {
    VLR_REQUIRE((bits_u8 || bits_kt_u8) && rows > 0 && cols > 0 && cols % 8 == 0 && p >= 0.f && p < 1.f && !((uintptr_t)bits_kt_u8 & 7),
                "vlr_dropout_bits2: cols %% 8 == 0, 0 <= p < 1 and an 8-byte aligned transposed buffer required (rows=%d cols=%d p=%g)", rows, cols, (double)p);
    const long nblk = (long)((rows + 63) / 64) * ((cols + 255) / 256);
    VLR_REQUIRE(nblk < (1L << 31), "vlr_dropout_bits2: too many blocks");
    hipLaunchKernelGGL(dropout_bits2_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (unsigned char*)bits_u8, (unsigned char*)bits_kt_u8, rows,
                       cols, vlr_mix64(seed), vlr_dropout_thr(p));
    return vlr_check_launch("vlr_dropout_bits2");
}
extern "C" int vlr_im2col(const float* pixel_values, void* patches, int n_img, int image_size, int patch, int Kp,
                          hipStream_t st) {
    VLR_REQUIRE(n_img > 0 && image_size % patch == 0 && Kp >= 3 * patch * patch && Kp % 8 == 0, "vlr_im2col: bad args");
    const long total = (long)n_img * (image_size / patch) * (image_size / patch) * Kp;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, pixel_values, (bf16_t*)patches, n_img,
                       image_size, patch, Kp);
    return vlr_check_launch("vlr_im2col");
}
extern "C" int vlr_gather_rows(const void* src, const int* rows, void* dst, int R, int H, hipStream_t st) {
    VLR_REQUIRE(R > 0 && H % 8 == 0, "vlr_gather_rows: bad shape");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(R), dim3(256), 0, st, (const bf16_t*)src, rows, (bf16_t*)dst, R, H);
    return vlr_check_launch("vlr_gather_rows");
}
extern "C" int vlr_scatter_rows(const void* src, const int* rows, void* dst, int R, int H, hipStream_t st) {
    VLR_REQUIRE(R > 0 && H % 8 == 0, "vlr_scatter_rows: bad shape");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(R), dim3(256), 0, st, (const bf16_t*)src, rows, (bf16_t*)dst, R, H);
    return vlr_check_launch("vlr_scatter_rows");
}
extern "C" int vlr_rows_gather(const void* src, int lds, const int* rows, void* dst, int R, int W, hipStream_t st) {
    VLR_REQUIRE(R > 0 && W > 0 && W % 8 == 0 && lds % 8 == 0 && lds >= W, "vlr_rows_gather: bad shape R=%d W=%d lds=%d", R, W, lds);
    hipLaunchKernelGGL(rows_gather_kernel, dim3(R), dim3(256), 0, st, (const bf16_t*)src, lds, rows, (bf16_t*)dst, W);
    return vlr_check_launch("vlr_rows_gather");
}
extern "C" int vlr_rows_add(const void* src, const int* rows, void* dst, int ldd, int R, int W, hipStream_t st) {
    VLR_REQUIRE(R > 0 && W > 0 && W % 8 == 0 && ldd % 8 == 0 && ldd >= W, "vlr_rows_add: bad shape R=%d W=%d ldd=%d", R, W, ldd);
    hipLaunchKernelGGL(rows_add_kernel, dim3(R), dim3(256), 0, st, (const bf16_t*)src, rows, (bf16_t*)dst, ldd, W);
    return vlr_check_launch("vlr_rows_add");
}
extern "C" int vlr_cast_f32_to_bf16(const float* src, void* dst, long n, hipStream_t st) {
    VLR_REQUIRE(n > 0, "vlr_cast_f32_to_bf16: n");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, src, (bf16_t*)dst, n);
    return vlr_check_launch("vlr_cast_f32_to_bf16");
}
extern "C" int vlr_cast_bf16_to_f32(const void* src, float* dst, long n, hipStream_t st) {
    VLR_REQUIRE(n > 0, "vlr_cast_bf16_to_f32: n");
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, (const bf16_t*)src, dst, n);
    return vlr_check_launch("vlr_cast_bf16_to_f32");
}
extern "C" int vlr_rowdot(const void* X, const float* v, float* out, int M, int H, hipStream_t st) {
    VLR_REQUIRE(M > 0 && H % 8 == 0, "vlr_rowdot: bad shape");
    hipLaunchKernelGGL(rowdot_kernel, dim3(M), dim3(256), 0, st, (const bf16_t*)X, v, out, H);
    return vlr_check_launch("vlr_rowdot");
}

// diagnostics (include/vlr.h vlr_comm_probe): what an RCCL ring kernel looks like to the compute kernels - a fixed, small number of
// workgroups that stream a bucket through HBM for a while.  16 B per lane, grid-stride.
__global__ __launch_bounds__(256) void comm_probe_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int vlr_comm_probe(const void* src, void* dst, long n_bytes, int wgs, hipStream_t st) {
    VLR_REQUIRE(src && dst && n_bytes > 0 && n_bytes % 16 == 0 && wgs > 0 && wgs <= 1024, "vlr_comm_probe: bad arguments");
    hipLaunchKernelGGL(comm_probe_kernel, dim3(wgs), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, n_bytes / 16);
    return vlr_check_launch("vlr_comm_probe");
}

// rows of x [M][ld] (first `cols` columns) whose mask byte is 0 are zeroed - the row restriction of InternLM-XComposer2's PLoRA
// (`res[im_mask] += Plora_B(Plora_A(x[im_mask]))`, reference models/InternLMXC2/build_mlp.py:194-202) applied to the skinny adapter
// tensors u = drop(x) A^T (forward) and v = dy B (backward): with their text rows zero the dense adapter kernels compute exactly it
__global__ __launch_bounds__(256) void rows_mask_kernel(bf16_t* __restrict__ x, int ld, int cols, const unsigned char* __restrict__ mask) {
    const size_t row = blockIdx.x;
    if (mask[row]) return;
    for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) *reinterpret_cast<u32x4*>(x + row * ld + c) = u32x4{0u, 0u, 0u, 0u};
}
// out[0] = n, out[1 .. n] = the 64-row tiles of [0, M) that hold a marked row, ascending (one wave: ballot + prefix count per 64 tiles)
__global__ __launch_bounds__(64) void rows_tile_list_kernel(const unsigned char* __restrict__ mask, int M, int* __restrict__ out) {
    const int lane = threadIdx.x, nt = (M + 63) / 64;
    int cnt = 0;
    for (int c = 0; c < nt; c += 64) {
        const int t = c + lane;
        bool any = false;
        if (t < nt) {
            const int r1 = min(M, t * 64 + 64);
            if (r1 - t * 64 == 64 && ((uintptr_t)(mask + t * 64) & 15) == 0) {
                // a whole tile: four 16-byte loads in flight (64 dependent byte loads took ~43 us per launch, once per layer)
                const u32x4* p = reinterpret_cast<const u32x4*>(mask + t * 64);
                const u32x4 a = p[0], b = p[1], c2 = p[2], d = p[3];
                const u32x4 o = a | b | c2 | d;
                any = (o[0] | o[1] | o[2] | o[3]) != 0u;
            } else {
                for (int r = t * 64; r < r1; ++r) any = any || mask[r] != 0;
            }
        }
        const unsigned long long b = __ballot(any);
        if (any) out[1 + cnt + __popcll(b & ((1ull << lane) - 1ull))] = t;
        cnt += __popcll(b);
    }
    if (lane == 0) out[0] = cnt;
}
extern "C" int vlr_rows_tile_list(const unsigned char* rowmask, int M, int* out, hipStream_t st) {
    VLR_REQUIRE(rowmask && out && M > 0, "vlr_rows_tile_list: bad arguments");
    hipLaunchKernelGGL(rows_tile_list_kernel, dim3(1), dim3(64), 0, st, rowmask, M, out);
    return vlr_check_launch("vlr_rows_tile_list");
}
// flags[t] = 1 when NO row of the `tile_rows`-row tile t of [0, M) is marked (the tiles an adapter-segment GEMM may run short: vlr_gemm_seg_rowskip)
__global__ __launch_bounds__(256) void rows_tile_flags_kernel(const unsigned char* __restrict__ mask, int M, int tile_rows, unsigned char* __restrict__ flags) {
    const int t = blockIdx.x, r0 = t * tile_rows, r1 = min(M, r0 + tile_rows);
    int any = 0;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) any |= mask[r] != 0;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flags[t] = any ? 0 : 1;
}
extern "C" int vlr_rows_tile_flags(const unsigned char* rowmask, int M, int tile_rows, unsigned char* flags, hipStream_t st) {
    VLR_REQUIRE(rowmask && flags && M > 0 && tile_rows > 0, "vlr_rows_tile_flags: bad arguments");
    hipLaunchKernelGGL(rows_tile_flags_kernel, dim3((M + tile_rows - 1) / tile_rows), dim3(256), 0, st, rowmask, M, tile_rows, flags);
    return vlr_check_launch("vlr_rows_tile_flags");
}
extern "C" int vlr_rows_mask(void* x, int ld, int cols, const unsigned char* rowmask, int M, hipStream_t st) {
    VLR_REQUIRE(x && rowmask && M > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols, "vlr_rows_mask: bad arguments");
    hipLaunchKernelGGL(rows_mask_kernel, dim3(M), dim3(256), 0, st, (bf16_t*)x, ld, cols, rowmask);
    return vlr_check_launch("vlr_rows_mask");
}
