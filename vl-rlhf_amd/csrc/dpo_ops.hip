// Integer / scalar side of the DPO hot path (gfx950): image-text merge index + gather/scatter, response-row
// compaction, per-token log-prob from logits, sequence sums, the DPO loss (all five reference loss types) and the
// flat-buffer optimizer (grad norm, clip coefficient, fused AdamW).
#include <stdlib.h>

#include "common.h"

#define IGNORE_INDEX (-100)
#define SRC_ZERO (INT32_MIN)

// ------------------------------------------------------------------------------------------------------------
// Merge index (LlavaForRL._merge_input_ids_with_image_features, reference Llava/__init__.py:36-109).
// One thread per batch row walks its T tokens.  src[b][s]: >= 0 text token index t; < 0 and != SRC_ZERO image
// feature row -(f+1) (f indexes the DEDUPLICATED feature table of n_feat_rows rows, see `dup`); SRC_ZERO = zeros.
// info[0] = number of image slots found, info[1] = expected (n_img_total*P) -> host raises ValueError on mismatch.
// ------------------------------------------------------------------------------------------------------------
__global__ void merge_index_kernel(const long* __restrict__ ids, const long* __restrict__ amask,
                                   const long* __restrict__ labels, int Bn, int T, int S, int P, int image_token,
                                   int pad_token, int n_feat_rows, int dup, int* __restrict__ src, int* __restrict__ out_mask,
                                   long* __restrict__ out_labels, int* __restrict__ out_pos,
                                   unsigned char* __restrict__ img_map, int* __restrict__ inv_map, int* __restrict__ info) {
    __shared__ int s_left;
    __shared__ int s_cnt[1024];
    const int b = threadIdx.x;
    if (b == 0) s_left = 1;
    __syncthreads();
    if (b < Bn && ids[(size_t)b * T + T - 1] == pad_token) s_left = 0;   // reference :39
    int nimg = 0;
    if (b < Bn)
        for (int t = 0; t < T; ++t) nimg += ids[(size_t)b * T + t] == image_token;
    s_cnt[b] = b < Bn ? nimg : 0;
    __syncthreads();
    if (b >= Bn) return;
    // rank offset of this row's image slots among all rows (row-major order of image_to_overwrite)
    int slot_base = 0, total_slots = 0;
    for (int i = 0; i < Bn; ++i) {
        const int last = T - 1 + s_cnt[i] * (P - 1);
        const int pad_i = S - 1 - last;
        // slots of row i = free positions with rank >= pad_i  =>  (S - (T - cnt_i)) - pad_i
        const int slots_i = (S - (T - s_cnt[i])) - pad_i;
        if (i < b) slot_base += slots_i;
        total_slots += slots_i;
    }
    const int nb_pad = S - 1 - (T - 1 + nimg * (P - 1));
    const int shift = s_left ? nb_pad : 0;
    int* srow = src + (size_t)b * S;
    int* mrow = out_mask + (size_t)b * S;
    long* lrow = out_labels + (size_t)b * S;
    for (int s = 0; s < S; ++s) {
        srow[s] = SRC_ZERO;
        mrow[s] = 0;
        lrow[s] = IGNORE_INDEX;
        img_map[(size_t)b * S + s] = 0;
    }
    // pass 1: text tokens
    int np = -1;
    for (int t = 0; t < T; ++t) {
        const long id = ids[(size_t)b * T + t];
        np += (id == image_token) ? P : 1;
        if (id != image_token) {
            const int d = np + shift;
            srow[d] = (id == pad_token) ? SRC_ZERO + 1 : t;   // SRC_ZERO+1: written-but-zeroed (reference step 6)
            mrow[d] = (int)amask[(size_t)b * T + t];
            lrow[d] = labels ? labels[(size_t)b * T + t] : IGNORE_INDEX;
        }
    }
    // pass 2: free positions with rank >= nb_pad are image slots, filled in order
    int rank = 0, k = 0;
    for (int s = 0; s < S; ++s) {
        if (srow[s] == SRC_ZERO) {
            if (rank >= nb_pad) {
                const int g = slot_base + k;           // global rank over the whole (2B) batch
                const int f = g % n_feat_rows;         // duplicated images share one feature row
                srow[s] = -(f + 1);
                if (g / n_feat_rows < dup) inv_map[(size_t)(g / n_feat_rows) * n_feat_rows + f] = b * S + s;
                mrow[s] = 1;
                img_map[(size_t)b * S + s] = 1;
                ++k;
            }
            ++rank;
        } else if (srow[s] == SRC_ZERO + 1) {
            srow[s] = SRC_ZERO;
        }
    }
    // position ids = cumsum(mask) - 1, 1 where masked (reference :98)
    int c = 0;
    for (int s = 0; s < S; ++s) {
        c += mrow[s] != 0;
        out_pos[(size_t)b * S + s] = mrow[s] ? c - 1 : 1;
    }
    atomicAdd(&info[0], k);   // host compares with n_feat_rows * dup (reference :90-94 raises ValueError)
    (void)total_slots;
}

// Block-parallel version of the same map: one 256-thread workgroup per batch row, the three serial walks (token -> new
// position, free position -> image-slot rank, mask -> position id) become chunked block scans.  Bit-identical outputs
// (tests/test_hip_kernels.py compares both with the golden merge); the serial kernel above stays as the Bn > 64 path.
__device__ __forceinline__ int block_exclusive_scan256(int v, int* sh, int* total) {
    const int t = threadIdx.x;
    __syncthreads();
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int o = 1; o < 256; o <<= 1) {
        const int add = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    const int incl = sh[t];
    *total = sh[255];
    return incl - v;
}
__global__ __launch_bounds__(256) void merge_index_block_kernel(const long* __restrict__ ids, const long* __restrict__ amask,
                                                                const long* __restrict__ labels, int Bn, int T, int S, int P,
                                                                int image_token, int pad_token, int n_feat_rows, int dup,
                                                                int* __restrict__ src, int* __restrict__ out_mask,
                                                                long* __restrict__ out_labels, int* __restrict__ out_pos,
                                                                unsigned char* __restrict__ img_map, int* __restrict__ inv_map,
                                                                int* __restrict__ info) {
    __shared__ int sh[256];
    __shared__ int s_cnt[64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int not_left = 0;
    if (t < Bn) not_left = ids[(size_t)t * T + T - 1] == pad_token;          // reference :39
    const int s_left = !__syncthreads_or(not_left);
    for (int i = wave; i < Bn; i += 4) {                                      // image tokens per row
        int c = 0;
        for (int k = lane; k < T; k += 64) c += ids[(size_t)i * T + k] == image_token;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) s_cnt[i] = c;
    }
    __syncthreads();
    int slot_base = 0;
    for (int i = 0; i < b; ++i) {
        const int pad_i = S - 1 - (T - 1 + s_cnt[i] * (P - 1));
        slot_base += (S - (T - s_cnt[i])) - pad_i;
    }
    const int nimg = s_cnt[b];
    const int nb_pad = S - 1 - (T - 1 + nimg * (P - 1));
    const int shift = s_left ? nb_pad : 0;
    int* srow = src + (size_t)b * S;
    int* mrow = out_mask + (size_t)b * S;
    long* lrow = out_labels + (size_t)b * S;
    unsigned char* irow = img_map + (size_t)b * S;
    for (int k = t; k < S; k += 256) {
        srow[k] = SRC_ZERO;
        mrow[k] = 0;
        lrow[k] = IGNORE_INDEX;
        irow[k] = 0;
    }
    // pass 1: text tokens -> new positions (np = inclusive prefix of {P for <image>, 1 otherwise} - 1)
    const int ct = (T + 255) / 256, t0 = min(T, t * ct), t1 = min(T, t0 + ct);
    int inc = 0, tot;
    for (int k = t0; k < t1; ++k) inc += ids[(size_t)b * T + k] == image_token ? P : 1;
    int np = block_exclusive_scan256(inc, sh, &tot) - 1;        // also the barrier after the row initialisation
    for (int k = t0; k < t1; ++k) {
        const long id = ids[(size_t)b * T + k];
        np += (id == image_token) ? P : 1;
        if (id != image_token) {
            const int d = np + shift;
            srow[d] = (id == pad_token) ? SRC_ZERO + 1 : k;      // SRC_ZERO+1: written-but-zeroed (reference step 6)
            mrow[d] = (int)amask[(size_t)b * T + k];
            lrow[d] = labels ? labels[(size_t)b * T + k] : IGNORE_INDEX;
        }
    }
    __syncthreads();
    // pass 2: free positions with rank >= nb_pad are image slots, filled in order
    const int cs = (S + 255) / 256, s0 = min(S, t * cs), s1 = min(S, s0 + cs);
    int nfree = 0;
    for (int k = s0; k < s1; ++k) nfree += srow[k] == SRC_ZERO;
    int total_free;
    int rank = block_exclusive_scan256(nfree, sh, &total_free);
    for (int k = s0; k < s1; ++k) {
        const int v = srow[k];
        if (v == SRC_ZERO) {
            if (rank >= nb_pad) {
                const int g = slot_base + (rank - nb_pad);       // global rank over the whole (2B) batch
                const int f = g % n_feat_rows;                    // duplicated images share one feature row
                srow[k] = -(f + 1);
                if (g / n_feat_rows < dup) inv_map[(size_t)(g / n_feat_rows) * n_feat_rows + f] = b * S + k;
                mrow[k] = 1;
                irow[k] = 1;
            }
            ++rank;
        } else if (v == SRC_ZERO + 1) {
            srow[k] = SRC_ZERO;
        }
    }
    if (t == 0) atomicAdd(&info[0], max(0, total_free - max(nb_pad, 0)));   // host compares with n_feat_rows * dup
    __syncthreads();
    // position ids = cumsum(mask) - 1, 1 where masked (reference :98)
    int nm = 0, tm;
    for (int k = s0; k < s1; ++k) nm += mrow[k] != 0;
    int c = block_exclusive_scan256(nm, sh, &tm);
    for (int k = s0; k < s1; ++k) {
        c += mrow[k] != 0;
        out_pos[(size_t)b * S + k] = mrow[k] ? c - 1 : 1;
    }
}

// embeds[b][s][:] = embed_tokens[ids[b][t]] | image_features[f] | 0
__global__ __launch_bounds__(256) void merge_gather_kernel(const int* __restrict__ src, const long* __restrict__ ids,
                                                           const bf16_t* __restrict__ table,
                                                           const bf16_t* __restrict__ feats, bf16_t* __restrict__ out,
                                                           int T, int S, int H) {
    const size_t pos = blockIdx.x;         // b*S + s
    const int b = (int)(pos / S);
    const int sv = src[pos];
    const bf16_t* from = nullptr;
    if (sv >= 0) from = table + (size_t)ids[(size_t)b * T + sv] * H;
    else if (sv != SRC_ZERO) from = feats + (size_t)(-(sv + 1)) * H;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (from) v = *reinterpret_cast<const u32x4*>(from + c);
        *reinterpret_cast<u32x4*>(out + pos * H + c) = v;
    }
}
// the same for the fp32 residual stream: out fp32; embedding rows are bf16 (exact in fp32), feature rows fp32 (the projector's
// unrounded output) or bf16 (feats_f32 = 0)
__global__ __launch_bounds__(256) void merge_gather_f32_kernel(const int* __restrict__ src, const long* __restrict__ ids,
                                                               const bf16_t* __restrict__ table, const void* __restrict__ feats,
                                                               int feats_f32, float* __restrict__ out, int T, int S, int H) {
    const size_t pos = blockIdx.x;
    const int b = (int)(pos / S);
    const int sv = src[pos];
    const bf16_t* from16 = nullptr;
    const float* from32 = nullptr;
    if (sv >= 0) from16 = table + (size_t)ids[(size_t)b * T + sv] * H;
    else if (sv != SRC_ZERO) {
        if (feats_f32) from32 = reinterpret_cast<const float*>(feats) + (size_t)(-(sv + 1)) * H;
        else from16 = reinterpret_cast<const bf16_t*>(feats) + (size_t)(-(sv + 1)) * H;
    }
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (from32) {
            lo = *reinterpret_cast<const f32x4*>(from32 + c);
            hi = *reinterpret_cast<const f32x4*>(from32 + c + 4);
        } else if (from16) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(from16 + c), v);
            lo = f32x4{v[0], v[1], v[2], v[3]};
            hi = f32x4{v[4], v[5], v[6], v[7]};
        }
        *reinterpret_cast<f32x4*>(out + pos * H + c) = lo;
        *reinterpret_cast<f32x4*>(out + pos * H + c + 4) = hi;
    }
}
// d_feats[f] = sum over the `dup` positions that consumed feature row f
__global__ __launch_bounds__(256) void merge_bwd_feats_kernel(const bf16_t* __restrict__ dmerged,
                                                              const int* __restrict__ inv_map, bf16_t* __restrict__ dfeats,
                                                              int n_feat_rows, int dup, int H) {
    const int f = blockIdx.x;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < dup; ++d) {
            const int pos = inv_map[(size_t)d * n_feat_rows + f];
            if (pos >= 0) {
                float v[8];
                unpack8(*reinterpret_cast<const u32x4*>(dmerged + (size_t)pos * H + c), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += v[e];
            }
        }
        *reinterpret_cast<u32x4*>(dfeats + (size_t)f * H + c) = pack8(a);
    }
}
// d_embed_tokens[id] += sum over the positions holding token `id` of d_merged[pos]  - DETERMINISTIC: the workgroup of the FIRST
// position of an id gathers every later position with the same id in position order, accumulates in fp32 and writes the row
// once (read-modify-write, no atomics: each table row has exactly one writer).  The earlier version scatter-added with packed
// bf16 atomics; their arrival order changed the rounding from run to run, and after three AdamW steps on random weights the
// 7B loss differed by 5 % between two runs of the same binary (tools/determinism.py).
__global__ __launch_bounds__(256) void merge_bwd_embed_kernel(const bf16_t* __restrict__ dmerged,
                                                              const int* __restrict__ src, const long* __restrict__ ids,
                                                              bf16_t* __restrict__ dtable, int T, int S, int H, int npos) {
    __shared__ int s_hit[256];
    const int pos = blockIdx.x, t = threadIdx.x;
    const int sv = src[pos];
    if (sv < 0) return;                                    // image feature row or zero row: no embedding behind it
    const long id = ids[(size_t)(pos / S) * T + sv];
    auto id_at = [&](int q) -> long {
        const int sq = src[q];
        return sq < 0 ? -1 : ids[(size_t)(q / S) * T + sq];
    };
    int earlier = 0;
    for (int q = t; q < pos; q += 256) earlier |= id_at(q) == id;
    if (__syncthreads_or(earlier)) return;                 // an earlier position owns this id
    constexpr int NB = 4;                                  // thread t owns columns [8*(t + 256 k), +8), k < NB: H <= 8192
    float acc[NB][8];
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    for (int q0 = pos; q0 < npos; q0 += 256) {             // every thread takes part in the barriers, whatever H is
        __syncthreads();
        s_hit[t] = (q0 + t < npos) && id_at(q0 + t) == id;
        __syncthreads();
        for (int h = 0; h < 256; ++h) {
            if (!s_hit[h]) continue;                       // uniform: every thread walks the hits in position order
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int c0 = (t + 256 * k) * 8;
                if (c0 < H) {
                    float v[8];
                    unpack8(*reinterpret_cast<const u32x4*>(dmerged + (size_t)(q0 + h) * H + c0), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[k][e] += v[e];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int c0 = (t + 256 * k) * 8;
        if (c0 < H) {
            bf16_t* dst = dtable + (size_t)id * H + c0;
            float o[8];
            unpack8(*reinterpret_cast<const u32x4*>(dst), o);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += acc[k][e];
            *reinterpret_cast<u32x4*>(dst) = pack8(o);
        }
    }
}

// The same sums in the same order (bit-identical), without the quadratic walk through global memory (round 6): merge_bwd_embed_kernel
// spends one workgroup per position on two dependent loads (src, then ids) for EVERY earlier position just to learn whether it owns its
// id - 82 M of them at 8 x 1599 positions, 1.4 ms of a step for a 0.05 ms gather.  Here a workgroup takes EMB_PB consecutive positions,
// loads the token id of every position into LDS once, and each WAVE scans that array by itself (64 positions per ds_read + ballot, no
// barrier): first for an earlier holder of the id, then - owners only - for the later positions to add, in position order; a thread
// accumulates its own columns, so the four waves never have to meet.  npos ids must fit 156 KiB of dynamic LDS (vlr_merge_bwd falls back).
#define EMB_PB 16
__global__ __launch_bounds__(256) void merge_bwd_embed2_kernel(const bf16_t* __restrict__ dmerged, const int* __restrict__ src,
                                                               const long* __restrict__ ids, bf16_t* __restrict__ dtable, int T, int S,
                                                               int H, int npos) {
    extern __shared__ int pid[];                           // token id per position, -1 = image feature row / zero row
    const int t = threadIdx.x, lane = t & 63;
    for (int q = t; q < npos; q += 1024) {                 // four positions per thread and trip: the id load depends on the src load
        int sq[4], id4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sq[u] = q + 256 * u < npos ? src[q + 256 * u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) id4[u] = sq[u] < 0 ? -1 : (int)ids[(size_t)((q + 256 * u) / S) * T + sq[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q + 256 * u < npos) pid[q + 256 * u] = id4[u];
    }
    __syncthreads();
    const int p0 = blockIdx.x * EMB_PB;
    for (int j = 0; j < EMB_PB; ++j) {
        const int pos = p0 + j;
        if (pos >= npos) break;
        const int id = pid[pos];
        if (id < 0) continue;                              // (uniform)
        unsigned long long earlier = 0ull;
        for (int q0 = 0; q0 < pos && !earlier; q0 += 64) earlier = __builtin_amdgcn_ballot_w64(q0 + lane < pos && pid[q0 + lane] == id);
        if (earlier) continue;                             // an earlier position owns this id (every wave finds the same answer)
        constexpr int NB = 4;                              // thread t owns columns [8*(t + 256 k), +8), k < NB: H <= 8192
        float acc[NB][8];
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
        for (int q0 = pos & ~63; q0 < npos; q0 += 64) {
            const int q = q0 + lane;
            unsigned long long hits = __builtin_amdgcn_ballot_w64(q >= pos && q < npos && pid[q] == id);
            while (hits) {                                 // in position order
                const int h = __builtin_ctzll(hits);
                hits &= hits - 1;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int c0 = (t + 256 * k) * 8;
                    if (c0 < H) {
                        float v[8];
                        unpack8(*reinterpret_cast<const u32x4*>(dmerged + (size_t)(q0 + h) * H + c0), v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[k][e] += v[e];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int c0 = (t + 256 * k) * 8;
            if (c0 < H) {
                bf16_t* dst = dtable + (size_t)id * H + c0;
                float o[8];
                unpack8(*reinterpret_cast<const u32x4*>(dst), o);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += acc[k][e];
                *reinterpret_cast<u32x4*>(dst) = pack8(o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Response-row compaction: rows[i] = b*S+s for every (b, s < S-1) with labels[b][s+1] != pad (and shared[b][s] when
// given: DDPO mask on shifted positions).  One workgroup, sequences in order, so rows are sorted and each sequence
// owns the contiguous range [seq_off[b], seq_off[b+1]).  tgt[i] = label, also emitted.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void build_rows_kernel(const long* __restrict__ labels,
                                                          const unsigned char* __restrict__ shared, int Bn, int S,
                                                          int label_pad, int* __restrict__ rows, int* __restrict__ tgt,
                                                          int* __restrict__ seq_off) {
    __shared__ int s_scan[1024];
    __shared__ int s_base;
    if (threadIdx.x == 0) { s_base = 0; seq_off[0] = 0; }
    __syncthreads();
    for (int b = 0; b < Bn; ++b) {
        for (int s0 = 0; s0 < S - 1; s0 += 1024) {
            const int s = s0 + threadIdx.x;
            int keep = 0;
            long lab = 0;
            if (s < S - 1) {
                lab = labels[(size_t)b * S + s + 1];
                keep = lab != label_pad;
                if (keep && shared) keep = shared[(size_t)b * (S - 1) + s] != 0;
            }
            s_scan[threadIdx.x] = keep;
            __syncthreads();
            for (int o = 1; o < 1024; o <<= 1) {
                int v = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
                __syncthreads();
                s_scan[threadIdx.x] += v;
                __syncthreads();
            }
            const int base = s_base;
            if (keep) {
                const int i = base + s_scan[threadIdx.x] - 1;
                rows[i] = b * S + s;
                tgt[i] = (int)lab;
            }
            __syncthreads();
            if (threadIdx.x == 1023) s_base = base + s_scan[1023];
            __syncthreads();
        }
        if (threadIdx.x == 0) seq_off[b + 1] = s_base;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// Per-row log-softmax pick:  tok_logp[i] = logits[i][tgt[i]] - logsumexp(logits[i][:]);  lse[i] kept for backward.
// logits fp32 [R][ld] (row r = row_idx ? row_idx[i] : i).  One workgroup per row, single pass (online max/sum).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logp_rows_kernel(const float* __restrict__ logits, const int* __restrict__ row_idx,
                                                        const int* __restrict__ tgt, int V, long ld,
                                                        float* __restrict__ tok_logp, float* __restrict__ lse) {
    __shared__ float red[16];
    const int i = blockIdx.x;
    const float* l = logits + (row_idx ? (long)row_idx[i] : (long)i) * ld;
    float m = -INFINITY, s = 0.f;
    for (int c = threadIdx.x * 4; c < V; c += 256 * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(l + c);
        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (mx > m) { s *= __expf(m - mx); m = mx; }
        s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
    }
    const float M = block_max(m, red);
    s = (m == -INFINITY) ? 0.f : s * __expf(m - M);
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float z = M + logf(s);
        lse[i] = z;
        tok_logp[i] = l[tgt[i]] - z;
    }
}
// dlogits[i][v] = g_i * ([v == tgt_i] - exp(logits[i][v] - lse_i)),  g_i = dlogps[seq(i)] * scale_i   (bf16 out)
__global__ __launch_bounds__(256) void dlogits_rows_kernel(const float* __restrict__ logits, const int* __restrict__ tgt,
                                                           const float* __restrict__ lse, const int* __restrict__ seq_off,
                                                           int nseq, const float* __restrict__ dlogps, int average,
                                                           int V, long ld, bf16_t* __restrict__ dl, long ldd) {
    const int i = blockIdx.x;
    int b = 0;
    while (b + 1 < nseq && seq_off[b + 1] <= i) ++b;
    float g = dlogps[b];
    if (average) g /= (float)(seq_off[b + 1] - seq_off[b]);
    const float* l = logits + (long)i * ld;
    const float z = lse[i];
    const int t = tgt[i];
    for (int c = threadIdx.x * 8; c < V; c += 256 * 8) {
        float v[8];
        *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(l + c);
        *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(l + c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g * ((c + e == t ? 1.f : 0.f) - __expf(v[e] - z));
        *reinterpret_cast<u32x4*>(dl + (long)i * ldd + c) = pack8(v);
    }
}
// logps[b] = sum_{i in seq b} tok_logp[i]  (/ count when average)
__global__ __launch_bounds__(256) void seq_sum_kernel(const float* __restrict__ tok, const int* __restrict__ seq_off,
                                                      int average, float* __restrict__ out) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const int lo = seq_off[b], hi = seq_off[b + 1];
    float s = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += tok[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[b] = average ? s / (float)(hi - lo) : s;
}

// ------------------------------------------------------------------------------------------------------------
// DPO loss, forward and backward in one launch (reference VLDPOTrainer.dpo_loss, base/trainer.py:244-301).
// loss_type: 0 sigmoid / ddpo, 1 hinge, 2 ipo, 3 kto_pair.  n pairs <= 1024, one workgroup.
// outputs: losses[n] (kto: [2n]), chosen_rewards[n], rejected_rewards[n], and d(mean loss)/d(pc,pr) * gscale.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(1024) void dpo_loss_kernel(const float* __restrict__ pc, const float* __restrict__ pr,
                                                        const float* __restrict__ rc, const float* __restrict__ rr, int n,
                                                        float beta, float ls, int loss_type, int reference_free,
                                                        float* __restrict__ losses, float* __restrict__ cr,
                                                        float* __restrict__ rrw, float* __restrict__ dpc,
                                                        float* __restrict__ dpr, float* __restrict__ loss_mean,
                                                        const float* __restrict__ gl) {
    __shared__ float red[16];
    const int i = threadIdx.x;
    const bool on = i < n;
    const float a = on ? pc[i] : 0.f, b = on ? pr[i] : 0.f, c = on ? rc[i] : 0.f, d = on ? rr[i] : 0.f;
    const float x = (a - b) - (reference_free ? 0.f : (c - d));
    float li = 0.f, li2 = 0.f, ga = 0.f, gb = 0.f;
    const int nl = loss_type == 3 ? 2 * n : n;
    // upstream gradient per loss element (default: d mean / d loss_i = 1/nl)
    const float g1 = on ? (gl ? gl[i] : 1.f / (float)nl) : 0.f;
    const float g2 = (on && loss_type == 3) ? (gl ? gl[n + i] : 1.f / (float)nl) : 0.f;
    if (loss_type == 0) {
        li = -log_sigmoid(beta * x) * (1.f - ls) - log_sigmoid(-beta * x) * ls;
        const float gx = -beta * (1.f - ls) * sigmoidf_(-beta * x) + beta * ls * sigmoidf_(beta * x);
        ga = gx; gb = -gx;
    } else if (loss_type == 1) {
        li = fmaxf(1.f - beta * x, 0.f);
        const float gx = (1.f - beta * x) > 0.f ? -beta : 0.f;
        ga = gx; gb = -gx;
    } else if (loss_type == 2) {
        const float u = x - 1.f / (2.f * beta);
        li = u * u;
        ga = 2.f * u; gb = -2.f * u;
    } else {
        const float ckl_raw = block_sum(on ? (a - c) : 0.f, red) / (float)n;
        const float rkl_raw = block_sum(on ? (b - d) : 0.f, red) / (float)n;
        const float ckl = fmaxf(ckl_raw, 0.f), rkl = fmaxf(rkl_raw, 0.f);
        const float u = beta * ((a - c) - rkl), v = beta * (ckl - (b - d));
        const float su = sigmoidf_(u), sv = sigmoidf_(v);
        li = 1.f - su;
        li2 = 1.f - sv;
        const float du = g1 * -beta * su * (1.f - su);   // g * d li / d(a-c-rkl)
        const float dv = g2 * -beta * sv * (1.f - sv);   // g * d li2 / d(ckl-(b-d))
        const float sum_du = block_sum(du, red), sum_dv = block_sum(dv, red);
        ga = du + (ckl_raw > 0.f ? sum_dv / (float)n : 0.f);
        gb = -dv - (rkl_raw > 0.f ? sum_du / (float)n : 0.f);
    }
    if (loss_type != 3) { ga *= g1; gb *= g1; }
    const float tot = block_sum(on ? li + li2 : 0.f, red);
    if (on) {
        losses[i] = li;
        if (loss_type == 3) losses[n + i] = li2;
        cr[i] = beta * (a - c);
        rrw[i] = beta * (b - d);
        dpc[i] = ga;
        dpr[i] = gb;
    }
    if (i == 0) *loss_mean = tot / (float)nl;
}

// ------------------------------------------------------------------------------------------------------------
// Optimizer on flat buffers.  sqnorm: two stages, deterministic.  clip: coef = min(1, max_norm/(norm+1e-6)) * gscale.
// AdamW (torch.optim.AdamW): p *= 1-lr*wd; m,v EMA; p -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps).  28 B per parameter.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const bf16_t* __restrict__ g, long n8, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    const long step = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n8; i += 4 * step) {             // four loads in flight, the squares added in the order of the plain loop
        u32x4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const u32x4*>(g + (i + u * step) * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[8];
            unpack8(w[u], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e] * v[e];
        }
    }
    for (; i < n8; i += step) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(g + i * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e] * v[e];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ part, int P, float extra_sq,
                                                           float max_norm, float gscale, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(s * gscale * gscale + extra_sq);   // norm of the (already scaled) gradient
        out[0] = nrm;
        out[1] = (max_norm > 0.f ? fminf(1.f, max_norm / (nrm + 1e-6f)) : 1.f) * gscale;
        out[2] = s;
    }
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const bf16_t* __restrict__ g, bf16_t* __restrict__ p16, long n8,
                                                    float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                    float rsqrt_bc2, const float* __restrict__ coef_ptr) {
    const float coef = coef_ptr ? coef_ptr[1] : 1.f;
    const float step = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float gr[8], w[8], mm[8], vv[8];
        unpack8(*reinterpret_cast<const u32x4*>(g + i * 8), gr);
        *reinterpret_cast<f32x4*>(w) = *reinterpret_cast<const f32x4*>(master + i * 8);
        *reinterpret_cast<f32x4*>(w + 4) = *reinterpret_cast<const f32x4*>(master + i * 8 + 4);
        *reinterpret_cast<f32x4*>(mm) = *reinterpret_cast<const f32x4*>(m + i * 8);
        *reinterpret_cast<f32x4*>(mm + 4) = *reinterpret_cast<const f32x4*>(m + i * 8 + 4);
        *reinterpret_cast<f32x4*>(vv) = *reinterpret_cast<const f32x4*>(v + i * 8);
        *reinterpret_cast<f32x4*>(vv + 4) = *reinterpret_cast<const f32x4*>(v + i * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = gr[e] * coef;
            w[e] *= 1.f - lr * wd;
            mm[e] = beta1 * mm[e] + (1.f - beta1) * gg;
            vv[e] = beta2 * vv[e] + (1.f - beta2) * gg * gg;
            w[e] -= step * mm[e] / (sqrtf(vv[e]) * rsqrt_bc2 + eps);
        }
        *reinterpret_cast<f32x4*>(master + i * 8) = *reinterpret_cast<f32x4*>(w);
        *reinterpret_cast<f32x4*>(master + i * 8 + 4) = *reinterpret_cast<f32x4*>(w + 4);
        *reinterpret_cast<f32x4*>(m + i * 8) = *reinterpret_cast<f32x4*>(mm);
        *reinterpret_cast<f32x4*>(m + i * 8 + 4) = *reinterpret_cast<f32x4*>(mm + 4);
        *reinterpret_cast<f32x4*>(v + i * 8) = *reinterpret_cast<f32x4*>(vv);
        *reinterpret_cast<f32x4*>(v + i * 8 + 4) = *reinterpret_cast<f32x4*>(vv + 4);
        *reinterpret_cast<u32x4*>(p16 + i * 8) = pack8(w);
    }
}

// ============================================================================================================
extern "C" int vlr_merge_index(const long* input_ids, const long* attention_mask, const long* labels, int Bn, int T,
                               int S, int P, int image_token, int pad_token, int n_feat_rows, int dup, int* src,
                               int* out_mask, long* out_labels, int* out_pos, unsigned char* img_map, int* inv_map,
                               int* info, hipStream_t st) {
    VLR_REQUIRE(Bn > 0 && Bn <= 1024 && T > 0 && S >= T && P > 0, "vlr_merge_index: bad shape Bn=%d T=%d S=%d P=%d", Bn, T, S, P);
    VLR_REQUIRE(n_feat_rows > 0 && dup >= 1, "vlr_merge_index: n_feat_rows/dup");
    hipMemsetAsync(inv_map, 0xff, (size_t)dup * n_feat_rows * sizeof(int), st);
    hipMemsetAsync(info, 0, 2 * sizeof(int), st);
    static int serial = -1;
    if (serial < 0) { const char* e = getenv("VLR_MERGE_SERIAL"); serial = (e && e[0] == '1') ? 1 : 0; }
    if (Bn <= 64 && !serial)
        hipLaunchKernelGGL(merge_index_block_kernel, dim3(Bn), dim3(256), 0, st, input_ids, attention_mask, labels, Bn, T, S, P,
                           image_token, pad_token, n_feat_rows, dup, src, out_mask, out_labels, out_pos, img_map, inv_map, info);
    else
        hipLaunchKernelGGL(merge_index_kernel, dim3(1), dim3(1024), 0, st, input_ids, attention_mask, labels, Bn, T, S, P,
                           image_token, pad_token, n_feat_rows, dup, src, out_mask, out_labels, out_pos, img_map, inv_map, info);
    return vlr_check_launch("vlr_merge_index");
}
extern "C" int vlr_merge_fwd(const int* src, const long* input_ids, const void* embed_table, const void* feats,
                             void* out, int Bn, int T, int S, int H, hipStream_t st) {
    VLR_REQUIRE(Bn > 0 && H % 8 == 0, "vlr_merge_fwd: bad shape");
    hipLaunchKernelGGL(merge_gather_kernel, dim3(Bn * S), dim3(256), 0, st, src, input_ids, (const bf16_t*)embed_table,
                       (const bf16_t*)feats, (bf16_t*)out, T, S, H);
    return vlr_check_launch("vlr_merge_fwd");
}
extern "C" int vlr_merge_fwd_f32(const int* src, const long* input_ids, const void* embed_table, const void* feats, int feats_f32,
                                 float* out, int Bn, int T, int S, int H, hipStream_t st) {
    VLR_REQUIRE(src && input_ids && embed_table && out, "vlr_merge_fwd_f32: null argument");
    VLR_REQUIRE(Bn > 0 && H % 8 == 0, "vlr_merge_fwd_f32: bad shape");
    hipLaunchKernelGGL(merge_gather_f32_kernel, dim3(Bn * S), dim3(256), 0, st, src, input_ids, (const bf16_t*)embed_table, feats,
                       feats_f32, out, T, S, H);
    return vlr_check_launch("vlr_merge_fwd_f32");
}
extern "C" int vlr_merge_bwd(const void* dmerged, const int* src, const int* inv_map, const long* input_ids,
                             void* dfeats, void* dembed_table, int Bn, int T, int S, int H, int n_feat_rows, int dup,
                             hipStream_t st) {
    VLR_REQUIRE(Bn > 0 && H % 8 == 0 && H <= 8192, "vlr_merge_bwd: bad shape (H %% 8 == 0, H <= 8192)");
    if (dfeats)
        hipLaunchKernelGGL(merge_bwd_feats_kernel, dim3(n_feat_rows), dim3(256), 0, st, (const bf16_t*)dmerged, inv_map,
                           (bf16_t*)dfeats, n_feat_rows, dup, H);
    if (dembed_table) {
        const int npos = Bn * S;
        static int emb2 = -1;
        constexpr size_t EMB2_LDS = 156 * 1024;                   // the positions' ids in LDS: up to 39936 positions (LLaVA-Next recipe: 8 x 4814)
        if (emb2 < 0) {
            const char* e = getenv("VLR_MERGE_EMBED2");
            emb2 = (e && e[0] == '0') ? 0 : 1;
            if (hipFuncSetAttribute((const void*)merge_bwd_embed2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EMB2_LDS) != hipSuccess) emb2 = 0;
        }
        if (emb2 && (size_t)npos * sizeof(int) <= EMB2_LDS)
            hipLaunchKernelGGL(merge_bwd_embed2_kernel, dim3((npos + EMB_PB - 1) / EMB_PB), dim3(256), (size_t)npos * sizeof(int), st,
                               (const bf16_t*)dmerged, src, input_ids, (bf16_t*)dembed_table, T, S, H, npos);
        else
            hipLaunchKernelGGL(merge_bwd_embed_kernel, dim3(npos), dim3(256), 0, st, (const bf16_t*)dmerged, src, input_ids,
                               (bf16_t*)dembed_table, T, S, H, npos);
    }
    return vlr_check_launch("vlr_merge_bwd");
}
extern "C" int vlr_build_rows(const long* labels, const unsigned char* shared_mask, int Bn, int S, int label_pad,
                              int* rows, int* tgt, int* seq_off, hipStream_t st) {
    VLR_REQUIRE(Bn > 0 && S > 1, "vlr_build_rows: bad shape");
    hipLaunchKernelGGL(build_rows_kernel, dim3(1), dim3(1024), 0, st, labels, shared_mask, Bn, S, label_pad, rows, tgt,
                       seq_off);
    return vlr_check_launch("vlr_build_rows");
}
extern "C" int vlr_logp_rows(const float* logits, const int* row_idx, const int* tgt, int R, int V, long ld,
                             float* tok_logp, float* lse, hipStream_t st) {
    VLR_REQUIRE(R > 0 && V % 4 == 0 && ld % 4 == 0, "vlr_logp_rows: bad shape R=%d V=%d", R, V);
    hipLaunchKernelGGL(logp_rows_kernel, dim3(R), dim3(256), 0, st, logits, row_idx, tgt, V, ld, tok_logp, lse);
    return vlr_check_launch("vlr_logp_rows");
}
extern "C" int vlr_dlogits_rows(const float* logits, const int* tgt, const float* lse, const int* seq_off, int nseq,
                                const float* dlogps, int average, int R, int V, long ld, void* dlogits, long ldd,
                                hipStream_t st) {
    VLR_REQUIRE(R > 0 && V % 8 == 0 && ld % 4 == 0 && ldd % 8 == 0, "vlr_dlogits_rows: bad shape");
    hipLaunchKernelGGL(dlogits_rows_kernel, dim3(R), dim3(256), 0, st, logits, tgt, lse, seq_off, nseq, dlogps, average, V,
                       ld, (bf16_t*)dlogits, ldd);
    return vlr_check_launch("vlr_dlogits_rows");
}
extern "C" int vlr_seq_sum(const float* tok_logp, const int* seq_off, int nseq, int average, float* out,
                           hipStream_t st) {
    VLR_REQUIRE(nseq > 0, "vlr_seq_sum: nseq");
    hipLaunchKernelGGL(seq_sum_kernel, dim3(nseq), dim3(256), 0, st, tok_logp, seq_off, average, out);
    return vlr_check_launch("vlr_seq_sum");
}
extern "C" int vlr_dpo_loss(const float* pc, const float* pr, const float* rc, const float* rr, int n, float beta,
                            float label_smoothing, int loss_type, int reference_free, float* losses,
                            float* chosen_rewards, float* rejected_rewards, float* dpc, float* dpr, float* loss_mean,
                            const float* grad_losses, hipStream_t st) {
    VLR_REQUIRE(n > 0 && n <= 1024, "vlr_dpo_loss: 1 <= n <= 1024 pairs per rank, got %d", n);
    VLR_REQUIRE(loss_type >= 0 && loss_type <= 3,
                "Unknown loss type: %d. Should be one of ['sigmoid', 'hinge', 'ipo', 'kto_pair']", loss_type);
    hipLaunchKernelGGL(dpo_loss_kernel, dim3(1), dim3(1024), 0, st, pc, pr, rc, rr, n, beta, label_smoothing, loss_type,
                       reference_free, losses, chosen_rewards, rejected_rewards, dpc, dpr, loss_mean, grad_losses);
    return vlr_check_launch("vlr_dpo_loss");
}

#define VLR_SQNORM_BLOCKS 1024
extern "C" int vlr_grad_sqnorm_workspace_bytes(void) { return VLR_SQNORM_BLOCKS * 4; }
extern "C" int vlr_grad_sqnorm(const void* grads, long n, float max_norm, float gscale, float extra_sq, void* workspace,
                               float* out3, hipStream_t st) {
    VLR_REQUIRE(n > 0 && n % 8 == 0 && workspace && out3, "vlr_grad_sqnorm: n %% 8 and workspace/out required");
    const long n8 = n / 8;
    const int P = (int)(n8 < VLR_SQNORM_BLOCKS * 256L ? (n8 + 255) / 256 : VLR_SQNORM_BLOCKS);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(P), dim3(256), 0, st, (const bf16_t*)grads, n8, (float*)workspace);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, P, extra_sq, max_norm,
                       gscale, out3);
    return vlr_check_launch("vlr_grad_sqnorm");
}
extern "C" int vlr_adamw_step(float* master, float* m, float* v, const void* grads, void* params_bf16, long n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, const float* coef,
                              hipStream_t st) {
    VLR_REQUIRE(n > 0 && n % 8 == 0 && step >= 1, "vlr_adamw_step: n %% 8 == 0 and step >= 1");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    const long n8 = n / 8;
    const int G = (int)(n8 < 256L * 256 * 8 ? (n8 + 255) / 256 : 256 * 8);
    hipLaunchKernelGGL(adamw_kernel, dim3(G), dim3(256), 0, st, master, m, v, (const bf16_t*)grads, (bf16_t*)params_bf16,
                       n8, lr, beta1, beta2, eps, weight_decay, bc1, 1.f / sqrtf(bc2), coef);
    return vlr_check_launch("vlr_adamw_step");
}
