// Workgroup -> (batch, head, 128-row block) map of the LDS-DMA attention kernels (attention.hip).  Plain integer arithmetic, host and
// device: tests/test_attn_grid.py compiles this header with g++ and checks that every (batch, head, block) is visited exactly once
// for a sweep of shapes, head groupings and bundle sizes.
#pragma once
#if defined(__HIPCC__)
#define VLR_HD __host__ __device__ __forceinline__
#else
#define VLR_HD inline
#endif

struct AttnGrid {
    int heads, kv_heads, group, nblk, n_kvp;   // n_kvp = batch * kv_heads
    int epi;                                   // 1: epilogue stores whole rows through LDS (write_rows_staged)
    unsigned* ctr;                             // persistent kernels: [8 XCDs][32] ticket / exit counters of this stream, else NULL
    int items;                                 // persistent kernels: work items per XCD
    // Order of the blocks inside an XCD.  The slots are sorted by work for the causal kernels (slot 0 = the block with the most
    // tiles) and the hardware hands workgroups to free CUs in index order - list scheduling.  K/V-head-major order (lpt = 1: all
    // blocks of one head, then the next head) keeps a head's K/V in the XCD's L2 but starts the last head's heaviest block when
    // the chip is nearly drained: 13-15 % over the balanced time at S = 1599, 30 % at S = 4096 / batch 1 (simulated and
    // measured, tools/attn_sweep.py).  Slot-major order over ALL heads is longest-job-first (within 1.5 % of balanced) but
    // streams every head's K/V through the L2 for every slot.  In between: bundles of `lpt` heads, slot-major inside a bundle,
    // the remainder bundle FIRST (a small last bundle has the same late-start problem).
    int lpt;
    VLR_HD void split(int s, int per, int& g0, int& gn, int& r) const {
        const int n_x = (n_kvp + 7) / 8;
        const int G = lpt < n_x ? lpt : n_x;
        int first = n_x % G;
        if (first == 0) first = G;
        if (s < first * per) { g0 = 0; gn = first; r = s; }
        else {
            const int s2 = s - first * per;
            g0 = first + (s2 / (G * per)) * G;
            gn = G;
            r = s2 % (G * per);
        }
    }
    VLR_HD int per_kvp(bool loop_members) const { return loop_members ? nblk : group * nblk; }
    VLR_HD int grid(bool loop_members) const { return 8 * ((n_kvp + 7) / 8) * per_kvp(loop_members); }
    // member-major inside a K/V head; returns false for the padding workgroups of the last XCD round
    VLR_HD bool decode(int L, int& head, int& kvhead, int& b, int& slot) const {
        const int xcd = L & 7, s = L >> 3, per = group * nblk;
        int g0, gn, r;
        split(s, per, g0, gn, r);
        const int gw = gn * group;           // slot-major inside the bundle; the query heads of one K/V head stay adjacent
        slot = r / gw;
        const int r2 = r % gw;
        const int member = r2 % group;
        const int kvp = (g0 + r2 / group) * 8 + xcd;
        if (kvp >= n_kvp) return false;
        b = kvp / kv_heads;
        kvhead = kvp % kv_heads;
        head = kvhead * group + member;
        return true;
    }
    // one workgroup per (K/V head, key block): the kernel loops over the group's query heads itself
    VLR_HD bool decode_kv(int L, int& kvhead, int& b, int& slot) const {
        const int xcd = L & 7, s = L >> 3;
        int g0, gn, r;
        split(s, nblk, g0, gn, r);
        const int kvp = (g0 + r % gn) * 8 + xcd;
        if (kvp >= n_kvp) return false;
        b = kvp / kv_heads;
        kvhead = kvp % kv_heads;
        slot = r / gn;
        return true;
    }
};
