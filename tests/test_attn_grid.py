"""The attention kernels' workgroup -> (batch, head, block) map (vl-rlhf_amd/csrc/attn_grid.h) is plain integer code shared by host and
device: compiled here with g++ and enumerated - every (batch, query head, block) of the forward / dQ grid and every (batch, K/V head,
block) of the dK,dV grid must be visited exactly once, padding workgroups must be rejected, and the bundle order must put the
remainder bundle first and run slot-major inside a bundle (what the schedule's balance depends on)."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r"""
#include <cstdio>
#include <vector>
#include "attn_grid.h"
int main() {
    long cases = 0;
    for (int batch = 1; batch <= 9; batch += (batch < 3 ? 1 : 3))
        for (int kv_heads : {1, 2, 3, 8, 32})
            for (int group : {1, 4})
                for (int nblk : {1, 2, 7, 13})
                    for (int lpt : {1, 2, 8, 1000}) {
                        AttnGrid ag;
                        ag.heads = kv_heads * group; ag.kv_heads = kv_heads; ag.group = group; ag.nblk = nblk;
                        ag.n_kvp = batch * kv_heads; ag.epi = 1; ag.ctr = nullptr; ag.items = 0; ag.lpt = lpt;
                        // forward / dQ: one workgroup per (batch, query head, block)
                        std::vector<int> seen((size_t)batch * ag.heads * nblk, 0);
                        const int n = ag.grid(false);
                        if (n % 8) { printf("grid not a multiple of 8\n"); return 1; }
                        int pad = 0;
                        for (int L = 0; L < n; ++L) {
                            int head, kvhead, b, slot;
                            if (!ag.decode(L, head, kvhead, b, slot)) { ++pad; continue; }
                            if (b < 0 || b >= batch || head < 0 || head >= ag.heads || slot < 0 || slot >= nblk || kvhead != head / group) {
                                printf("decode out of range\n"); return 1;
                            }
                            if ((b * kv_heads + kvhead) % 8 != (L & 7)) { printf("K/V head on the wrong XCD\n"); return 1; }
                            ++seen[((size_t)b * ag.heads + head) * nblk + slot];
                        }
                        for (int v : seen) if (v != 1) { printf("decode: block visited %d times (batch %d kv %d group %d nblk %d lpt %d)\n", v, batch, kv_heads, group, nblk, lpt); return 1; }
                        if (pad != n - batch * ag.heads * nblk) { printf("padding count\n"); return 1; }
                        // dK,dV: one workgroup per (batch, K/V head, block)
                        std::vector<int> seen2((size_t)ag.n_kvp * nblk, 0);
                        const int n2 = ag.grid(true);
                        for (int L = 0; L < n2; ++L) {
                            int kvhead, b, slot;
                            if (!ag.decode_kv(L, kvhead, b, slot)) continue;
                            if (b < 0 || b >= batch || kvhead < 0 || kvhead >= kv_heads || slot < 0 || slot >= nblk) { printf("decode_kv out of range\n"); return 1; }
                            ++seen2[((size_t)b * kv_heads + kvhead) * nblk + slot];
                        }
                        for (int v : seen2) if (v != 1) { printf("decode_kv: block visited %d times\n", v); return 1; }
                        // order inside XCD 0: slots never decrease inside a bundle, the first bundle is the remainder
                        const int n_x = (ag.n_kvp + 7) / 8, G = lpt < n_x ? lpt : n_x;
                        const int first = (n_x % G) ? n_x % G : G;
                        int prev_slot = -1, prev_bundle = -1;
                        for (int s = 0; s < n2 / 8; ++s) {
                            int g0, gn, r;
                            ag.split(s, nblk, g0, gn, r);
                            const int bundle = g0 < first ? 0 : 1 + (g0 - first) / G;
                            if (gn != (bundle == 0 ? first : G)) { printf("bundle size\n"); return 1; }
                            if (bundle != prev_bundle) { if (bundle != prev_bundle + 1) { printf("bundle order\n"); return 1; } prev_bundle = bundle; prev_slot = -1; }
                            const int slot = r / gn;
                            if (slot < prev_slot) { printf("slot order inside a bundle\n"); return 1; }
                            prev_slot = slot;
                        }
                        ++cases;
                    }
    printf("ok %ld\n", cases);
    return 0;
}
"""


def test_attention_block_map_is_a_bijection_in_bundle_order():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "grid.cpp")
        exe = os.path.join(d, "grid")
        with open(src, "w") as f:
            f.write(PROGRAM)
        r = subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "vl-rlhf_amd", "csrc"), src, "-o", exe],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
        assert int(r.stdout.split()[1]) == 5 * 5 * 2 * 4 * 4
