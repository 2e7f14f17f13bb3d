"""The persistent GEMM's tile order (vl-rlhf_amd/csrc/gemm_tilemap.h, sched bit 5) is plain integer code shared by host and device:
compiled here with g++ and enumerated.  Every output tile must be visited exactly once for any grid of tiles (ragged bands, ragged
super-columns), and the order must have the locality the map exists for: the run of gridDim / 8 tiles an XCD computes in a round spans
about 12 row + column panels (what its 4 MB L2 shares), and the 8 runs of a round together about 40 (what crosses the fabric per round
and has to come out of the 256 MB Infinity Cache instead of HBM) - against up to 96 for eight unrelated 8 x 4 blocks (the map of rounds 1-4)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r"""
#include <cstdio>
#include <set>
#include <vector>
#include "gemm_tilemap.h"
int main() {
    // 1. bijective on every grid
    for (int rot = 0; rot < 2; ++rot)      // both forms of the map: rows rotating with the super-column (unequal tiles) or not
    for (int tm = 1; tm <= 90; ++tm)
        for (int tn = 1; tn <= 90; ++tn) {
            std::vector<char> seen((size_t)tm * tn, 0);
            for (int pl = 0; pl < tm * tn; ++pl) {
                int r = -1, c = -1;
                vlr_tile_of_shared(pl, tm, tn, &r, &c, rot);
                if (r < 0 || r >= tm || c < 0 || c >= tn) { printf("tile %d of %d x %d out of range: (%d, %d)\n", pl, tm, tn, r, c); return 1; }
                if (seen[(size_t)r * tn + c]++) { printf("tile (%d, %d) of %d x %d visited twice\n", r, c, tm, tn); return 1; }
            }
        }
    // 2. locality on the tile grids of the LLaVA-1.5-7B step (M = 12792 token rows): panels per XCD run and per round of 256 workgroups
    const int shapes[][2] = {{50, 48}, {48, 16}, {50, 86}, {50, 43}, {85, 16}, {16, 43}, {16, 16}, {48, 48}};
    for (auto& sh : shapes) {
        const int tm = sh[0], tn = sh[1], n = tm * tn, G = 256, G8 = 32;
        double xs = 0, rs = 0;
        int nx = 0, nr = 0, xmax = 0;
        for (int i = 0; i * G < n; ++i) {
            std::set<int> rr, rc;
            for (int x = 0; x < 8; ++x) {
                std::set<int> xr, xc;
                for (int L = i * G + x * G8; L < i * G + (x + 1) * G8 && L < n; ++L) {
                    int r, c;
                    vlr_tile_of_shared(L, tm, tn, &r, &c);
                    xr.insert(r); xc.insert(c); rr.insert(r); rc.insert(c);
                }
                if (!xr.empty()) { const int v = (int)(xr.size() + xc.size()); xs += v; ++nx; if (v > xmax) xmax = v; }
            }
            rs += rr.size() + rc.size(); ++nr;
        }
        printf("%d x %d: panels per XCD run %.1f (max %d), per round %.1f\n", tm, tn, xs / nx, xmax, rs / nr);
        if (xs / nx > 13.0 || xmax > 24 || rs / nr > 48.0) { printf("locality lost\n"); return 1; }
    }
    printf("OK\n");
    return 0;
}
"""


def test_shared_panel_tile_map_is_bijective_and_local():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        with open(src, "w") as f:
            f.write(PROGRAM)
        exe = os.path.join(d, "t")
        subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "vl-rlhf_amd", "csrc"), src, "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
