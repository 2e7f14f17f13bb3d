#!/usr/bin/env python3
"""List-scheduling model of the attention grids (no GPU): the hardware hands workgroups to free CU slots in index order, so the block
ORDER is the schedule.  Prints, per shape, the balanced time, the makespan of K/V-head-major order, of bundles of G heads (slot-major
inside a bundle, remainder bundle first - AttnGrid::split) and of slot-major order over all heads, in units of (a + b * tiles) with the
per-block cost a and the per-tile cost b fitted from tools/attn_sweep.py.  attn_schedule_sim.py [a] [b]"""
import heapq
import sys


def makespan(weights, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for w in weights:
        heapq.heappush(h, heapq.heappop(h) + w)
    return max(h)


def fwd_tiles(S):      # causal forward / dQ: block j (128 queries) sees ceil(min(S, 128 j + 128) / 64) key tiles; heaviest first
    return sorted(((min(S, 128 * j + 128) + 63) // 64 for j in range((S + 127) // 128)), reverse=True)


def dkv_tiles(S):      # causal dK,dV: key block j sees the query tiles from 128 j on
    return [(S - (j * 128 // 64) * 64 + 63) // 64 for j in range((S + 127) // 128)]


def order(tiles, n_x, G, a, b):
    first = n_x % G or G
    seq = []
    for n in [first] + [G] * ((n_x - first) // G):
        seq += [a + b * t for t in tiles for _ in range(n)]
    return seq


if __name__ == "__main__":
    a = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    b = float(sys.argv[2]) if len(sys.argv) > 2 else 2.3
    print(f"per-block cost {a}, per-tile cost {b} (us); 32 heads, 8 XCDs")
    for name, tiles_of, slots in (("forward / dQ (2 workgroups per CU: 64 slots per XCD)", fwd_tiles, 64),
                                  ("dK,dV (1 workgroup per CU: 32 slots per XCD)", dkv_tiles, 32)):
        print(name)
        for S, B in ((512, 78), (831, 8), (1024, 20), (1599, 8), (2048, 5), (4096, 1), (4975, 4)):
            t = tiles_of(S)
            n_x = (B * 32 + 7) // 8
            ideal = sum(a + b * x for x in t) * n_x / slots
            row = [f"G={G if G < n_x else 'all'}: {makespan(order(t, n_x, min(G, n_x), a, b), slots) / ideal:.3f}" for G in (1, 4, 8, 16, 10 ** 6)]
            print(f"  S={S:5d} batch={B:3d} heads/XCD={n_x:4d} balanced={ideal:8.1f} us | makespan / balanced: " + "  ".join(row))
