#!/usr/bin/env python3
"""per-tile cycle anatomy of attn_fwd2_kernel from an ATTN_TRACE2 build (python vl-rlhf_amd/build_hip.py --define ATTN_TRACE2=1 --tag _tr2):
   VLR_LIB=vl-rlhf_amd/libvlr_hip_tr2.so python tools/attn_fwd2_trace.py [B S heads kv]
stamps of wave 0 of the workgroup that takes work item 0 (the heaviest block): 1 tile top | 2 behind wait + barrier | 3 behind the LDS-DMA issue |
4 behind K Q^T + row maximum (the first reader of the scores) | 5 behind the exponentials / row sums | 6 behind the P V MFMAs (acc consumed)"""
import ctypes, math, os, sys, torch, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip
B, S, nh, nkv = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 1599, 32, 32)
hd = 128
H, HK = nh * hd, nkv * hd
qkv = (torch.randn(B * S, H + 2 * HK, device="cuda") * 0.5).bfloat16()
o = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
Sp = (S + 63) // 64 * 64
lse = torch.zeros(B, nh, Sp, device="cuda")
sc = 1 / math.sqrt(hd)
for _ in range(20):      # back to back: the clock of a warm kernel
    _hip.call("vlr_attn_fwd_gqa", qkv, qkv[:, H:], qkv[:, H + HK:], H + 2 * HK, o, H, lse, None, B, S, nh, nkv, hd, 1, sc)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["VLR_LIB"])
buf = (ctypes.c_ulonglong * 4096)()
assert lib.vlr_attn_fwd2_trace(buf, 4096) == 0
ev = [(v >> 56, v & ((1 << 56) - 1)) for v in buf if v]
names = {(1, 2): "wait vmcnt(0) + barrier", (2, 3): "LDS-DMA issue (8 pieces)", (3, 4): "K Q^T (16 MFMAs) + masks + row maximum", (4, 5): "exponentials, row sums",
         (5, 6): "rescale test + P V (16 MFMAs)", (6, 1): "loop"}
acc = collections.defaultdict(list)
for (a, ta), (b, tb) in zip(ev, ev[1:]):
    if tb >= ta:
        acc[(a, b)].append(tb - ta)
print(f"B={B} S={S} heads={nh}/{nkv}: {len(ev)} stamps (s_memtime ticks; every stamp costs ~200 of them itself)")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"  {names.get(k, str(k)):44s} n={len(v):4d}  median {v2[len(v2) // 2]:7d}  mean {sum(v) / len(v):9.1f}  min {v2[0]:6d}  max {v2[-1]:7d}")
tops = [e for e in ev if e[0] == 1]
tiles = [tb - ta for (a, ta), (b, tb) in zip(tops, tops[1:]) if tb >= ta]
if tiles:
    t2 = sorted(tiles)
    print(f"  tile (top to top): n={len(tiles)} median {t2[len(t2) // 2]} mean {sum(tiles) / len(tiles):.1f}")
