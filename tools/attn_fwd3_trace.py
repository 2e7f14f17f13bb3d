#!/usr/bin/env python3
"""per-tile cycle anatomy of attn_fwd3_kernel from a F3_TRACE build (python vl-rlhf_amd/build_hip.py is not needed: build with
build_hip.build(defines=("F3_TRACE=1",), tag="_tr")):  VLR_ATTN_FWD3=1 VLR_LIB=.../libvlr_hip_tr.so python tools/attn_fwd3_trace.py [B S heads kv]"""
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
for _ in range(3):
    _hip.call("vlr_attn_fwd_gqa", qkv, qkv[:, H:], qkv[:, H + HK:], H + 2 * HK, o, H, lse, None, B, S, nh, nkv, hd, 1, sc)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["VLR_LIB"])
buf = (ctypes.c_ulonglong * 4096)()
assert lib.vlr_attn_fwd3_trace(buf, 4096) == 0
ev = [(v >> 56, v & ((1 << 56) - 1)) for v in buf if v]
# stamps: 9 loop start | 6 iteration top | 7 after barrier | 8 before tile | 1 tile entry | 2 after running max | 3 after phase 1 | 4 after phase 2 | 5 tile end
names = {(6, 7): "wait vmcnt + barrier", (7, 8): "DMA issue", (8, 1): "call", (1, 2): "frag prefetch + running max", (2, 3): "phase 1 (16 slots)",
         (3, 4): "phase 2 (16 slots)", (4, 5): "pair max", (5, 6): "loop"}
acc = collections.defaultdict(list)
for (a, ta), (b, tb) in zip(ev, ev[1:]):
    if tb >= ta:
        acc[(a, b)].append(tb - ta)
print(f"B={B} S={S} heads={nh}/{nkv}: {len(ev)} stamps of workgroup 0 / wave 0 (s_memtime ticks = shader cycles at 100 MHz? see below)")
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"  {names.get(k, str(k)):32s} n={len(v):4d}  median {v2[len(v2) // 2]:7d}  mean {sum(v) / len(v):9.1f}  min {v2[0]:6d}  max {v2[-1]:7d}")
tiles = [tb - ta for (a, ta), (b, tb) in zip([e for e in ev if e[0] == 6], [e for e in ev if e[0] == 6][1:])]
if tiles:
    t2 = sorted(tiles)
    print(f"  iteration (top to top): n={len(tiles)} median {t2[len(t2) // 2]} mean {sum(tiles) / len(tiles):.1f}")
