#!/usr/bin/env python3
"""times the decoder attention forward / backward: attn_time.py [reps [B S heads kv_heads]]   (default: the C2 shape 8 x 1599, 32 / 32)"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip
B, S, nh, hd = 8, 1599, 32, 128
nkv = nh
if len(sys.argv) > 5:
    B, S, nh, nkv = (int(x) for x in sys.argv[2:6])
H = nh * hd
HK = nkv * hd
qkv = (torch.randn(B * S, H + 2 * HK, device="cuda") * 0.5).bfloat16()
o = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
Sp = (S + 63) // 64 * 64
lse = torch.zeros(B, nh, Sp, device="cuda")
do = (torch.randn(B * S, H, device="cuda") * 0.5).bfloat16()
dqkv = torch.empty_like(qkv)
LD = H + 2 * HK
delta = torch.zeros_like(lse)
sc = 1 / math.sqrt(hd)
fwd = lambda: _hip.call("vlr_attn_fwd_gqa", qkv, qkv[:, H:], qkv[:, H + HK:], LD, o, H, lse, None, B, S, nh, nkv, hd, 1, sc)
bwd = lambda: _hip.call("vlr_attn_bwd_gqa", qkv, qkv[:, H:], qkv[:, H + HK:], LD, o, do, H, lse, delta, None, dqkv, dqkv[:, H:], dqkv[:, H + HK:], LD, B, S, nh, nkv, hd, 1, sc)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name, f, fl in (("fwd", fwd, 4.0), ("bwd", bwd, 10.0)):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / reps * 1e-3
    print(f"B={B} S={S} heads={nh}/{nkv} attn {name}: {t*1e3:.3f} ms {fl*S*S*nh*hd*B*0.5/t/1e12:.1f} TF/s")
