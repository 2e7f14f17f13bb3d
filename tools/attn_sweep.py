#!/usr/bin/env python3
"""attention forward / backward time against sequence length at constant work (B * S^2 fixed), causal, 32 heads x 128:
a per-workgroup fixed cost shows as TF/s falling with S.  attn_sweep.py [reps]  (env VLR_ATTN_EPI=0/1 selects the epilogue)"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip
nh, hd = 32, 128
H = nh * hd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc = 1 / math.sqrt(hd)
shapes = [(max(1, round(8 * 1599 * 1599 / (S * S))), S) for S in (512, 1024, 1599, 2048, 3072, 4096, 8192)]
if os.environ.get("ATTN_SHAPES"):          # "B,S;B,S;..."
    shapes = [tuple(int(v) for v in p.split(",")) for p in os.environ["ATTN_SHAPES"].split(";")]
for B, S in shapes:
    qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
    o = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
    Sp = (S + 63) // 64 * 64
    lse = torch.zeros(B, nh, Sp, device="cuda")
    do = (torch.randn(B * S, H, device="cuda") * 0.5).bfloat16()
    dqkv = torch.empty_like(qkv)
    delta = torch.zeros_like(lse)
    for mk_name, mk in (("nomask", None), ("mask", torch.ones(B, S, dtype=torch.int32, device="cuda"))):
        fwd = lambda: _hip.call("vlr_attn_fwd", qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, o, H, lse, mk, B, S, nh, hd, 1, sc)
        bwd = lambda: _hip.call("vlr_attn_bwd", qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, o, do, H, lse, delta, mk, dqkv, dqkv[:, H:], dqkv[:, 2 * H:], 3 * H, B, S, nh, hd, 1, sc)
        out = []
        for name, f, fl in (("fwd", fwd, 4.0), ("bwd", bwd, 10.0)):
            for _ in range(2): f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps): f()
            e.record(); torch.cuda.synchronize()
            t = s.elapsed_time(e) / reps * 1e-3
            out.append(f"{name} {t*1e3:7.3f} ms {fl*S*S*nh*hd*B*0.5/t/1e12:6.1f} TF/s")
        print(f"B={B:3d} S={S:5d} {mk_name:6s} " + " | ".join(out), flush=True)
    del qkv, o, lse, do, dqkv, delta
