#!/usr/bin/env python3
"""runs the decoder attention forward + backward a few times at the C2 shape (for rocprofv3 --pmc passes)"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip
B, S, nh, hd = 8, 1599, 32, 128
H = nh * hd
qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
o = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
Sp = (S + 63) // 64 * 64
lse = torch.zeros(B, nh, Sp, device="cuda")
do = (torch.randn(B * S, H, device="cuda") * 0.5).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.zeros_like(lse)
sc = 1 / math.sqrt(hd)
for _ in range(2):
    _hip.call("vlr_attn_fwd", qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, o, H, lse, None, B, S, nh, hd, 1, sc)
    _hip.call("vlr_attn_bwd", qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, o, do, H, lse, delta, None, dqkv, dqkv[:, H:], dqkv[:, 2 * H:], 3 * H, B, S, nh, hd, 1, sc)
torch.cuda.synchronize()
