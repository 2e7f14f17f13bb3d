#!/usr/bin/env python3
"""debug print of the 64-query forward kernel against fp32 torch at small shapes: VLR_ATTN_FWD3=1 python tools/attn_fwd3_debug.py"""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip
torch.manual_seed(0)
for (B, S, nh) in ((1, 64, 1), (1, 128, 1), (2, 200, 2), (1, 1599, 2)):
    hd = 128
    H = nh * hd
    qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).bfloat16()
    o = torch.full((B * S, H), 7.0, dtype=torch.bfloat16, device="cuda")
    Sp = (S + 63) // 64 * 64
    lse = torch.full((B, nh, Sp), 3.0, device="cuda")
    sc = 1 / math.sqrt(hd)
    _hip.call("vlr_attn_fwd", qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, o, H, lse, None, B, S, nh, hd, 1, sc)
    torch.cuda.synchronize()
    q, k, v = (qkv[:, i * H:(i + 1) * H].float().view(B, S, nh, hd).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) * sc
    s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, H)
    lref = torch.logsumexp(s, -1) * math.log2(math.e)
    of = o.float()
    err = (of - ref).norm() / ref.norm()
    print(f"B={B} S={S} nh={nh}: rel err {float(err):.3e}  |o| {float(of.norm()):.3f} |ref| {float(ref.norm()):.3f}  nan {int(torch.isnan(of).sum())}  "
          f"sevens {int((of == 7.0).sum())}  lse err {float((lse[:, :, :S] - lref).abs().max()):.3e}")
    # per 32-row block error
    blk = [(float((of[i:i + 32] - ref[i:i + 32]).norm() / ref[i:i + 32].norm())) for i in range(0, min(S, 256), 32)]
    print("   per 32-query block (sequence 0, all heads):", " ".join(f"{x:.2e}" for x in blk))
    print("   o[0,:6]", of[0, :6].tolist(), "ref", ref[0, :6].tolist())
    print("   o[40,:6]", of[min(40, S - 1), :6].tolist(), "ref", ref[min(40, S - 1), :6].tolist())
