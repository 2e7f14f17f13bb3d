#!/usr/bin/env python3
"""debug aid: per-tile error map of a GEMM under each persistent schedule mode (vlr_gemm_set_sched)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd")):
    sys.path.insert(0, p)
import torch
from vlrlhf import _hip

M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (12792, 4096, 1024))]
dev = "cuda"
_hip.ensure_splitk_workspace(dev, force=True)
g = torch.Generator().manual_seed(1)
a = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(dev)
b = (torch.randn(N, K, generator=g) * 0.5).bfloat16().to(dev)
ref = a.float() @ b.float().t()
tm, tn = (M + 255) // 256, (N + 255) // 256
for mode in (0, 1, 2, 3):
    _hip.helper("vlr_gemm_set_sched", mode)
    c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    _hip.call("vlr_gemm_bf16", 0, a, b, c, None, None, M, N, K, K, K, N, 0, 0, 0, 0)
    torch.cuda.synchronize()
    d = (c.float() - ref).abs()
    d = torch.nan_to_num(d, nan=1e9)
    pad = torch.zeros(tm * 256, tn * 256, device=dev)
    pad[:M, :N] = d
    per = pad.view(tm, 256, tn, 256).amax(dim=(1, 3)) / float(ref.abs().max())
    bad = (per > 2e-2).nonzero().tolist()
    print(f"mode {mode}: max rel err {float(per.max()):.3e}, bad tiles {len(bad)} of {tm * tn}: {bad[:24]}")
    if bad:
        i, j = bad[0]
        blk = d[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256]
        rows = (blk.amax(dim=1) > 2e-2 * float(ref.abs().max())).nonzero().flatten().tolist()
        cols = (blk.amax(dim=0) > 2e-2 * float(ref.abs().max())).nonzero().flatten().tolist()
        print(f"   first bad tile ({i},{j}): bad rows {len(rows)} [{rows[:8]}...], bad cols {len(cols)} [{cols[:8]}...]; c sample {c[i*256, j*256:j*256+4].tolist()} ref {ref[i*256, j*256:j*256+4].tolist()}")
_hip.helper("vlr_gemm_set_sched", -1)
