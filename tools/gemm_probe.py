#!/usr/bin/env python3
"""Runs one GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_probe.py layout M N K [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip  # noqa: E402

layout, M, N, K = (int(x) for x in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16() if layout != 2 else (torch.randn(K, M, device="cuda") * 0.5).bfloat16()
b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16() if layout == 0 else (torch.randn(K, N, device="cuda") * 0.5).bfloat16()
c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
lda = K if layout != 2 else M
ldb = K if layout == 0 else N
for _ in range(iters):
    _hip.call("vlr_gemm_bf16", layout, a, b, c, None, None, M, N, K, lda, ldb, N, 0, 0, 0, 0)
torch.cuda.synchronize()
