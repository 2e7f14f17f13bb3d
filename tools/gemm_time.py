#!/usr/bin/env python3
"""times one GEMM shape: gemm_time.py layout M N K"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip
layout, M, N, K = (int(x) for x in sys.argv[1:5])
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16() if layout != 2 else (torch.randn(K, M, device="cuda") * 0.5).bfloat16()
b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16() if layout == 0 else (torch.randn(K, N, device="cuda") * 0.5).bfloat16()
c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
lda = K if layout != 2 else M
ldb = K if layout == 0 else N
f = lambda: _hip.call("vlr_gemm_bf16", layout, a, b, c, None, None, M, N, K, lda, ldb, N, 0, 0, 0, 0)
for _ in range(3): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 10 * 1e-3
print(f"ABL={os.environ.get('VLR_GEMM_ABLATE','0')} layout {layout} {M}x{N}x{K}: {t*1e3:.3f} ms {2*M*N*K/t/1e12:.1f} TF/s")
