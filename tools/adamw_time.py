#!/usr/bin/env python3
"""times vlr_adamw_step on n parameters (default 2e9): adamw_time.py [n]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000_000
n -= n % 8
master = torch.randn(n, device="cuda"); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
g = torch.randn(n, device="cuda").bfloat16(); p = master.bfloat16()
coef = torch.ones(3, device="cuda")
f = lambda i: _hip.call("vlr_adamw_step", master, m, v, g, p, n, 1e-6, 0.9, 0.999, 1e-8, 0.0, i, coef)
for i in range(1, 3): f(i)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(3, 13): f(i)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 10 * 1e-3
print(f"adamw {n/1e9:.2f} G params: {t*1e3:.3f} ms  {28*n/t/1e12:.2f} TB/s")
