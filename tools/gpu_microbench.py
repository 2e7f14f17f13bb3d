#!/usr/bin/env python3
"""Per-kernel timings at the LLaVA-1.5-7B / C2 shapes (M = 8 x 1599 rows).  Prints TFLOP/s or GB/s per kernel."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from vlrlhf import _hip  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def bf(*shape):
    return (torch.randn(*shape, device=DEV) * 0.5).bfloat16()


def main():
    M = 8 * 1599
    H, I, V = 4096, 11008, 32064
    print("device", torch.cuda.get_device_name(0), flush=True)
    for name, layout, (m, n, k) in [
        ("qkv  NT", 0, (M, 3 * H, H)), ("o    NT", 0, (M, H, H)), ("gu   NT", 0, (M, 2 * I, H)), ("down NT", 0, (M, H, I)),
        ("dact NN", 1, (M, I, H)), ("dxn2 NN", 1, (M, H, 2 * I)), ("dqkv NN", 1, (M, H, 3 * H)),
        ("dWdn TN", 2, (H, I, M)), ("dWgu TN", 2, (2 * I, H, M)), ("dWqkv TN", 2, (3 * H, H, M)), ("dWo  TN", 2, (H, H, M)),
        ("lmhd NT", 0, (4096, V, H)), ("sq4k NT", 0, (4096, 4096, 4096)), ("sq8k NT", 0, (8192, 8192, 8192)),
    ]:
        a = bf(m, k) if layout != 2 else bf(k, m)
        b = bf(n, k) if layout == 0 else bf(k, n)
        c = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
        lda = k if layout != 2 else m
        ldb = k if layout == 0 else n
        t = timeit(lambda: _hip.call("vlr_gemm_bf16", layout, a, b, c, None, None, m, n, k, lda, ldb, n, 0, 0, 0, 0))
        print(f"gemm {name} {m}x{n}x{k}: {t*1e3:8.3f} ms  {2*m*n*k/t/1e12:8.1f} TF/s", flush=True)
    B, S, nh, hd = 8, 1599, 32, 128
    qkv = bf(B * S, 3 * H)
    o = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    Sp = (S + 63) // 64 * 64
    lse = torch.zeros(B, nh, Sp, device=DEV)
    sc = 1 / math.sqrt(hd)
    t = timeit(lambda: _hip.call("vlr_attn_fwd", qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, o, H, lse, None, B, S, nh, hd, 1, sc))
    fl = 4 * S * S * H * B / 2
    print(f"attn fwd causal: {t*1e3:8.3f} ms  {fl/t/1e12:8.1f} TF/s (causal-half flops)", flush=True)
    do = bf(B * S, H)
    dqkv = torch.empty_like(qkv)
    delta = torch.zeros_like(lse)
    t = timeit(lambda: _hip.call("vlr_attn_bwd", qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, o, do, H, lse, delta, None, dqkv,
                                 dqkv[:, H:], dqkv[:, 2 * H:], 3 * H, B, S, nh, hd, 1, sc))
    print(f"attn bwd causal: {t*1e3:8.3f} ms  {2.5*fl/t/1e12:8.1f} TF/s (2.5x fwd flops)", flush=True)
    # ViT attention
    Bv, Sv, nhv, hdv = 4, 577, 16, 64
    Hv = nhv * hdv
    qv = bf(Bv * Sv, 3 * Hv)
    ov = torch.empty(Bv * Sv, Hv, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: _hip.call("vlr_attn_fwd", qv, qv[:, Hv:], qv[:, 2 * Hv:], 3 * Hv, ov, Hv, None, None, Bv, Sv, nhv, hdv, 0, 0.125))
    print(f"attn vit fwd: {t*1e3:8.3f} ms  {4*Sv*Sv*Hv*Bv/t/1e12:8.1f} TF/s", flush=True)
    # HBM-bound kernels
    x = bf(M, H)
    w = bf(H)
    y = torch.empty_like(x)
    rstd = torch.empty(M, device=DEV)
    t = timeit(lambda: _hip.call("vlr_rmsnorm_fwd", x, w, y, rstd, M, H, 1e-5))
    print(f"rmsnorm fwd: {t*1e6:8.1f} us  {2*M*H*2/t/1e9:8.1f} GB/s", flush=True)
    ws = torch.empty(_hip.helper("vlr_rmsnorm_bwd_workspace_bytes", H), dtype=torch.uint8, device=DEV)
    dw = torch.empty(H, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: _hip.call("vlr_rmsnorm_bwd", y, x, w, rstd, x, y, dw, 0, ws, M, H))
    print(f"rmsnorm bwd: {t*1e6:8.1f} us  {4*M*H*2/t/1e9:8.1f} GB/s", flush=True)
    gu = bf(M, 2 * I)
    act = torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: _hip.call("vlr_swiglu_fwd", gu, act, M, I))
    print(f"swiglu fwd: {t*1e6:8.1f} us  {3*M*I*2/t/1e9:8.1f} GB/s", flush=True)
    t = timeit(lambda: _hip.call("vlr_swiglu_bwd", gu, act, M, I))
    print(f"swiglu bwd: {t*1e6:8.1f} us  {5*M*I*2/t/1e9:8.1f} GB/s", flush=True)
    pos = torch.arange(S, device=DEV, dtype=torch.int32).repeat(B)
    cos_t = torch.empty(2048, 64, device=DEV)
    sin_t = torch.empty_like(cos_t)
    _hip.call("vlr_rope_table", cos_t, sin_t, 2048, 128, 10000.0)
    t = timeit(lambda: _hip.call("vlr_rope", qkv, pos, cos_t, sin_t, M, H, 128, 3 * H, 2048, 0))
    print(f"rope: {t*1e6:8.1f} us  {4*M*H*2/t/1e9:8.1f} GB/s", flush=True)
    n = 8 * 1024 * 1024 * 64   # 0.5 G params
    master, m_, v_ = (torch.zeros(n, device=DEV) for _ in range(3))
    g = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    p16 = torch.empty_like(g)
    t = timeit(lambda: _hip.call("vlr_adamw_step", master, m_, v_, g, p16, n, 1e-5, 0.9, 0.98, 1e-6, 0.0, 1, None), iters=5)
    print(f"adamw: {t*1e3:8.3f} ms  {28*n/t/1e9:8.1f} GB/s", flush=True)
    logits = torch.randn(4096, V, device=DEV)
    tgt = torch.randint(0, V, (4096,), device=DEV, dtype=torch.int32)
    tok, ls = torch.empty(4096, device=DEV), torch.empty(4096, device=DEV)
    t = timeit(lambda: _hip.call("vlr_logp_rows", logits, None, tgt, 4096, V, V, tok, ls))
    print(f"logp_rows: {t*1e6:8.1f} us  {4096*V*4/t/1e9:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
