#!/usr/bin/env python3
"""Timing of the 128x128-tile GEMM path on the shapes it runs in the DPO step: the peeled last tile rows of the decoder GEMMs
(504 rows at M = 12792), and whole decoder GEMMs (256x256 rounds + peeled rows).  VLR_GEMM128P=0|2|3|4 selects the register-staged
kernel / the LDS-DMA ring depth (read once per process: run once per setting).

    VLR_GEMM128P=4 python tools/gemm128_bench.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from vlrlhf import _hip  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = "cuda"
    _hip.ensure_splitk_workspace(dev, force=True)
    H, I = 4096, 11008
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).bfloat16()      # noqa: E731
    print(f"VLR_GEMM128P={os.environ.get('VLR_GEMM128P', '(default)')}")
    print(f"{'case':44s} {'us':>9s} {'TF/s':>8s}")
    for M in (504, 12792):
        x, xi = rn(M, H), rn(M, I)
        dq, dgu = rn(M, 3 * H), rn(M, 2 * I)
        wqkv, wo, wgu, wdown = rn(3 * H, H), rn(H, H), rn(2 * I, H), rn(H, I)
        res = torch.randn(M, H, device=dev, generator=g)
        yf = torch.empty(M, H, device=dev)
        dx = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
        cases = {
            f"NT o_proj f32res   [{M},4096,4096]": (lambda: _hip.call("vlr_gemm_bf16_f32res", 0, x, wo, yf, res, M, H, H, H, H, H, H), 2.0 * M * H * H),
            f"NT down f32res     [{M},4096,11008]": (lambda: _hip.call("vlr_gemm_bf16_f32res", 0, xi, wdown, yf, res, M, H, I, I, I, H, H), 2.0 * M * H * I),
            f"NN dgrad o         [{M},4096,4096]": (lambda: _hip.call("vlr_gemm_bf16", 1, x, wo, dx, None, None, M, H, H, H, H, H, 0, 0, 0, 0), 2.0 * M * H * H),
            f"NN dgrad qkv       [{M},4096,12288]": (lambda: _hip.call("vlr_gemm_bf16", 1, dq, wqkv, dx, None, None, M, H, 3 * H, 3 * H, H, H, 0, 0, 0, 0), 2.0 * M * H * 3 * H),
            f"NN dgrad gate|up   [{M},4096,22016]": (lambda: _hip.call("vlr_gemm_bf16", 1, dgu, wgu, dx, None, None, M, H, 2 * I, 2 * I, H, H, 0, 0, 0, 0), 2.0 * M * H * 2 * I),
        }
        for name, (fn, fl) in cases.items():
            us = timeit(fn)
            print(f"{name:44s} {us:9.1f} {fl / us / 1e6:8.1f}")
    # TN: a wgrad with few 256-tiles (vision projector / small layers) and the LoRA dB shape
    M = 12792
    dy, x = rn(M, H), rn(M, 1024)
    gw = torch.empty(H, 1024, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: _hip.call("vlr_gemm_bf16", 2, dy, x, gw, None, None, H, 1024, M, H, 1024, 1024, 0, 0, 0, 0))
    print(f"{'TN wgrad           [4096,1024,12792]':44s} {us:9.1f} {2.0 * M * H * 1024 / us / 1e6:8.1f}")
    u = rn(M, 128)
    gb = torch.empty(H, 128, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: _hip.call("vlr_gemm_bf16", 2, dy, u, gb, None, None, H, 128, M, H, 128, 128, 0, 0, 0, 0))
    print(f"{'TN dB = dy^T u     [4096,128,12792]':44s} {us:9.1f} {2.0 * M * H * 128 / us / 1e6:8.1f}")


if __name__ == "__main__":
    main()
