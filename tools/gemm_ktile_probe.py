#!/usr/bin/env python3
"""What does the steady-state K tile of the persistent 256x256 GEMM cost as a function of the operands' FOOTPRINT?  (round 5)
The NT launches of the 7B layer share one instruction stream, yet their K tile takes 1.30 us at [12792, 4096, 4096] and 1.6-1.9 us at the
larger shapes (tools/gemm_tile_trace.py).  This probe varies one thing at a time on the plain NT kernel - N (the weight's size), M (the
activation's size), K, and the row stride of A / B at constant bytes touched (address-space footprint: TLB reach / DRAM pages) - and
prints the timeline line of every case.  Diagnostics build: VLR_LIB=vl-rlhf_amd/libvlr_hip_trace.so VLR_GEMM_SPLIT=0 python tools/gemm_ktile_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from vlrlhf import _hip  # noqa: E402
from gemm_tile_trace import measure  # noqa: E402


def main():
    dev = "cuda"
    _hip.ensure_splitk_workspace(dev, force=True)
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()      # noqa: E731
    trace = torch.zeros(256 * 256, dtype=torch.int32, device=dev)
    only = sys.argv[1] if len(sys.argv) > 1 else ""

    def nt(M, N, K, lda=None, ldb=None, layout=0):
        lda, ldb = lda or K, ldb or K
        a = rn(M, lda)
        b = rn(N, ldb)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        return lambda: _hip.call("vlr_gemm_bf16", layout, a, b, c, None, None, M, N, K, lda, ldb, N, 0, 0, 0, 0)

    def nn(M, N, K):      # dgrad form: A [M][K], B [K][N]
        a, b = rn(M, K), rn(K, N)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        return lambda: _hip.call("vlr_gemm_bf16", 1, a, b, c, None, None, M, N, K, K, N, N, 0, 0, 0, 0)

    cases = [
        ("NT [12288,4096,4096] base", lambda: nt(12288, 4096, 4096)),
        ("NT [12288,8192,4096]  N x2", lambda: nt(12288, 8192, 4096)),
        ("NT [12288,12288,4096] N x3", lambda: nt(12288, 12288, 4096)),
        ("NT [12288,22016,4096] N x5.4", lambda: nt(12288, 22016, 4096)),
        ("NT [24576,4096,4096]  M x2", lambda: nt(24576, 4096, 4096)),
        ("NT [49152,4096,4096]  M x4", lambda: nt(49152, 4096, 4096)),
        ("NT [6144,4096,4096]   M /2", lambda: nt(6144, 4096, 4096)),
        ("NT [12288,4096,4096] lda=22016", lambda: nt(12288, 4096, 4096, lda=22016)),
        ("NT [12288,4096,4096] ldb=22016", lambda: nt(12288, 4096, 4096, ldb=22016)),
        ("NT [12288,4096,4096] lda=ldb=22016", lambda: nt(12288, 4096, 4096, lda=22016, ldb=22016)),
        ("NT [12288,4096,11008] K x2.7", lambda: nt(12288, 4096, 11008)),
        ("NT [12288,4096,22016] K x5.4", lambda: nt(12288, 4096, 22016)),
        ("NT [12288,4096,1024]  K /4", lambda: nt(12288, 4096, 1024)),
        ("NN [12288,4096,4096]", lambda: nn(12288, 4096, 4096)),
        ("NN [12288,4096,12288]", lambda: nn(12288, 4096, 12288)),
        ("NN [12288,11008,4096]", lambda: nn(12288, 11008, 4096)),
    ]
    warm = int(os.environ.get("PROBE_WARM", "3"))      # back-to-back launches in front of the traced one (sustained-load clock)
    for name, mk in cases:
        if only and only not in name:
            continue
        fn = mk()
        measure(name, fn, trace, warm=warm)
        del fn
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
