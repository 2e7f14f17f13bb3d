#!/usr/bin/env python3
"""Per-call timing of the adapter (LoRA / PLoRA) work of one decoder layer against its algorithmic HBM bytes: the skinny GEMMs
(u = drop(x) A^T, v = dy B, dA, dB), the input-gradient term (dx += mask . (v A)) and the three projections that carry the adapter
as extra K tiles (against the same projection without it).

    python tools/lora_gemm_bench.py [--shape llava|internlm] [--iters 10] [--seg_ab]

llava:    M = 12792 rows, r = 128, q / k / v separately adapted, lora_dropout 0.05 (scripts/ddpo_llava.sh)
internlm: M = 13888 rows (4 pairs x S = 1736), r = 256, intermediate 14336, ONE adapter over wqkv (PLoRA), dropout 0.05
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from vlrlhf import _hip  # noqa: E402


REPS = 1      # --reps N: N back-to-back calls between one pair of events (no idle gap in front of a call: the clock the power manager grants
              # inside a training step, not the dip behind an idle gap - profiles/r05_gemm_ktile_cycles_and_clock.txt; launch latency overlapped)


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(REPS):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / REPS)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="llava")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--seg_ab", action="store_true", help="also time the adapter-carrying projections with the adapter K tiles on the general staging path")
    ap.add_argument("--M", type=int, default=None)
    ap.add_argument("--dx_ablate", action="store_true", help="extra rows: the streaming dx kernel without its dx read / without the mask")
    ap.add_argument("--hash", action="store_true", help="the masked kernels hash in the kernel (ABI v4 behaviour) instead of reading packed masks")
    ap.add_argument("--reps", type=int, default=1, help="back-to-back calls per timed interval (20: sustained clock, launch latency hidden)")
    a = ap.parse_args()
    global REPS
    REPS = max(1, a.reps)
    dev = "cuda"
    _hip.ensure_splitk_workspace(dev, force=True)
    H, I = 4096, 11008
    if a.shape == "llava":
        M, r, nq, hd = a.M or 12792, 128, 3, 128
    else:
        M, r, nq, hd, I = a.M or 13888, 256, 1, 128, 14336        # InternLM2-7B: intermediate 14336 (its grouped-query wqkv is 6144 wide; the qkv rows below keep 12288)
    p, seed, sc = 0.05, 1234, 2.0
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).bfloat16()      # noqa: E731
    bf = lambda *s: torch.empty(*s, device=dev, dtype=torch.bfloat16)                 # noqa: E731
    xH, xI = rn(M, H), rn(M, I)
    dyH, dq, dgu = rn(M, H), rn(M, 3 * H), rn(M, 2 * I)
    ldu = 7 * r
    u = rn(M, ldu)
    v = bf(M, 3 * r)
    dxH, dxI = rn(M, H), rn(M, I)
    scratch = bf(M, I)
    groups = {   # name: (n, in, [outs], x, dy, lddy, dx)
        "qkv": (nq, H, [H] * 3 if nq == 3 else [3 * H], xH, dq, 3 * H, dxH),
        "o": (1, H, [H], xH, dyH, H, dxH),
        "gate_up": (2, H, [I, I], xH, dgu, 2 * I, dxH),
        "down": (1, I, [H], xI, dyH, H, dxI),
    }
    rows = []

    def rec(name, us, nbytes, flops):
        rows.append((name, us, nbytes / us / 1e3, nbytes / 8e6, flops / us / 1e6))       # GB/s, floor us at 8 TB/s, TFLOP/s

    for gname, (n, din, outs, x, dy, lddy, dx) in groups.items():
        out = outs[0]
        A = rn(n * r, din)
        B = rn(sum(outs), r)
        dA, dB = bf(n * r, din), bf(sum(outs), r)
        nr = n * r
        gst = M * din // 8
        bits = None if a.hash else torch.zeros(n * gst, dtype=torch.uint8, device=dev)
        tst = _hip.helper("vlr_dropout_bits_kt_bytes", M, din)
        bits_kt = None if a.hash else torch.zeros(n * tst, dtype=torch.uint8, device=dev)
        if bits is not None:
            def draw():
                for t in range(n):
                    _hip.call("vlr_dropout_bits2", bits[t * gst:], bits_kt[t * tst:], M, din, p, seed + t)
            us = timeit(draw, a.iters)
            rec(f"{gname:8s} draw the packed masks x{n}", us, 1.0 * n * gst, 0.0)
        us = timeit(lambda: _hip.call("vlr_gemm_grouped_bits", 0, x, A, u, M, r, din, din, din, ldu, n, 0, r * din, r, sc / (1 - p), 0, 1, seed, p, din, bits, gst), a.iters)
        rec(f"{gname:8s} u = drop(x) A^T  [M,{nr},{din}]", us, 2.0 * M * din + 2.0 * M * nr, 2.0 * M * nr * din)
        if bits is not None:
            us = timeit(lambda: _hip.call("vlr_lora_rows_u", n, x, din, A, u, ldu, r, M, din, r, sc / (1 - p), bits, gst, None), a.iters)
            rec(f"{gname:8s} u, streaming row slabs (r06)", us, 2.0 * M * din + 2.0 * M * nr, 2.0 * M * nr * din)
        outs_h = torch.tensor(outs, dtype=torch.int32)
        us = timeit(lambda: _hip.call("vlr_lora_rows_v", n, dy, lddy, outs_h, B, v, nr, M, r, None), a.iters)
        rec(f"{gname:8s} v, streaming row slabs (r06)", us, 2.0 * M * out * n + 2.0 * M * nr, 2.0 * M * nr * out)
        us = timeit(lambda: _hip.call("vlr_gemm_grouped", 1, dy, B, v, M, r, out, lddy, r, nr, n, out, out * r, r, 1.0, 0, 0, 0, 0.0, 0), a.iters)
        rec(f"{gname:8s} v = dy B         [M,{nr},{out}]", us, 2.0 * M * out * n + 2.0 * M * nr, 2.0 * M * nr * out)
        us = timeit(lambda: _hip.call("vlr_gemm_grouped", 2, dy, u, dB, out, r, M, lddy, ldu, r, n, out, r, out * r, 1.0, 0, 0, 0, 0.0, 0), a.iters)
        rec(f"{gname:8s} dB = dy^T u      [{out * n},{r},M]", us, 2.0 * M * out * n + 2.0 * M * nr, 2.0 * M * nr * out)
        if bits is None:
            us = timeit(lambda: _hip.call("vlr_gemm_grouped_bits", 2, v, x, dA, r, din, M, nr, din, din, n, r, 0, r * din, sc / (1 - p), 0, 2, seed, p, din, None, 0), a.iters)
        else:
            us = timeit(lambda: _hip.call("vlr_gemm_grouped_bits", 2, v, x, dA, r, din, M, nr, din, din, n, r, 0, r * din, sc / (1 - p), 0, 3, seed, p, din, bits_kt, tst), a.iters)
        rec(f"{gname:8s} dA = v^T drop(x) [{nr},{din},M]", us, 2.0 * M * din + 2.0 * M * nr, 2.0 * M * nr * din)

        def dxall():
            for t in range(n):
                _hip.call("vlr_gemm_dropout_acc_bits", v[:, t * r:], nr, A[t * r:], dx, scratch, M, din, r, p, seed + t, sc, None if bits is None else bits[t * gst:])
        us = timeit(dxall, a.iters)
        rec(f"{gname:8s} dx += mask.(v A) x{n} [M,{din},{r}]", us, 4.0 * M * din + 2.0 * M * nr, 2.0 * M * nr * din)
        us = timeit(lambda: _hip.call("vlr_gemm_dropout_acc_multi_bits", n, v, nr, A, dx, M, din, r, p, seed, sc, 1, bits, gst), a.iters)
        rec(f"{gname:8s} dx (one pass, multi)", us, 4.0 * M * din + 2.0 * M * nr, 2.0 * M * nr * din)
        if a.dx_ablate:
            us = timeit(lambda: _hip.call("vlr_gemm_dropout_acc_multi_bits", n, v, nr, A, dx, M, din, r, p, seed, sc, 0, bits, gst), a.iters)
            rec(f"{gname:8s} dx multi, WRITE only (no dx read)", us, 2.0 * M * din, 0.0)
            us = timeit(lambda: _hip.call("vlr_gemm_dropout_acc_multi_bits", n, v, nr, A, dx, M, din, r, 0.0, seed, sc, 1, None, 0), a.iters)
            rec(f"{gname:8s} dx multi, no mask", us, 4.0 * M * din, 0.0)

    # the projections with / without the adapter K tiles
    wqkv, wo, wgu, wdown = rn(3 * H, H), rn(H, H), rn(2 * I, H), rn(H, I)
    bq, bo, bgu, bd = rn(3 * H, r), rn(H, r), rn(2 * I, r), rn(H, r)
    qkv, gu, act = bf(M, 3 * H), bf(M, 2 * I), bf(M, I)
    res = torch.randn(M, H, device=dev, generator=g)
    yf = torch.empty(M, H, device=dev)
    pos = torch.arange(M, device=dev, dtype=torch.int32) % 1599
    cos, sin = torch.empty(4096, 64, device=dev), torch.empty(4096, 64, device=dev)
    _hip.call("vlr_rope_table", cos, sin, 4096, hd, 10000.0)
    proj = {
        "qkv+rope": (lambda: _hip.call("vlr_gemm_qkv_rope", xH, wqkv, qkv, pos, cos, sin, M, 3 * H, 2 * H, H, H, hd, 4096),
                     lambda: _hip.call("vlr_gemm_qkv_rope_lora", xH, wqkv, None, qkv, pos, cos, sin, M, 3 * H, 2 * H, H, H, hd, 4096, u, ldu, bq, r,
                                       3 * H if nq == 1 else H, 0 if nq == 1 else H), 2.0 * M * 3 * H * H),
        "o_proj f32": (lambda: _hip.call("vlr_gemm_bf16_f32res", 0, xH, wo, yf, res, M, H, H, H, H, H, H),
                       lambda: _hip.call("vlr_gemm_lora_f32res", xH, H, wo, yf, H, res, H, M, H, H, u, ldu, bo, r), 2.0 * M * H * H),
        "swiglu": (lambda: _hip.call("vlr_gemm_swiglu", xH, wgu, gu, act, M, I, H, H, 1),
                   lambda: _hip.call("vlr_gemm_swiglu_lora", xH, wgu, gu, act, M, I, H, H, u, ldu, bgu, r), 2.0 * M * 2 * I * H),
        "down f32": (lambda: _hip.call("vlr_gemm_bf16_f32res", 0, xI, wdown, yf, res, M, H, I, I, I, H, H),
                     lambda: _hip.call("vlr_gemm_lora_f32res", xI, I, wdown, yf, H, res, H, M, H, I, u, ldu, bd, r), 2.0 * M * H * I),
    }
    print(f"shape {a.shape}: M = {M}, r = {r}, dropout {p}")
    print(f"{'call':52s} {'us':>8s} {'GB/s':>8s} {'floor us':>9s} {'x floor':>8s} {'TF/s':>7s}")
    tot = fl = 0.0
    for name, us, gbs, floor, tf in rows:
        print(f"{name:52s} {us:8.1f} {gbs:8.0f} {floor:9.1f} {us / floor:8.2f} {tf:7.1f}")
        if "multi" not in name and "r06" not in name:
            tot += us
            fl += floor
    print(f"{'sum (per layer, without the multi rows)':52s} {tot:8.1f} {'':8s} {fl:9.1f} {tot / fl:8.2f}")
    print()
    print(f"{'projection':14s} {'plain us':>9s} {'+adapter us':>12s} {'ratio':>6s}" + ("   general-path us  ratio" if a.seg_ab else ""))
    for name, (f0, f1, _) in proj.items():
        t0, t1 = timeit(f0, a.iters), timeit(f1, a.iters)
        line = f"{name:14s} {t0:9.1f} {t1:12.1f} {t1 / t0:6.3f}"
        if a.seg_ab:
            _hip.helper("vlr_gemm_set_sched", 8 | 32)
            t2 = timeit(f1, a.iters)
            _hip.helper("vlr_gemm_set_sched", -1)
            line += f"   {t2:15.1f} {t2 / t0:6.3f}"
        print(line)


if __name__ == "__main__":
    main()
