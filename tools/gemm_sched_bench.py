#!/usr/bin/env python3
"""Within-process A/B of the persistent GEMM's switches (vlr_gemm_set_sched: 0 default, 16 serial epilogue order; round 3 used it for the
stream-K / rotation schedules 1-3, since removed) on the GEMM shapes of the LLaVA-1.5-7B DPO step (M = 12792 token rows): interleaved
rounds, median and min per mode.  With one mode it is the per-layer GEMM microbenchmark (A/B two builds of the library through VLR_LIB).

    python tools/gemm_sched_bench.py [--rounds 5] [--modes 0,16]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from vlrlhf import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--modes", default="32", help="vlr_gemm_set_sched values to compare (32 = the production default since round 5: shared-panel tile map; 0 = the tile map of rounds 1-4)")
    ap.add_argument("--M", type=int, default=12792)
    a = ap.parse_args()
    modes = [int(m) for m in a.modes.split(",")]
    dev = "cuda"
    _hip.ensure_splitk_workspace(dev, force=True)
    M, H, I = a.M, 4096, 11008
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()      # noqa: E731
    x, xi, dyH = rn(M, H), rn(M, I), rn(M, H)
    wqkv, wo, wgu, wdown = rn(3 * H, H), rn(H, H), rn(2 * I, H), rn(H, I)
    dqkv, dgu = rn(M, 3 * H), rn(M, 2 * I)
    res = torch.randn(M, H, device=dev, generator=g)
    yf = torch.empty(M, H, device=dev)
    qkv, gu, act = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16), torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16), torch.empty(M, I, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    gq, go, ggu, gd = torch.empty_like(wqkv), torch.empty_like(wo), torch.empty_like(wgu), torch.empty_like(wdown)
    pos = torch.arange(M, device=dev, dtype=torch.int32) % 1599
    cos, sin = torch.empty(4096, 64, device=dev), torch.empty(4096, 64, device=dev)
    _hip.call("vlr_rope_table", cos, sin, 4096, 128, 10000.0)
    dws = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
    cases = {
        "qkv+rope NT  [M,12288,4096]": (lambda: _hip.call("vlr_gemm_qkv_rope", x, wqkv, qkv, pos, cos, sin, M, 3 * H, 2 * H, H, H, 128, 4096), 2.0 * M * 3 * H * H),
        "o_proj f32res NT [M,4096,4096]": (lambda: _hip.call("vlr_gemm_bf16_f32res", 0, x, wo, yf, res, M, H, H, H, H, H, H), 2.0 * M * H * H),
        "swiglu NT   [M,22016,4096]": (lambda: _hip.call("vlr_gemm_swiglu", x, wgu, gu, act, M, I, H, H, 1), 2.0 * M * 2 * I * H),
        "down f32res NT [M,4096,11008]": (lambda: _hip.call("vlr_gemm_bf16_f32res", 0, xi, wdown, yf, res, M, H, I, I, I, H, H), 2.0 * M * H * I),
        "dgrad qkv NN [M,4096,12288]": (lambda: _hip.call("vlr_gemm_bf16", 1, dqkv, wqkv, dx, None, None, M, H, 3 * H, 3 * H, H, H, 0, 0, 0, 0), 2.0 * M * H * 3 * H),
        "dgrad gu NN  [M,4096,22016]": (lambda: _hip.call("vlr_gemm_bf16", 1, dgu, wgu, dx, None, None, M, H, 2 * I, 2 * I, H, H, 0, 0, 0, 0), 2.0 * M * H * 2 * I),
        "swiglu-bwd NN [M,11008,4096]": (lambda: _hip.call("vlr_gemm_swiglu_bwd", dyH, wdown, gu, dws, M, I, H), 2.0 * M * I * H),
        "dattn NN    [M,4096,4096]": (lambda: _hip.call("vlr_gemm_bf16", 1, dyH, wo, dx, None, None, M, H, H, H, H, H, 0, 0, 0, 0), 2.0 * M * H * H),
        "wgrad qkv TN [12288,4096,M]": (lambda: _hip.call("vlr_gemm_bf16", 2, dqkv, x, gq, None, None, 3 * H, H, M, 3 * H, H, H, 0, 0, 0, 0), 2.0 * M * 3 * H * H),
        "wgrad o TN   [4096,4096,M]": (lambda: _hip.call("vlr_gemm_bf16", 2, dyH, x, go, None, None, H, H, M, H, H, H, 0, 0, 0, 0), 2.0 * M * H * H),
        "wgrad gu TN  [22016,4096,M]": (lambda: _hip.call("vlr_gemm_bf16", 2, dgu, x, ggu, None, None, 2 * I, H, M, 2 * I, H, H, 0, 0, 0, 0), 2.0 * M * 2 * I * H),
        "wgrad down TN [4096,11008,M]": (lambda: _hip.call("vlr_gemm_bf16", 2, dyH, xi, gd, None, None, H, I, M, H, I, I, 0, 0, 0, 0), 2.0 * M * H * I),
        "wgrad gu + down, one launch": (lambda: _hip.call("vlr_gemm_bf16_tn_pair", dgu, x, ggu, 2 * I, H, 2 * I, H, H, dyH, xi, gd, H, I, H, I, I, M, 0), 2.0 * M * 3 * I * H),
    }
    if not hasattr(_hip.lib(), "vlr_gemm_bf16_tn_pair"):      # an older build of the library through VLR_LIB
        del cases["wgrad gu + down, one launch"]
    times = {k: {m: [] for m in modes} for k in cases}
    for name, (fn, _) in cases.items():
        for m in modes:
            _hip.helper("vlr_gemm_set_sched", m)
            fn()
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for name, (fn, _) in cases.items():
            for m in modes:
                _hip.helper("vlr_gemm_set_sched", m)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.iters):
                    fn()
                e.record()
                e.synchronize()
                times[name][m].append(s.elapsed_time(e) / a.iters)
    _hip.helper("vlr_gemm_set_sched", -1)
    per_layer = {"qkv+rope NT  [M,12288,4096]": 2, "o_proj f32res NT [M,4096,4096]": 2, "swiglu NT   [M,22016,4096]": 2, "down f32res NT [M,4096,11008]": 2}
    if "wgrad gu + down, one launch" in cases:                # what the layer backward calls; the two single rows are then for comparison only
        per_layer.update({"wgrad gu TN  [22016,4096,M]": 0, "wgrad down TN [4096,11008,M]": 0})
    tot = {m: 0.0 for m in modes}
    print(f"{'shape':34s} " + " ".join(f"{'mode ' + str(m) + ' ms (TF/s)':>22s}" for m in modes))
    for name, (fn, fl) in cases.items():
        row = []
        for m in modes:
            ts = sorted(times[name][m])
            med = ts[len(ts) // 2]
            tot[m] += med * per_layer.get(name, 1)
            row.append(f"{med:8.4f} ({fl / med / 1e9:6.0f}) min {ts[0]:.4f}"[:22].rjust(22))
        print(f"{name:34s} " + " ".join(row))
    print("per decoder layer (policy fwd + reference fwd + bwd), GEMMs only: " + "  ".join(f"mode {m}: {tot[m]:.3f} ms (x32 = {32 * tot[m]:.1f})" for m in modes))


if __name__ == "__main__":
    main()
