#!/usr/bin/env python3
"""Tile timeline of the persistent 256x256 GEMM launches (diagnostics build of the library: `python vl-rlhf_amd/build_hip.py --trace`,
run with VLR_LIB=vl-rlhf_amd/libvlr_hip_trace.so).  Wave 0 of every workgroup stamps the 100 MHz clock after the first K tile, after
the K loop and after the epilogue's stores were issued, per tile (vlr_gemm_set_trace).  Printed per GEMM shape of the 7B layer:

  k-tile   steady-state time of one K tile (us), from (K loop end - first K tile end) / (nt - 1)
  first    time of the FIRST K tile of a tile, measured from the end of the previous tile's epilogue: what the counted wait behind the
           epilogue's stores costs (a K tile that takes 5 us instead of 1.4 is a 3.6 us stall)
  epi      last MFMA -> last epilogue store issued (us)
  tile     whole tile (us) and the share of it that is not steady-state K loop
  phase    spread (max - min, us) over the 256 workgroups of the time their r-th tile ends: are the epilogue bursts in phase?

    VLR_LIB=vl-rlhf_amd/libvlr_hip_trace.so python tools/gemm_tile_trace.py [--M 12792] [--delay_us 0]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vlrlhf import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=12792)
    ap.add_argument("--only", default="")
    ap.add_argument("--dump", default="")
    a = ap.parse_args()
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
    qkv, gu, act = (torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16), torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16),
                    torch.empty(M, I, device=dev, dtype=torch.bfloat16))
    dx = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    gq, go, ggu, gd = torch.empty_like(wqkv), torch.empty_like(wo), torch.empty_like(wgu), torch.empty_like(wdown)
    pos = torch.arange(M, device=dev, dtype=torch.int32) % 1599
    cos, sin = torch.empty(4096, 64, device=dev), torch.empty(4096, 64, device=dev)
    _hip.call("vlr_rope_table", cos, sin, 4096, 128, 10000.0)
    dws = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
    cases = {
        "qkv+rope NT  [M,12288,4096]": lambda: _hip.call("vlr_gemm_qkv_rope", x, wqkv, qkv, pos, cos, sin, M, 3 * H, 2 * H, H, H, 128, 4096),
        "o_proj f32res NT [M,4096,4096]": lambda: _hip.call("vlr_gemm_bf16_f32res", 0, x, wo, yf, res, M, H, H, H, H, H, H),
        "swiglu NT   [M,22016,4096]": lambda: _hip.call("vlr_gemm_swiglu", x, wgu, gu, act, M, I, H, H, 1),
        "swiglu NT no gu store (ref pass)": lambda: _hip.call("vlr_gemm_swiglu", x, wgu, gu, act, M, I, H, H, 0),
        "down f32res NT [M,4096,11008]": lambda: _hip.call("vlr_gemm_bf16_f32res", 0, xi, wdown, yf, res, M, H, I, I, I, H, H),
        "dgrad qkv NN [M,4096,12288]": lambda: _hip.call("vlr_gemm_bf16", 1, dqkv, wqkv, dx, None, None, M, H, 3 * H, 3 * H, H, H, 0, 0, 0, 0),
        "swiglu-bwd NN [M,11008,4096]": lambda: _hip.call("vlr_gemm_swiglu_bwd", dyH, wdown, gu, dws, M, I, H),
        "dattn NN    [M,4096,4096]": lambda: _hip.call("vlr_gemm_bf16", 1, dyH, wo, dx, None, None, M, H, H, H, H, H, 0, 0, 0, 0),
        "wgrad qkv TN [12288,4096,M]": lambda: _hip.call("vlr_gemm_bf16", 2, dqkv, x, gq, None, None, 3 * H, H, M, 3 * H, H, H, 0, 0, 0, 0),
        "wgrad gu TN  [22016,4096,M]": lambda: _hip.call("vlr_gemm_bf16", 2, dgu, x, ggu, None, None, 2 * I, H, M, 2 * I, H, H, 0, 0, 0, 0),
    }
    trace = torch.zeros(256 * 256, dtype=torch.int32, device=dev)
    for name, fn in cases.items():
        if a.only and a.only not in name:
            continue
        measure(name, fn, trace, a.dump)


def measure(name, fn, trace, dump="", warm=3):
    """one traced launch of `fn` (after `warm` warm-up calls) -> the timeline line of the module docstring"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    trace.zero_()
    assert _hip.helper("vlr_gemm_set_trace", trace.data_ptr(), trace.numel() * 4) == 0, _hip.lib().vlr_last_error()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    _hip.helper("vlr_gemm_set_trace", None, 0)
    t = trace.cpu().numpy().astype(np.uint32).reshape(256, 64, 4)
    if dump:
        np.save(os.path.join(dump, name.split()[0].replace("+", "_") + "_" + name.split()[1] + ".npy"), t)
    npc = t[:, 63, 0].astype(int)
    act_b = np.nonzero(npc)[0]
    if len(act_b) == 0:
        print(f"{name:34s} no persistent continuous-pipeline launch recorded")
        return
    nt = int(t[act_b[0], 63, 2])
    t0 = t[act_b, 63, 1].astype(np.int64)
    kt, first, epi, tile = [], [], [], []
    ends = {}
    for bi, b in enumerate(act_b):
        prev = int(t0[bi])
        for i in range(min(npc[b], 63)):
            f, k, ep = (int(v) for v in t[b, i, 1:4])
            if f == 0:      # a tile with fewer than 3 K tiles on the fast path: no first-tile stamp
                f = prev
            kt.append((k - f) / max(nt - 1, 1) / 100.0)
            first.append((f - prev) / 100.0)
            epi.append((ep - k) / 100.0)
            tile.append((ep - prev) / 100.0)
            ends.setdefault(i, []).append(ep - int(t0.min()))
            prev = ep
    kt, first, epi, tile = (np.array(v) for v in (kt, first, epi, tile))
    # tiles after the first of a workgroup (the first one has the cold prologue in its "first K tile")
    later = np.concatenate([np.arange(sum(min(npc[b], 63) for b in act_b[:bi]) + 1, sum(min(npc[b], 63) for b in act_b[:bi + 1])) for bi in range(len(act_b))]) if npc.max() > 1 else np.arange(0)
    fl = first[later] if len(later) else first
    phase = " ".join(f"{(max(v) - min(v)) / 100.0:.0f}" for i, v in sorted(ends.items()) if len(v) > 128)
    print(f"{name:34s} {s.elapsed_time(e) * 1e3:7.0f} us  nt {nt:3d}  k-tile {np.median(kt):5.2f} (p10 {np.percentile(kt, 10):4.2f} p90 {np.percentile(kt, 90):4.2f})  first (later tiles) med {np.median(fl):5.2f} p90 {np.percentile(fl, 90):5.2f}"
          f"  epi med {np.median(epi):5.2f} p90 {np.percentile(epi, 90):5.2f}  tile med {np.median(tile):6.1f} (overhead {100 * (1 - np.median(kt) * nt / np.median(tile)):.1f} %)"
          f"  rounds' end spread [{phase}] us", flush=True)


if __name__ == "__main__":
    main()
