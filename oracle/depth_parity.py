#!/usr/bin/env python3
"""Full-depth (LLaVA-1.5-7B, 32 decoder layers) runs of the CPU oracle.  TEST INFRASTRUCTURE - NOT THE PRODUCT.

    python oracle/depth_parity.py small      # 1 pair, T = 128 (S = 703), ragged: fp32 + bf16 error budget
    python oracle/depth_parity.py configs0   # BASELINE.json configs[0]: 4 pairs, T = 256 (S = 831): fp32 + bf16-emulated
    python oracle/depth_parity.py grads | grads_sharp   # fp32 autograd of the `small` case (VLR_DEPTH_LAYERS=2|32); _sharp: q / k weights x 2
    python oracle/depth_parity.py outliers   # (round 6) the `small` case on one hashed model with planted outlier channels: fp32 + the floor model
    python oracle/depth_parity.py seeds      # (round 4) the `small` case under 16 (round 4: 8) hashed models / batches: fp32 + the floor model

The weights are the machine-independent hashed weights of oracle.llava_dpo_oracle.HashedWeights (reference = seed 0,
policy = reference + 1e-3 * n'), so the GPU test regenerates the SAME 7B model on the MI355X from (seed, name) alone and
compares its loss / log-probs with the numbers written here (tests/golden/llava7b_depth32_<case>.json) - the CPU side
of the comparison takes minutes to hours on host cores and therefore runs offline, here, once.

Error budget (`small`): the fp32 oracle (weights are bf16-representable in every variant, like a bf16 checkpoint run in
fp32) against variants that round ONE group of tensors to bf16, showing which rounding points of a bf16 pipeline move
the DPO loss by more than north_star's rtol = 1e-3.  Written to profiles/r02_bf16_error_budget.txt.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import llava_dpo_oracle as O  # noqa: E402

CASES = {
    "small": dict(pairs=1, text_len=128, seed=11, ragged=True),
    "configs0": dict(pairs=4, text_len=256, seed=12, ragged=True),
}
ALL = frozenset(("w", "vit", "x0", "xn", "qkv", "v", "rope", "p", "attn", "resid", "gu", "act", "hidden"))
MFMA_OPERANDS = frozenset(("w", "xn", "rope", "p", "attn", "act", "hidden"))      # what any bf16-MFMA pipeline must round
VARIANTS = [
    ("fp32", False, "reference-exact arithmetic (bf16-representable weights)"),
    ("bf16_emulated", True, "every tensor the HIP path stores as bf16 (the oracle's emulate_bf16=True mode)"),
    ("all+p", ALL, "the above + softmax probabilities rounded before P.V"),
    ("only_hidden", frozenset(("hidden",)), "final-norm output (A operand of the lm-head GEMM) only"),
    ("only_resid", frozenset(("resid", "x0")), "residual stream only"),
    ("only_gemm_out", frozenset(("qkv", "gu")), "q/k/v and gate/up GEMM outputs only (storage choice)"),
    ("only_vit", frozenset(("vit",)), "vision tower + projector only"),
    ("mfma_operands", MFMA_OPERANDS, "only the operands of the MFMAs (xn, rope'd q/k, P, attn, act, hidden): floor of ANY bf16-MFMA path"),
    ("hip_f32resid", MFMA_OPERANDS | {"vit", "v"}, "what the HIP path rounds with the fp32 residual stream (round 3): MFMA operands + v + the vision tower"),
    ("f32resid-hidden", (MFMA_OPERANDS | {"vit", "v"}) - {"hidden"}, "hip_f32resid with an unrounded hidden state into the lm-head"),
    ("f32resid-vit", MFMA_OPERANDS | {"v"}, "hip_f32resid with an exact vision tower + projector"),
    ("f32resid+vit_f32", MFMA_OPERANDS | {"v", "vit_op"}, "hip_f32resid with an fp32 residual stream inside the ViT and an fp32 projector output (operands still bf16)"),
    ("f32resid+vit_f32resid", MFMA_OPERANDS | {"v", "vit_op", "vit_out"}, "... fp32 residual stream inside the ViT only (projector output rounded)"),
    ("f32resid+vit_f32out", MFMA_OPERANDS | {"v", "vit_op", "vit_resid"}, "... fp32 projector output only (bf16 residual stream inside the ViT)"),
    ("only_xn", frozenset(("xn",)), "RMSNorm outputs (A operand of the q|k|v and gate|up GEMMs) only"),
    ("only_ropev", frozenset(("rope", "v")), "q / k after RoPE and v (the attention operands) only"),
    ("only_p", frozenset(("p",)), "softmax probabilities (A operand of P.V) only"),
    ("only_attn", frozenset(("attn",)), "attention output (A operand of o_proj) only"),
    ("only_act", frozenset(("act",)), "silu(gate) * up (A operand of down_proj) only"),
    ("mfma_operands-hidden", MFMA_OPERANDS - {"hidden"}, "the floor if the lm-head consumed an unrounded hidden state"),
]


PROBE_FEATURES = 32


class _Probe:
    """list-like `collect` sink of llama_hidden: keeps, per decoder layer, x_out[sequence 0 and the last sequence,
    positions {0, S//2, S-1}, first 32 features] - enough to tell a mis-wired layer from bf16 noise, small enough to commit."""

    def __init__(self):
        self.rows = []

    def append(self, x):
        S = x.shape[1]
        self.rows.append(x[[0, -1]][:, [0, S // 2, S - 1], :PROBE_FEATURES].reshape(-1).tolist())


def run(case, variants, cfg, log):
    spec = CASES[case]
    batch = O.synthetic_batch(spec["pairs"], spec["text_len"], cfg["image_token"], 32000, cfg["image_size"], spec["seed"],
                              ragged=spec["ragged"])
    Wr = O.HashedWeights(cfg, seed=0, cache=True)
    Wp = O.HashedWeights(cfg, seed=0, delta=1e-3, seed_delta=1, cache=True)
    out = {}
    for name, emu, what in variants:
        t0 = time.time()
        col = _Probe() if name in ("fp32", "bf16_emulated") else None
        with torch.no_grad():
            pc, pr, _, _ = O.concatenated_forward(Wp, cfg, batch, "sigmoid", emu, collect=col)
            rc, rr, _, _ = O.concatenated_forward(Wr, cfg, batch, "sigmoid", emu)
            losses, cr, rj = O.dpo_loss(pc, pr, rc, rr, 0.1)
        out[name] = dict(loss=float(losses.mean()), policy_chosen_logps=pc.tolist(), policy_rejected_logps=pr.tolist(),
                         reference_chosen_logps=rc.tolist(), reference_rejected_logps=rr.tolist(), what=what)
        if col is not None:
            out[name]["layer_probe"] = col.rows     # residual stream after every decoder layer of the POLICY pass (see _Probe)
        log(f"{case} {name}: loss {out[name]['loss']:.7f}  pc {pc.tolist()} pr {pr.tolist()} rc {rc.tolist()} rr {rr.tolist()}"
            f"  [{time.time() - t0:.0f} s]")
    return spec, out


GRAD_LAYERS = (0, 17, 31)
SHARP_QK = 2.0          # `grads_sharp`: scale of the q_proj / k_proj weights of the sharp-softmax fixture


class _Leaves:
    """HashedWeights with a few tensors replaced by autograd leaves (the weight gradients of the other ~280 tensors are never formed)"""

    def __init__(self, base, names):
        self.base = base
        self.leaves = {n: base[n].clone().requires_grad_(True) for n in names}

    def __getitem__(self, k):
        return self.leaves[k] if k in self.leaves else self.base[k]

    def __contains__(self, k):
        return k in self.base

    def keys(self):
        return self.base.keys()


SEED_SWEEP = int(os.environ.get("VLR_SEED_SWEEP", "16"))   # `seeds`: weight seeds 1 .. SEED_SWEEP (reference = seed s, policy delta seed 100 + s), batch seed 20 + s (round 4: 8; round 5: 16)
FLOOR_TAGS = ("f32resid+vit_f32out",)


def run_seeds(cfg, log):
    """The `small` case under SEED_SWEEP different hashed models and batches: fp32 and the floor model of the HIP path's rounding
    (oracle.HIP_ROUNDING) -> tests/golden/llava7b_depth<L>_seeds.json.  The GPU test (tests/test_hip_depth.py::test_depth32_seed_sweep)
    rebuilds each model and asserts that the SIGNED loss / log-prob errors of the HIP path average to zero within 2 sigma / sqrt(n) - the
    statistic a systematic offset cannot pass (a single draw at <= 3 sigma can hide a bias of a full sigma)."""
    L = cfg["layers"]
    path = os.path.join(ROOT, "tests", "golden", f"llava7b_depth{L}_seeds.json")
    done = json.load(open(path))["seeds"] if os.path.exists(path) else []
    floor = [v for v in VARIANTS if v[0] in FLOOR_TAGS][0]
    for s in range(1 + len(done), SEED_SWEEP + 1):
        spec = dict(CASES["small"], seed=20 + s)
        batch = O.synthetic_batch(spec["pairs"], spec["text_len"], cfg["image_token"], 32000, cfg["image_size"], spec["seed"], ragged=spec["ragged"])
        Wr = O.HashedWeights(cfg, seed=s, cache=True)
        Wp = O.HashedWeights(cfg, seed=s, delta=1e-3, seed_delta=100 + s, cache=True)
        rec = dict(seed=s, seed_delta=100 + s, spec=spec)
        for tag, emu in (("fp32", False), ("floor", floor[1])):
            t0 = time.time()
            with torch.no_grad():
                pc, pr, _, _ = O.concatenated_forward(Wp, cfg, batch, "sigmoid", emu)
                rc, rr, _, _ = O.concatenated_forward(Wr, cfg, batch, "sigmoid", emu)
                losses, _, _ = O.dpo_loss(pc, pr, rc, rr, 0.1)
            rec[tag] = dict(loss=float(losses.mean()), policy_chosen_logps=pc.tolist(), policy_rejected_logps=pr.tolist(),
                            reference_chosen_logps=rc.tolist(), reference_rejected_logps=rr.tolist())
            log(f"seeds L{L} s={s} {tag}: loss {rec[tag]['loss']:.7f} [{time.time() - t0:.0f} s]")
        rec["weight_probe"] = {k: Wp[k].double().sum().item() for k in ("language_model.model.layers.0.self_attn.q_proj.weight",
                                                                          f"language_model.model.layers.{L - 1}.mlp.down_proj.weight")}
        done.append(rec)
        with open(path, "w") as f:      # after every seed: the sweep can be interrupted and resumed
            json.dump(dict(layers=L, cfg="LLAVA_1_5_7B", beta=0.1, floor=FLOOR_TAGS[0],
                           weights="HashedWeights(seed=s) reference; policy delta=1e-3 seed_delta=100+s; batch seed 20+s", seeds=done), f)
        del Wr, Wp


OUTLIERS = dict(layers=[1, 2], channels=[7, 1415, 2533], scale=32.0)


class _StreamStats:
    """list-like `collect` sink of llama_hidden: per decoder layer, mean |x| of the planted channels and the rms over all features"""

    def __init__(self, channels):
        self.channels, self.rows = list(channels), []

    def append(self, x):
        x = x.reshape(-1, x.shape[-1]).float()
        self.rows.append(dict(mean_abs_planted=x[:, self.channels].abs().mean().item(), rms_all=x.pow(2).mean().sqrt().item()))


def run_outliers(cfg, log):
    """(round 6) the `small` case on ONE hashed model with planted outlier channels (HashedWeights(outliers=OUTLIERS): three features of the
    residual stream receive 32 x larger MLP outputs in layers 1 and 2 - the massive-activation channels of a trained checkpoint, which a
    std-0.02 random model does not have): fp32 and the floor model -> tests/golden/llava7b_depth<L>_outliers.json, compared by
    tests/test_hip_depth.py::test_depth32_planted_outlier_channels.  Also records how large the planted channels are in the stream."""
    L = cfg["layers"]
    s = 17
    spec = dict(CASES["small"], seed=20 + s)
    batch = O.synthetic_batch(spec["pairs"], spec["text_len"], cfg["image_token"], 32000, cfg["image_size"], spec["seed"], ragged=spec["ragged"])
    Wr = O.HashedWeights(cfg, seed=s, cache=True, outliers=OUTLIERS)
    Wp = O.HashedWeights(cfg, seed=s, delta=1e-3, seed_delta=100 + s, cache=True, outliers=OUTLIERS)
    floor = [v for v in VARIANTS if v[0] in FLOOR_TAGS][0]
    rec = dict(seed=s, seed_delta=100 + s, spec=spec, outliers=OUTLIERS)
    for tag, emu in (("fp32", False), ("floor", floor[1])):
        t0 = time.time()
        col = _StreamStats(OUTLIERS["channels"]) if tag == "fp32" else None
        with torch.no_grad():
            pc, pr, _, _ = O.concatenated_forward(Wp, cfg, batch, "sigmoid", emu, collect=col)
            rc, rr, _, _ = O.concatenated_forward(Wr, cfg, batch, "sigmoid", emu)
            losses, _, _ = O.dpo_loss(pc, pr, rc, rr, 0.1)
        rec[tag] = dict(loss=float(losses.mean()), policy_chosen_logps=pc.tolist(), policy_rejected_logps=pr.tolist(),
                        reference_chosen_logps=rc.tolist(), reference_rejected_logps=rr.tolist())
        if col is not None:      # the residual stream behind every layer: |planted channels| against the rms of all features
            rec["stream"] = col.rows
        log(f"outliers L{L} {tag}: loss {rec[tag]['loss']:.7f} [{time.time() - t0:.0f} s]")
    rec["weight_probe"] = {k: Wp[k].double().sum().item() for k in ("language_model.model.layers.1.mlp.down_proj.weight",
                                                                      f"language_model.model.layers.{L - 1}.mlp.down_proj.weight")}
    path = os.path.join(ROOT, "tests", "golden", f"llava7b_depth{L}_outliers.json")
    with open(path, "w") as f:
        json.dump(dict(layers=L, cfg="LLAVA_1_5_7B", beta=0.1, floor=FLOOR_TAGS[0], record=rec), f)
    log(f"wrote {path}")


def run_grads(cfg, log, qk_scale=1.0):
    """fp32 gradients of the DPO loss of the `small` case w.r.t. every weight of decoder layers 0, 17, 31, the final norm and the
    lm-head: per tensor the Frobenius norm and a 256-element probe -> tests/golden/llava7b_depth<L>_small_grads.json
    (qk_scale != 1: the "sharp" fixture - q_proj / k_proj drawn qk_scale times larger, so that the softmax is peaked and dq, dk are
    first-order quantities whose DIRECTION can be asserted -> ..._sharp_grads.json)"""
    spec = CASES["small"]
    batch = O.synthetic_batch(spec["pairs"], spec["text_len"], cfg["image_token"], 32000, cfg["image_size"], spec["seed"], ragged=spec["ragged"])
    Wr = O.HashedWeights(cfg, seed=0, qk_scale=qk_scale)
    Wp = O.HashedWeights(cfg, seed=0, delta=1e-3, seed_delta=1, qk_scale=qk_scale)
    L = cfg["layers"]
    names = ["language_model.model.norm.weight", "language_model.lm_head.weight"]
    for l in sorted(set(min(x, L - 1) for x in GRAD_LAYERS)):
        p = f"language_model.model.layers.{l}."
        names += [p + n for n in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                                  "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight", "input_layernorm.weight",
                                  "post_attention_layernorm.weight")]
    t0 = time.time()
    WL = _Leaves(Wp, names)
    loss, _ = O.compute_loss(WL, Wr, cfg, batch, 0.1)
    log(f"grads: forward done, loss {float(loss):.7f} [{time.time() - t0:.0f} s]")
    loss.backward()
    out = {}
    for n in names:
        g = WL.leaves[n].grad
        flat = g.reshape(-1)
        stride = max(1, flat.numel() // 256)
        out[n] = dict(norm=float(g.double().norm()), stride=stride, probe=flat[::stride][:256].tolist())
    log(f"grads: backward done [{time.time() - t0:.0f} s]")
    path = os.path.join(ROOT, "tests", "golden", f"llava7b_depth{L}_{'small' if qk_scale == 1.0 else 'sharp'}_grads.json")
    with open(path, "w") as f:
        json.dump(dict(layers=L, spec=spec, loss=float(loss), qk_scale=qk_scale, grads=out), f)
    log(f"grads -> {path}")


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "small"
    if case in ("grads", "grads_sharp"):
        layers = int(os.environ.get("VLR_DEPTH_LAYERS", "32"))
        return run_grads(dict(O.LLAVA_1_5_7B, layers=layers), lambda s_: print(s_, flush=True), qk_scale=SHARP_QK if case == "grads_sharp" else 1.0)
    if case == "outliers":
        layers = int(os.environ.get("VLR_DEPTH_LAYERS", "32"))
        return run_outliers(dict(O.LLAVA_1_5_7B, layers=layers), lambda s_: print(s_, flush=True))
    if case == "seeds":
        layers = int(os.environ.get("VLR_DEPTH_LAYERS", "32"))
        return run_seeds(dict(O.LLAVA_1_5_7B, layers=layers), lambda s_: print(s_, flush=True))
    layers = int(os.environ.get("VLR_DEPTH_LAYERS", "32"))
    cfg = dict(O.LLAVA_1_5_7B, layers=layers)
    variants = VARIANTS if case == "small" else VARIANTS[:2]
    if os.environ.get("VLR_DEPTH_VARIANTS"):
        keep = os.environ["VLR_DEPTH_VARIANTS"].split(",")
        variants = [v for v in VARIANTS if v[0] in keep]
    logf = open(os.path.join(ROOT, "gpurun_out", f"depth_parity_{case}_L{layers}.log"), "a") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None

    def log(s):
        print(s, flush=True)
        if logf:
            logf.write(s + "\n")
            logf.flush()

    spec, out = run(case, variants, cfg, log)
    tag = f"llava7b_depth{layers}_{case}"
    # a checksum of the generated weights so the GPU side can prove it rebuilt the same model
    W = O.HashedWeights(cfg, seed=0, delta=1e-3, seed_delta=1)
    probe = {k: W[k].double().sum().item() for k in ("language_model.model.layers.0.self_attn.q_proj.weight",
                                                     f"language_model.model.layers.{layers - 1}.mlp.down_proj.weight",
                                                     "language_model.model.norm.weight")}
    gpath = os.path.join(ROOT, "tests", "golden", tag + ".json")
    if os.path.exists(gpath):          # partial re-runs (VLR_DEPTH_VARIANTS) update the file
        old = json.load(open(gpath))["results"]
        old.update(out)
        out = old
    with open(gpath, "w") as f:
        json.dump(dict(case=case, spec=spec, layers=layers, cfg="LLAVA_1_5_7B", weights="HashedWeights(seed=0) reference; policy delta=1e-3 seed_delta=1",
                       beta=0.1, weight_probe=probe, results=out), f)
    if case == "small" and len(out) > 2:
        ref = out["fp32"]["loss"]
        lines = [f"bf16 error budget of the DPO loss, LLaVA-1.5-7B widths, {layers} decoder layers, {spec['pairs']} pair(s), T={spec['text_len']} "
                 f"(S={spec['text_len'] - 1 + 576}), beta 0.1, sigmoid; CPU oracle (oracle/depth_parity.py)",
                 f"{'variant':24s} {'loss':>11s} {'|d| vs fp32':>12s} {'rel':>9s}   max |d logp|   what is rounded to bf16"]
        for name, r in out.items():
            d = abs(r["loss"] - ref)
            dls = [abs(a - b) for k in ("policy_chosen_logps", "policy_rejected_logps", "reference_chosen_logps", "reference_rejected_logps")
                   for a, b in zip(r[k], out["fp32"][k])]
            dl = max(dls)
            rms = (sum(x * x for x in dls) / len(dls)) ** 0.5
            lines.append(f"{name:24s} {r['loss']:11.7f} {d:12.3e} {d / abs(ref):9.2e}   {dl:12.4f} (rms {rms:.4f})   {r['what']}")
        with open(os.path.join(ROOT, "profiles", f"r03_bf16_error_budget_L{layers}.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
        log("\n".join(lines))


if __name__ == "__main__":
    main()
