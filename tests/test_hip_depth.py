"""FULL-DEPTH parity of the HIP path: LLaVA-1.5-7B widths, all 32 decoder layers, policy != reference, against numbers the
fp32 CPU oracle produced offline (oracle/depth_parity.py -> tests/golden/llava7b_depth*_*.json; hours of host-core work,
so it is NOT re-run here).  Both sides build the same 7B model from (seed, tensor name) with the machine-independent
hashed weights (oracle.HashedWeights <-> vlrlhf.utils.synthetic.init_hashed_model), nothing is shipped.

What is compared, per case:
  * the residual stream after EVERY decoder layer at probe positions (a layer wired to the wrong weights / offsets in the
    flat parameter buffer changes it completely; bf16 rounding moves it by a few per cent at depth 32);
  * the four per-sequence log-prob sums and the DPO loss, against the fp32 oracle AND its bf16-emulating mode.
Tolerances come from the oracle's own error budget (profiles/r03_bf16_error_budget_L32.txt, oracle/depth_parity.py): since round 3 the
HIP path keeps the residual stream in fp32, so what it rounds to bf16 is what ANY bf16-MFMA pipeline must round (the MFMA operands) plus
the vision tower; the golden files hold the oracle run with exactly that rounding (variant "f32resid+vit_f32out" = oracle.HIP_ROUNDING) and
the HIP path is asserted to sit no further from fp32 than 1.3 x that model in the per-sequence log-probs."""
import json
import math
import os
from types import SimpleNamespace

import pytest
import torch

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu

PROBE_FEATURES = 32


def _golden(tag):
    return json.load(open(os.path.join(GOLDEN, tag + ".json")))


def _build(layers, qk_scale=1.0):
    from vlrlhf.models.Llava import LlavaDPOTrainer, LlavaForRL
    from vlrlhf.utils.synthetic import LLAVA_1_5_7B, init_hashed_model
    cfg = dict(LLAVA_1_5_7B, layers=layers)
    model = LlavaForRL(cfg)
    ref = init_hashed_model(model, seed=0, std=0.02, policy_delta=1e-3, seed_delta=1, qk_scale=qk_scale)
    tr = LlavaDPOTrainer(model, ref, 0.1, 0, "sigmoid", SimpleNamespace(gradient_accumulation_steps=1), None, -100, 0)
    tr.ref_on_side_stream = False
    return cfg, model, ref, tr


def _check_weights(model, g):
    sd = model.state_dict()
    for k, want in g["weight_probe"].items():
        got = float(sd[k].double().sum())
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), f"{k}: the GPU rebuilt different weights ({got} vs {want})"


def _run_case(cfg, model, ref, tr, g):
    from vlrlhf.utils.synthetic import synthetic_batch
    sp = g["spec"]
    batch = tr._prepare_inputs(synthetic_batch(sp["pairs"], sp["text_len"], cfg["image_token"], 32000, cfg["image_size"], sp["seed"],
                                               ragged=sp["ragged"]))
    model.train()
    pc, pr, _, _ = tr.concatenated_forward(model, batch)          # grad mode: keeps every layer's activations
    c = model._last_ctx
    S, Bn = c["S"], c["Bn"]
    rows = torch.tensor([b * S + p for b in (0, Bn - 1) for p in (0, S // 2, S - 1)], device="cuda")
    probe = [a["x_out"][rows][:, :PROBE_FEATURES].float().reshape(-1).cpu() for a in c["acts"]]
    with torch.no_grad():
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
    losses, _, _ = tr.dpo_loss(pc.detach(), pr.detach(), rc, rr)
    torch.cuda.synchronize()
    got = dict(loss=float(losses.mean()), policy_chosen_logps=pc.tolist(), policy_rejected_logps=pr.tolist(),
               reference_chosen_logps=rc.tolist(), reference_rejected_logps=rr.tolist())
    return got, probe


FLOOR = "f32resid+vit_f32out"      # oracle/depth_parity.py variant = oracle.HIP_ROUNDING: what the fp32-stream HIP path rounds to bf16


def _dlogp(r, f32):
    keys = ("policy_chosen_logps", "policy_rejected_logps", "reference_chosen_logps", "reference_rejected_logps")
    d = [a - b for k in keys for a, b in zip(r[k], f32[k])]
    return max(abs(x) for x in d), math.sqrt(sum(x * x for x in d) / len(d))


def _compare(got, probe, g, label, loss_cap=None):
    """HIP vs the fp32 oracle, judged against the oracle's own model of the path's bf16 rounding (FLOOR: bf16 MFMA operands + the vision
    tower, fp32 residual stream - no bf16-MFMA pipeline rounds less).  The DPO loss of these random-weight models is beta / 2 times a
    difference of four log-prob sums of -85 ... -1415, so what is asserted tightly is the per-sequence log-prob error (max and rms, at most
    1.3 x the floor's); the loss bound follows from it: sigma_loss = beta / 2 * 2 * rms_floor / sqrt(pairs) (slope 1/2 of -logsigmoid at a zero logit; up to 2x that for the
    random-weight logits of these fixtures), asserted at 3 sigma - five builds of this round landed at 0.2, 2.0, 1.2, 0.3 and 2.1 sigma with the
    SAME log-prob errors: the loss is a sample of that distribution, the log-prob bounds above are the parity statement (and at
    the fixed cap where one is given)."""
    f32 = g["results"]["fp32"]
    floor = g["results"].get(FLOOR) or g["results"]["bf16_emulated"]
    L = g["layers"]
    # ---- per-layer residual stream (a mis-wired layer gives O(1); measured worst 0.011 / 0.025 / 0.029 at depth 2 / 32 / 32)
    worst = 0.0
    for l in range(L):
        want = torch.tensor(f32["layer_probe"][l])
        rel = float((probe[l] - want).norm() / want.norm())
        worst = max(worst, rel)
        assert rel < 0.04, f"{label}: layer {l} residual stream is {rel:.3f} (relative) from the fp32 oracle"
    # ---- log-probs and loss
    mx, rms = _dlogp(got, f32)
    fmx, frms = _dlogp(floor, f32)
    l_f32, lf_f32 = abs(got["loss"] - f32["loss"]), abs(floor["loss"] - f32["loss"])
    sigma = g["beta"] / 2 * 2 * frms / math.sqrt(g["spec"]["pairs"])
    print(f"[depth {label}] loss HIP {got['loss']:.6f} fp32 {f32['loss']:.6f} floor model {floor['loss']:.6f} | |HIP-fp32| {l_f32:.2e} "
          f"|floor-fp32| {lf_f32:.2e} (3 sigma {3 * sigma:.2e}) | d logp HIP-fp32 max {mx:.3f} rms {rms:.3f} floor-fp32 max {fmx:.3f} rms {frms:.3f} | "
          f"worst layer residual rel err {worst:.4f}")
    assert math.isfinite(got["loss"])
    assert mx <= 1.3 * fmx + 0.02, (mx, fmx)
    assert rms <= 1.3 * frms + 0.01, (rms, frms)
    assert l_f32 <= 3 * sigma, (l_f32, sigma)
    if loss_cap is not None:
        assert l_f32 <= loss_cap, (l_f32, loss_cap)
    return dict(loss=got["loss"], d_loss_fp32=l_f32, d_logp_max=mx, d_logp_rms=rms, worst_layer_rel=worst)


def test_depth2_true_widths_vs_fp32_oracle():
    g = _golden("llava7b_depth2_small")
    cfg, model, ref, tr = _build(2)
    _check_weights(model, g)
    got, probe = _run_case(cfg, model, ref, tr, g)
    _compare(got, probe, g, "L2 small")
    del model, ref, tr
    import gc
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def full32():
    if torch.cuda.mem_get_info()[1] < 200 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    cfg, model, ref, tr = _build(32)
    yield cfg, model, ref, tr
    del model, ref, tr
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def test_depth32_small_batch_vs_fp32_oracle(full32):
    """1 pair, T = 128 (S = 703), ragged - VERDICT r01 item 1(a)."""
    cfg, model, ref, tr = full32
    g = _golden("llava7b_depth32_small")
    _check_weights(model, g)
    got, probe = _run_case(cfg, model, ref, tr, g)
    # no fixed cap on the loss: two builds of this path with the SAME floor-level log-prob errors (max 0.09 / rms 0.05 here) measured
    # |loss - fp32| = 9.9e-4 and 6.1e-3 at this case (and 9.8e-3 / 1.6e-3 at configs[0]) - the loss is beta / 2 x a difference of four
    # such sums and moves by +- sigma (4.9e-3 here) with any change of rounding order; DESIGN.md section 2
    _compare(got, probe, g, "L32 small")


def test_depth32_configs0_shape_vs_fp32_oracle(full32):
    """BASELINE.json configs[0] on the HIP path: 4 pairs, T = 256 (S = 831), all 32 layers, against the CPU fp32 reference run."""
    cfg, model, ref, tr = full32
    g = _golden("llava7b_depth32_configs0")
    got, probe = _run_case(cfg, model, ref, tr, g)
    _compare(got, probe, g, "L32 configs[0]")
    # and the step itself at this shape: backward + clip + AdamW run, gradient norm finite and non-zero
    from vlrlhf.utils.synthetic import synthetic_batch
    sp = g["spec"]
    batch = tr._prepare_inputs(synthetic_batch(sp["pairs"], sp["text_len"], cfg["image_token"], 32000, cfg["image_size"], sp["seed"], ragged=sp["ragged"]))
    model.engine.init_optimizer()
    model.engine.zero_grad()
    loss = tr.training_step(model, batch)
    model.engine.optimizer_step(1e-6, 0.9, 0.98, 1e-6, 0.0, 1.0)
    torch.cuda.synchronize()
    assert abs(float(loss) - got["loss"]) < 1e-6
    n = model.engine.grad_norm()
    assert math.isfinite(n) and n > 1e-4


def _check_grads(model, g, label, qk_cos=0.99, cos_min=0.99, norm_tol=0.03):
    named = dict(model.named_parameters())
    worst_cos, worst_norm, worst_qk = 1.0, 0.0, 1.0
    for name, e in g["grads"].items():
        hip = named[name].grad.float().reshape(-1)
        norm = float(hip.double().norm())
        probe = hip[:: e["stride"]][:256].cpu()
        want = torch.tensor(e["probe"])
        cs = float(torch.dot(probe, want) / (probe.norm() * want.norm() + 1e-30))
        nr = abs(norm / e["norm"] - 1.0)
        print(f"   {name:70s} norm hip {norm:10.4g} oracle {e['norm']:10.4g} probe cosine {cs:.4f}")
        if any(k in name for k in ("q_proj", "k_proj")):
            # d q / d k pass through the softmax backward dS = P o (dP - delta) with bf16 P and dS operands.  Round 3 only bounded their
            # size ("second-order small"); measured in round 4 they are 0.27 - 0.36 of d v in norm and as well aligned with fp32 autograd
            # as every other gradient: worst probe cosine 0.9993 (2 layers) / 0.9961 (32 layers, layers 17 and 31), norms within 1.3 %
            worst_qk = min(worst_qk, cs)
            assert cs > qk_cos, f"{label} {name}: gradient probe cosine {cs:.3f} vs the fp32 oracle (q / k bound {qk_cos})"
            assert nr < norm_tol, f"{label} {name}: gradient norm {norm:.4g} vs {e['norm']:.4g}"
            continue
        worst_cos, worst_norm = min(worst_cos, cs), max(worst_norm, nr)
        assert cs > cos_min, f"{label} {name}: gradient probe cosine {cs:.3f} vs the fp32 oracle"      # measured worst 0.9999 (2 layers) / 0.9973 (32); sharp fixture 0.986
        assert nr < norm_tol, f"{label} {name}: gradient norm {norm:.4g} vs {e['norm']:.4g}"            # measured worst 0.1 % / 0.7 %; sharp fixture 2.2 %
    print(f"[depth grads {label}] {len(g['grads'])} tensors: worst probe cosine {worst_cos:.4f} (q_proj / k_proj {worst_qk:.4f}), worst norm deviation {worst_norm:.3f}")


def _grad_case(layers, label, fixture="small"):
    from vlrlhf.utils.synthetic import synthetic_batch
    g = _golden(f"llava7b_depth{layers}_{fixture}_grads")
    cfg, model, ref, tr = _build(layers, qk_scale=g.get("qk_scale", 1.0))
    sp = g["spec"]
    batch = tr._prepare_inputs(synthetic_batch(sp["pairs"], sp["text_len"], cfg["image_token"], 32000, cfg["image_size"], sp["seed"], ragged=sp["ragged"]))
    model.engine.zero_grad()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    assert abs(float(loss) - g["loss"]) < 2e-2
    if fixture == "sharp":      # peaked softmax: every gradient is noisier (measured worst cosine 0.983, norms within 2.2 %)
        _check_grads(model, g, label, qk_cos=0.975, cos_min=0.975, norm_tol=0.04)
    else:
        _check_grads(model, g, label)
    del model, ref, tr
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def test_depth2_gradients_vs_fp32_oracle():
    """BACKWARD at the true widths: weight gradients of both layers, the final norm and the lm-head against fp32 autograd of the oracle"""
    _grad_case(2, "L2")


def test_depth32_gradients_vs_fp32_oracle():
    """BACKWARD at full depth: the gradients that have travelled through 31, 14 and 0 further layers (layers 0, 17, 31), the final
    norm and the lm-head, against fp32 autograd of the oracle on the same 7B model (norm within 3 %, 256-element probe cosine > 0.99)"""
    if torch.cuda.mem_get_info()[1] < 200 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    if not os.path.exists(os.path.join(GOLDEN, "llava7b_depth32_small_grads.json")):
        pytest.skip("golden not generated yet (python oracle/depth_parity.py grads)")
    _grad_case(32, "L32")


def test_depth2_sharp_softmax_gradients_vs_fp32_oracle():
    """The `sharp` fixture (q_proj / k_proj weights drawn 2 x larger: peaked softmax, d q and d k as large as d v): every gradient by
    direction and norm against fp32 autograd at the true widths.  Two layers only: at 32 layers this model is chaotic under ANY bf16
    rounding (the oracle's fp32 loss is 1.181, the HIP path's 0.920, layer-0 gradient norms ~1e3) - no parity statement can be made
    there, so the 32-layer direction check of q_proj / k_proj runs on the N(0, 0.02) fixture (test_depth32_gradients_vs_fp32_oracle)."""
    _grad_case(2, "L2 sharp", fixture="sharp")


def test_depth32_seed_sweep_signed_errors_average_to_zero():
    """north_star: loss to rtol 1e-3 of the reference.  One 7B model gives ONE draw of the loss error (sigma = beta x the rms log-prob
    error / sqrt(pairs)), and a draw inside 3 sigma cannot exclude a systematic offset of a whole sigma.  Here the
    `small` case runs under n >= 8 different hashed models and batches (oracle/depth_parity.py seeds: fp32 and the floor model of the
    path's rounding, offline); asserted on the SIGNED errors HIP - fp32:
      * mean loss error within 2 sigma / sqrt(n) of zero, mean log-prob error within 2 rms / sqrt(4 n) of zero (no bias);
      * rms of the loss errors at most 1.5 x the floor model's own rms (the path rounds what the floor model rounds, no more);
      * per-sequence log-prob errors: rms at most 1.3 x the floor's + 0.01 (as in the single-model tests)."""
    if torch.cuda.mem_get_info()[1] < 200 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    path = os.path.join(GOLDEN, "llava7b_depth32_seeds.json")
    if not os.path.exists(path):
        pytest.skip("golden not generated yet (python oracle/depth_parity.py seeds)")
    G = json.load(open(path))
    if len(G["seeds"]) < 4:
        pytest.skip("the sweep needs at least 4 seeds to say anything (oracle run still in progress)")
    from vlrlhf.utils.synthetic import init_hashed_model, synthetic_batch
    cfg, model, ref, tr = _build(G["layers"])
    keys = ("policy_chosen_logps", "policy_rejected_logps", "reference_chosen_logps", "reference_rejected_logps")
    e_loss, f_loss, e_lp, f_lp = [], [], [], []
    for rec in G["seeds"]:
        init_hashed_model(model, seed=rec["seed"], std=0.02, policy_delta=1e-3, seed_delta=rec["seed_delta"], ref=ref)
        sd = model.state_dict()
        for k, want in rec["weight_probe"].items():
            assert abs(float(sd[k].double().sum()) - want) <= 1e-9 * max(1.0, abs(want)), (rec["seed"], k)
        sp = rec["spec"]
        batch = tr._prepare_inputs(synthetic_batch(sp["pairs"], sp["text_len"], cfg["image_token"], 32000, cfg["image_size"], sp["seed"], ragged=sp["ragged"]))
        with torch.no_grad():
            pc, pr, _, _ = tr.concatenated_forward(model, batch)
            rc, rr, _, _ = tr.concatenated_forward(ref, batch)
            losses, _, _ = tr.dpo_loss(pc, pr, rc, rr)
        torch.cuda.synchronize()
        got = dict(loss=float(losses.mean()), policy_chosen_logps=pc.tolist(), policy_rejected_logps=pr.tolist(),
                   reference_chosen_logps=rc.tolist(), reference_rejected_logps=rr.tolist())
        e_loss.append(got["loss"] - rec["fp32"]["loss"])
        f_loss.append(rec["floor"]["loss"] - rec["fp32"]["loss"])
        e_lp += [a - b for k in keys for a, b in zip(got[k], rec["fp32"][k])]
        f_lp += [a - b for k in keys for a, b in zip(rec["floor"][k], rec["fp32"][k])]
        print(f"[seed sweep] seed {rec['seed']}: loss HIP {got['loss']:.6f} fp32 {rec['fp32']['loss']:.6f} floor {rec['floor']['loss']:.6f}  "
              f"HIP-fp32 {e_loss[-1]:+.2e} floor-fp32 {f_loss[-1]:+.2e}")
    n = len(e_loss)
    rms = lambda v: math.sqrt(sum(x * x for x in v) / len(v))      # noqa: E731
    mean = lambda v: sum(v) / len(v)                                  # noqa: E731
    sigma = G["beta"] * rms(f_lp) / math.sqrt(G["seeds"][0]["spec"]["pairs"])
    print(f"[seed sweep] n = {n}: loss error mean {mean(e_loss):+.2e} rms {rms(e_loss):.2e} | floor model mean {mean(f_loss):+.2e} rms {rms(f_loss):.2e} | "
          f"sigma (beta x floor rms log-prob error) {sigma:.2e}, 2 sigma / sqrt(n) {2 * sigma / math.sqrt(n):.2e} | log-prob error mean {mean(e_lp):+.4f} "
          f"rms {rms(e_lp):.4f} (floor mean {mean(f_lp):+.4f} rms {rms(f_lp):.4f})")
    assert abs(mean(e_loss)) <= 2 * sigma / math.sqrt(n), (mean(e_loss), sigma, n)
    assert abs(mean(e_lp)) <= 2 * rms(f_lp) / math.sqrt(len(e_lp)) + 0.005, (mean(e_lp), rms(f_lp))
    assert rms(e_loss) <= 1.5 * max(rms(f_loss), sigma / 2), (rms(e_loss), rms(f_loss), sigma)
    assert rms(e_lp) <= 1.3 * rms(f_lp) + 0.01, (rms(e_lp), rms(f_lp))
    del model, ref, tr
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def test_depth32_planted_outlier_channels():
    """every full-depth comparison used ONE weight distribution (hashed normal, std 0.02) - no massive-activation
    channels, which is where bf16 operand rounding bites in a trained checkpoint.  Here three features of the residual stream receive
    32 x larger MLP outputs in layers 1 and 2 (oracle.HashedWeights(outliers=...): rows of down_proj scaled by a power of two, exact in
    bf16, so the GPU still rebuilds the model bit for bit): they sit ~10 x above the stream's rms from layer 1 to the last layer
    (`stream` in the golden).  The `small` case, HIP against the fp32 oracle, judged by the floor model of the path's rounding ON THE
    SAME model (oracle/depth_parity.py outliers, offline): per-sequence log-prob error at most 1.3 x the floor's, loss within 3 sigma."""
    if torch.cuda.mem_get_info()[1] < 200 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    path = os.path.join(GOLDEN, "llava7b_depth32_outliers.json")
    if not os.path.exists(path):
        pytest.skip("golden not generated yet (python oracle/depth_parity.py outliers)")
    G = json.load(open(path))
    rec = G["record"]
    planted = [r_["mean_abs_planted"] / r_["rms_all"] for r_ in rec["stream"][1:]]
    assert min(planted) > 4.0, planted                   # the fixture does hold outlier channels, all the way down
    from vlrlhf.utils.synthetic import init_hashed_model, synthetic_batch
    cfg, model, ref, tr = _build(G["layers"])
    init_hashed_model(model, seed=rec["seed"], std=0.02, policy_delta=1e-3, seed_delta=rec["seed_delta"], ref=ref, outliers=rec["outliers"])
    _check_weights(model, rec)
    sp = rec["spec"]
    batch = tr._prepare_inputs(synthetic_batch(sp["pairs"], sp["text_len"], cfg["image_token"], 32000, cfg["image_size"], sp["seed"], ragged=sp["ragged"]))
    with torch.no_grad():
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
        losses, _, _ = tr.dpo_loss(pc, pr, rc, rr)
    torch.cuda.synchronize()
    got = dict(loss=float(losses.mean()), policy_chosen_logps=pc.tolist(), policy_rejected_logps=pr.tolist(),
               reference_chosen_logps=rc.tolist(), reference_rejected_logps=rr.tolist())
    mx, rms = _dlogp(got, rec["fp32"])
    fmx, frms = _dlogp(rec["floor"], rec["fp32"])
    sigma = G["beta"] / 2 * 2 * frms / math.sqrt(sp["pairs"])
    l_f32, lf_f32 = abs(got["loss"] - rec["fp32"]["loss"]), abs(rec["floor"]["loss"] - rec["fp32"]["loss"])
    print(f"[depth 32 planted outliers] planted / rms {min(planted):.1f} .. {max(planted):.1f} | loss HIP {got['loss']:.6f} fp32 {rec['fp32']['loss']:.6f} floor model "
          f"{rec['floor']['loss']:.6f} | |HIP-fp32| {l_f32:.2e} |floor-fp32| {lf_f32:.2e} (3 sigma {3 * sigma:.2e}) | d logp HIP-fp32 max {mx:.3f} rms {rms:.3f} "
          f"floor-fp32 max {fmx:.3f} rms {frms:.3f}")
    from tests.golden_util import within
    within("depth32.outliers.dlogp_rms_over_floor", rms / max(frms, 1e-9), default=1.3 + 0.01 / max(frms, 1e-9))
    assert math.isfinite(got["loss"])
    assert mx <= 1.3 * fmx + 0.02, (mx, fmx)
    assert rms <= 1.3 * frms + 0.01, (rms, frms)
    assert l_f32 <= 3 * sigma, (l_f32, sigma)
