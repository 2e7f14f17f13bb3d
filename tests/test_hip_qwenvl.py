"""Qwen-VL (BASELINE.json configs[2]) on the MI355X against tests/golden/qwenvl_small.npz - outputs of the reference's own
QWenLMHeadModel / VisionTransformer / Resampler forward, get_batch_logps, dpo_loss and autograd (oracle/make_golden_qwenvl.py) - and,
for the LoRA step (peft is not in the container), against the CPU oracle's restatement with the SAME dropout mask."""
import json
import math
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import llava_dpo_oracle as O  # noqa: E402  (checker only)
from oracle import qwenvl_oracle as Q  # noqa: E402  (checker only)
from tests.golden_util import load_case, t, within  # noqa: E402
from tests.test_hip_e2e import TOL_LOGPS_FP32, TOL_LOSS_BF16, TOL_LOSS_FP32, cosine, relmax  # noqa: E402

EMUQ = O.HIP_ROUNDING | {"vit"}      # fp32 residual stream in the decoder; the whole Qwen vision tower + resampler (incl. its output) in bf16


def build(lora=None, loss_type="sigmoid"):
    from vlrlhf.models.QwenVL import QwenVLDPOTrainer, QwenVLForRL
    z, cfg, W, W_ref, batch, _ = load_case("qwenvl_small")
    model = QwenVLForRL.from_state_dict(cfg, W)
    ref = None
    if lora is None:
        ref = model.create_reference_model()
        ref.weights.load_state_dict(W_ref)
    kw = dict(peft_config=lora) if lora is not None else {}
    tr = QwenVLDPOTrainer(model, ref, cfg["beta"], 0, loss_type, SimpleNamespace(gradient_accumulation_steps=1), None, -100,
                          cfg["pad_token_id"], **kw)
    return z, cfg, W, W_ref, batch, model, ref, tr


def test_qwenvl_forward_matches_reference_golden():
    z, cfg, W, W_ref, batch, model, ref, tr = build()
    px = batch["img_input_dict"]["pixel_values"]
    feat = model.engine.vision_features(px.cuda())
    nq, E = cfg["visual"]["n_queries"], cfg["visual"]["output_dim"]
    assert relmax(feat.reshape(-1, nq, E), t(z, "policy_visual_features")) < 3e-2   # ViT (head_dim 104 on the 128-wide kernel) + resampler
    xr = model.engine.vit_trunk(px.cuda())
    fr, _ = model.engine.resampler_fwd(ref.weights, xr, px.shape[0], "vf", False)   # the reference model's own (different) resampler
    assert relmax(fr.reshape(-1, nq, E), t(z, "visual_features")) < 3e-2
    cb = tr.concatenated_inputs(batch, device=torch.device("cuda"))
    model.eval()
    with torch.no_grad():
        out = model(input_ids=cb["concatenated_input_ids"], attention_mask=cb["concatenated_attention_mask"],
                    labels=cb["concatenated_labels"], use_cache=False, **cb["concatenated_img_input_dict"])
    assert torch.equal(out.image_position_map.cpu(), t(z, "image_position_map"))
    assert torch.equal(out.labels.cpu(), cb["concatenated_labels"].cpu())           # no token expansion: the labels pass through
    valid = cb["concatenated_attention_mask"].bool().cpu()
    logits = out.logits.materialize().cpu()
    assert relmax(logits[valid], t(z, "logits")[valid]) < 4e-2
    for lt, kw in (("sigmoid", {}), ("ddpo", dict(mask_shared_tokens=True))):
        lp = tr.get_batch_logps(out.logits, out.labels, **kw)
        assert float((lp.cpu() - t(z, f"{lt}.logps")).abs().max()) < TOL_LOGPS_FP32, lt
    n_tok = (out.labels[:, 1:] != -100).sum(-1).cpu()
    avg = tr.get_batch_logps(out.logits, out.labels, average_log_prob=True)
    assert float((avg.cpu() - t(z, "sigmoid.logps") / n_tok).abs().max()) < 2e-2
    with torch.no_grad():
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
    assert float((torch.cat([rc, rr]).cpu() - t(z, "sigmoid.ref_logps")).abs().max()) < TOL_LOGPS_FP32
    # the reference behaviour without pixels: the image files named in the ids are opened inside forward
    with pytest.raises(FileNotFoundError):
        model(input_ids=cb["concatenated_input_ids"], attention_mask=cb["concatenated_attention_mask"], labels=cb["concatenated_labels"])
    # fewer images than <img> spans
    with pytest.raises(ValueError, match="images"):
        model(input_ids=cb["concatenated_input_ids"], attention_mask=cb["concatenated_attention_mask"], labels=cb["concatenated_labels"],
              pixel_values=px[:1].cuda())
    bad = cb["concatenated_input_ids"].clone()
    a = int((bad[0] == cfg["image_start_id"]).nonzero()[0])
    bad[0, a + 2] = cfg["image_start_id"] + 1                        # a second </img>: unbalanced markers
    with pytest.raises(ValueError):
        model(input_ids=bad, attention_mask=cb["concatenated_attention_mask"], labels=cb["concatenated_labels"], **cb["concatenated_img_input_dict"])


@pytest.mark.parametrize("loss_type", ["sigmoid", "ipo", "ddpo"])
def test_qwenvl_losses_match_reference_golden(loss_type):
    z, cfg, W, W_ref, batch, model, ref, tr = build(loss_type=loss_type)
    model.eval()
    with torch.no_grad():
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
    losses, _, _ = tr.dpo_loss(pc, pr, rc, rr)
    exp = t(z, f"{loss_type}.losses")
    # vs the fp32 reference run by the reference's own classes: tolerances = 1.5 x the measured error (tests/golden_util.MEASURED_TOL)
    if loss_type == "ipo":       # (log-ratio - 1/(2 beta))^2: compare the roots - the log-ratio is a sum of four log-probs, each within ~0.15
        within("qwenvl.losses.ipo.sqrt", (losses.cpu().sqrt() - exp.sqrt()).abs().max())
    else:
        within(f"qwenvl.losses.{loss_type}", (losses.cpu() - exp).abs().max())


def test_qwenvl_train_step_gradients_match_reference_autograd():
    """full fine-tune as the reference runs it (ViT trunk frozen, language model AND the resampler `attn_pool` trained): loss and the
    gradients the reference's autograd produced - the fused biased c_attn (weight AND bias), c_proj, w1 / w2 (stored as up / gate),
    mlp.c_proj, norms, lm_head, wte, and the resampler's query, kv_proj, in_proj (weight / bias), out_proj, ln_q, ln_kv"""
    z, cfg, W, W_ref, batch, model, ref, tr = build()
    eng = model.engine
    eng.init_optimizer()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(z["sigmoid.loss"])) < TOL_LOSS_FP32, (float(loss), float(z["sigmoid.loss"]))
    g = {n: p.grad for n, p in model.named_parameters()}
    assert all(n.startswith("transformer.visual.attn_pool.") for n in g if n.startswith("transformer.visual"))
    n = 0
    for k in z.files:
        if k.startswith("grad."):
            name, ref_g = k[5:], t(z, k)
            mine = g[name].float().cpu()
        elif k.startswith("grad_probe."):
            name, ref_g = k[11:], t(z, k)
            mine = g[name].float().cpu().reshape(-1)[::17]
            assert abs(float(g[name].float().norm()) / float(z["grad_norm." + name]) - 1.0) < 6e-2, name
        else:
            continue
        cs = cosine(mine, ref_g)
        assert cs > (0.98 if "attn_pool" in name else 0.99), f"{name}: cosine {cs:.4f}"
        n += 1
    assert n == 20
    eng.optimizer_step(lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.05, max_grad_norm=1.0)
    loss2 = tr.training_step(model, batch)
    assert torch.isfinite(loss2) and float(loss2) < float(loss)


def test_qwenvl_policy_equal_reference_gives_ln2():
    z, cfg, W, W_ref, batch, model, ref, tr = build()
    ref.weights.load_state_dict(W)
    loss = tr.training_step(model, batch)
    assert float(loss) == pytest.approx(math.log(2.0), abs=1e-6)


@pytest.mark.parametrize("dropout", [0.0, 0.25])
def test_qwenvl_lora_step_matches_oracle(dropout):
    """scripts/dpo_qwenvl.sh: LoRA on c_attn (ONE adapter over the fused, biased q|k|v projection), attn.c_proj, w1, w2 - no adapter
    on mlp.c_proj; base frozen; reference = adapters disabled"""
    pc = dict(r=8, lora_alpha=16, lora_dropout=dropout, target_modules="auto", bias="none", seed=5)
    z, cfg, W, W_ref, batch, model, ref, tr = build(lora=pc)
    assert tr.ref_model is None and tr.is_peft_model
    lora = Q.random_lora(cfg, r=8, alpha=16, seed=3, b_std=0.05, dropout=dropout)
    lora["W"] = {k: v.bfloat16().float() for k, v in lora["W"].items()}
    eng = model.engine
    eng.load_lora_state_dict(lora["W"])
    names = [n for n, _ in model.named_parameters()]
    assert len(names) == 2 * 4 * cfg["layers"] and all(".lora_" in n for n in names) and not any("mlp.c_proj" in n for n in names)
    assert tuple(eng.lv["l0.a_qkv"].shape) == (8, cfg["hidden"]) and tuple(eng.lv["l0.b_qkv"].shape) == (3 * cfg["hidden"], 8)
    base_before = eng.policy.flat.clone()
    eng.init_optimizer()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    lora["seed"] = (5 << 40) + (eng._lora_calls << 16)
    px = batch["img_input_dict"]["pixel_values"]
    Wl = {k: v.clone().requires_grad_(True) for k, v in lora["W"].items()}
    l16, m16 = Q.compute_loss(W, W, cfg, dict(batch, pixel_values=px), cfg["beta"], emulate_bf16=EMUQ, lora=dict(lora, W=Wl))
    within(f"qwenvl.lora.loss.p{dropout}", abs(float(loss) - float(l16)), default=TOL_LOSS_BF16 + 1e-3)      # (this fixture's weights are scaled x3)
    l16.backward()
    named = dict(model.named_parameters())
    worst = 1.0
    for k, v in Wl.items():
        hf = k.replace(".weight", ".default.weight")
        cs = cosine(named[hf].grad, v.grad)
        worst = min(worst, cs)
        assert cs > 0.98, f"{k}: cosine {cs:.4f}"
    within(f"qwenvl.lora.one_minus_worst_cosine.p{dropout}", 1.0 - worst, default=0.02)
    eng.optimizer_step(lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.05, max_grad_norm=1.0)
    torch.cuda.synchronize()
    assert torch.equal(eng.policy.flat, base_before)                 # the base weights never move under LoRA
    sd = model.lora_state_dict()
    assert set(sd) == set(lora["W"])


def test_qwenvl_save_and_reload(tmp_path):
    from vlrlhf.models.QwenVL import QwenVLForRL
    from vlrlhf.utils.auto_load import MyAutoModel
    z, cfg, W, W_ref, batch, model, ref, tr = build()
    model.save_pretrained(str(tmp_path))
    hf = json.load(open(tmp_path / "config.json"))
    assert hf["architectures"] == ["QWenLMHeadModel"] and hf["intermediate_size"] == 2 * cfg["inter"]
    m2 = MyAutoModel.from_pretrained(str(tmp_path))
    assert isinstance(m2, QwenVLForRL)
    assert torch.equal(m2.engine.policy.flat, model.engine.policy.flat)
    px = batch["img_input_dict"]["pixel_values"].cuda()
    assert torch.equal(m2.engine.vision_features(px), model.engine.vision_features(px))      # the frozen tower round-trips too
