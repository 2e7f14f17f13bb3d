"""InternLM-XComposer2-VL (BASELINE.json configs[4]) on the MI355X against tests/golden/internlmxc2_small.npz - outputs of the
reference's own InternLMXC2ForRL forward (LLaVA-style merge, InternLM2 decoder with PLoRA on the image rows) in eval mode, its
get_batch_logps / dpo_loss and autograd (oracle/make_golden_internlm.py) - and, for the stochastic parts (PLoRA dropout, peft LoRA),
against the CPU oracle's restatement with the SAME counter-based masks."""
import math
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import internlm_oracle as IL  # noqa: E402  (checker only)
from oracle import llava_dpo_oracle as O  # noqa: E402  (checker only)
from tests.golden_util import load_case, t, within  # noqa: E402
from tests.test_hip_e2e import TOL_LOGPS_FP32, TOL_LOSS_BF16, TOL_LOSS_FP32, cosine, relmax  # noqa: E402


def build(lora=None, loss_type="sigmoid", **over):
    from vlrlhf.models.InternLMXC2 import InternLMXC2DPOTrainer, InternLMXC2ForRL
    z, cfg, W, W_ref, batch, _ = load_case("internlmxc2_small")
    cfg = dict(cfg, **over)
    model = InternLMXC2ForRL.from_state_dict(cfg, W)
    ref = None
    if lora is None:
        ref = model.create_reference_model()
        ref.weights.load_state_dict(W_ref)
    kw = dict(peft_config=lora) if lora is not None else {}
    tr = InternLMXC2DPOTrainer(model, ref, cfg["beta"], 0, loss_type, SimpleNamespace(gradient_accumulation_steps=1), None, -100,
                               cfg["model_pad_token_id"], **kw)
    return z, cfg, W, W_ref, batch, model, ref, tr


def test_internlm_forward_matches_reference_golden():
    z, cfg, W, W_ref, batch, model, ref, tr = build()
    cb = tr.concatenated_inputs(batch, device=torch.device("cuda"))
    model.eval()
    with torch.no_grad():
        out = model(input_ids=cb["concatenated_input_ids"], attention_mask=cb["concatenated_attention_mask"],
                    labels=cb["concatenated_labels"], use_cache=False, **cb["concatenated_img_input_dict"])
    c = out.logits.c
    assert torch.equal(out.labels.cpu(), t(z, "merged_labels")) and torch.equal(out.image_position_map.cpu(), t(z, "image_position_map"))
    P = (cfg["image_size"] // cfg["patch_size"]) ** 2
    assert relmax(c["feats"].reshape(-1, P, cfg["hidden"]), t(z, "image_features")) < 3e-2      # CLIP LAST hidden state -> mlp2x_gelu
    assert torch.equal(c["pos"][0].cpu().long(), torch.arange(c["S"]))                           # rotary position = index in the merged sequence
    valid = c["mask"].bool().cpu()
    logits = out.logits.materialize().cpu()
    assert relmax(logits[valid], t(z, "logits")[valid]) < 4e-2
    for lt, kw in (("sigmoid", {}), ("ddpo", dict(mask_shared_tokens=True))):
        lp = tr.get_batch_logps(out.logits, out.labels, **kw)
        assert float((lp.cpu() - t(z, f"{lt}.logps")).abs().max()) < TOL_LOGPS_FP32, lt
    with torch.no_grad():
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
    assert float((torch.cat([rc, rr]).cpu() - t(z, "sigmoid.ref_logps")).abs().max()) < TOL_LOGPS_FP32
    losses, _, _ = tr.dpo_loss(pc, pr, rc, rr)
    within("internlm.losses.sigmoid", (losses.cpu() - t(z, "sigmoid.losses")).abs().max())
    bad = cb["concatenated_input_ids"].clone()
    bad[0, 1] = cfg["image_token"]                                   # one more <ImageHere> than images
    with pytest.raises(ValueError, match="image tokens"):
        model(input_ids=bad, attention_mask=cb["concatenated_attention_mask"], labels=cb["concatenated_labels"], **cb["concatenated_img_input_dict"])


def test_internlm_train_step_gradients_match_reference_autograd():
    """full fine-tune (tower + projector frozen): base weights AND the PLoRA pairs get gradients; dropout off = the reference's eval-mode
    autograd.  wqkv / its Plora_B live in q | k | v row order inside the engine: compared after undoing the permutation."""
    z, cfg, W, W_ref, batch, model, ref, tr = build(plora_dropout=0.0)
    eng = model.engine
    eng.init_optimizer()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(z["sigmoid.loss"])) < TOL_LOSS_FP32, (float(loss), float(z["sigmoid.loss"]))
    gsd = type(eng.policy)(eng.layout, eng.dev, eng.grads).state_dict()      # gradients under checkpoint names / row order
    assert "vision_proj.0.weight" in gsd and float(gsd["vision_proj.0.weight"].float().abs().max()) == 0.0      # frozen: never written
    n = 0
    for k in z.files:
        if k.startswith("grad."):
            name, ref_g = k[5:], t(z, k)
            mine = gsd[name].float().cpu()
        elif k.startswith("grad_probe."):
            name, ref_g = k[11:], t(z, k)
            mine = gsd[name].float().cpu().reshape(-1)[::17]
            assert abs(float(gsd[name].float().norm()) / float(z["grad_norm." + name]) - 1.0) < 7e-2, name
        else:
            continue
        cs = cosine(mine, ref_g)
        assert cs > 0.99, f"{name}: cosine {cs:.4f}"
        n += 1
    assert n == 13
    before = eng.policy.v["proj.w1"].clone()
    eng.optimizer_step(lr=1e-3, beta1=0.9, beta2=0.95, eps=1e-6, weight_decay=0.1, max_grad_norm=1.0)
    torch.cuda.synchronize()
    assert torch.equal(eng.policy.v["proj.w1"], before)              # outside the optimizer's range: no update, no weight decay
    loss2 = tr.training_step(model, batch)
    assert torch.isfinite(loss2) and float(loss2) < float(loss)


def test_internlm_policy_equal_reference_gives_ln2():
    z, cfg, W, W_ref, batch, model, ref, tr = build(plora_dropout=0.0)
    ref.weights.load_state_dict(W)
    loss = tr.training_step(model, batch)
    assert float(loss) == pytest.approx(math.log(2.0), abs=1e-6)


def test_internlm_plora_dropout_step_matches_oracle():
    """training mode: PLoRA's dropout (p = 0.05 in the model code; 0.5 here to make it bite) on the image rows of the policy pass only"""
    z, cfg, W, W_ref, batch, model, ref, tr = build(plora_dropout=0.5)
    eng = model.engine
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    from vlrlhf.engine_internlm import PLORA_SEED_XOR
    pseed = ((eng.plora_seed << 40) + (eng._plora_calls << 16)) ^ PLORA_SEED_XOR
    with torch.no_grad():
        l16, a16 = IL.compute_loss(W, W_ref, cfg, batch, cfg["beta"], emulate_bf16=O.HIP_ROUNDING, plora=dict(seed=pseed, p=0.5))
        l_nodrop, a_nd = IL.compute_loss(W, W_ref, cfg, batch, cfg["beta"], emulate_bf16=O.HIP_ROUNDING)
    # the mask matters and it is the right one: judged on the policy log-probs (mean over the pairs, as the trainer logs them) - on this
    # fixture the scalar loss moves by 1e-3 under the mask, the log-probs by whole units
    m = tr._stored_metrics["train"]
    hip_lp = torch.tensor([float(m["logps/chosen"][-1]), float(m["logps/rejected"][-1])])
    with_mask = torch.tensor([float(a16["pc"].mean()), float(a16["pr"].mean())])
    without = torch.tensor([float(a_nd["pc"].mean()), float(a_nd["pr"].mean())])
    print(f"plora dropout: hip {float(loss):.6f} oracle with the mask {float(l16):.6f} without {float(l_nodrop):.6f}; mean policy log-probs "
          f"hip {hip_lp.tolist()} with {with_mask.tolist()} without {without.tolist()}")
    within("internlm.plora_dropout.loss", abs(float(loss) - float(l16)), default=TOL_LOSS_BF16 + 1e-3)
    err, effect = float((hip_lp - with_mask).abs().max()), float((with_mask - without).abs().min())
    within("internlm.plora_dropout.mean_logp", err, default=0.15)
    assert effect > 5 * err, (err, effect)


@pytest.mark.parametrize("dropout,plora_dropout,fused", [(0.0, 0.0, 1), (0.25, 0.0, 1), (0.25, 0.5, 1), (0.25, 0.0, 0)])
def test_internlm_lora_step_matches_oracle(dropout, plora_dropout, fused, monkeypatch):
    """scripts/dpo_internlmxc2vl7b.sh: peft LoRA on the five PLoRA linears (one adapter over the fused wqkv, lora_B rows in the
    checkpoint's per-K/V-head order), base + PLoRA frozen but active, reference = adapters disabled.  fused = 1: the two-adapter C
    layer passes (vlr_decoder_layer_fwd_lora2 / bwd_lora2, fp32 residual stream), with lora_dropout and with PLoRA's own dropout on the
    image rows under it; fused = 0: the layer composed from bf16 primitives (VLR_ILM_LORA_FUSED=0), the cross-check."""
    monkeypatch.setenv("VLR_ILM_LORA_FUSED", str(fused))
    pc = dict(r=8, lora_alpha=8, lora_dropout=dropout, target_modules="auto", bias="none", seed=5)
    z, cfg, W, W_ref, batch, model, ref, tr = build(lora=pc, plora_dropout=plora_dropout)
    assert tr.ref_model is None and tr.is_peft_model
    assert model.engine.lora_fused == bool(fused) and model.engine.resid_f32 == bool(fused)
    lora = IL.random_lora(cfg, r=8, alpha=8, seed=3, b_std=0.05, dropout=dropout)
    lora["W"] = {k: v.bfloat16().float() for k, v in lora["W"].items()}
    eng = model.engine
    eng.load_lora_state_dict(lora["W"])
    names = [n for n, _ in model.named_parameters()]
    assert len(names) == 2 * 5 * cfg["layers"] and all(".lora_" in n for n in names)
    base_before = eng.policy.flat.clone()
    eng.init_optimizer()
    calls0 = eng._plora_calls
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    lora["seed"] = (5 << 40) + (eng._lora_calls << 16)
    Wl = {k: v.clone().requires_grad_(True) for k, v in lora["W"].items()}
    from vlrlhf.engine_internlm import PLORA_SEED_XOR
    pl = None
    if plora_dropout > 0:                       # the seed of the policy pass (the adapters-off reference pass is a forward of the same engine)
        assert eng.last_train_plora_seed in [((eng.plora_seed << 40) + ((calls0 + i) << 16)) ^ PLORA_SEED_XOR for i in (1, 2)]
        pl = dict(seed=eng.last_train_plora_seed, p=plora_dropout, index="full")
    l16, a16 = IL.compute_loss(W, W, cfg, batch, cfg["beta"], emulate_bf16=O.HIP_ROUNDING if fused else True, lora=dict(lora, W=Wl), plora=pl)
    m = tr._stored_metrics["train"]
    hip_lp = torch.tensor([float(m["logps/chosen"][-1]), float(m["logps/rejected"][-1])])
    with_mask = torch.tensor([float(a16["pc"].mean()), float(a16["pr"].mean())])
    err = float((hip_lp - with_mask).abs().max())
    print(f"lora {dropout} plora {plora_dropout} fused {fused}: loss hip {float(loss):.6f} oracle {float(l16):.6f}; mean policy log-probs hip "
          f"{hip_lp.tolist()} oracle {with_mask.tolist()}")
    tag = f"internlm.lora{'2' if fused else '.composed'}.p{dropout}.pp{plora_dropout}"
    within(tag + ".mean_logp", err, default=0.15)
    if pl is not None:                          # and the PLoRA mask is a visible part of the answer (judged on the log-probs, as above)
        with torch.no_grad():
            l_nd, a_nd = IL.compute_loss(W, W, cfg, batch, cfg["beta"], emulate_bf16=O.HIP_ROUNDING, lora=dict(lora, W=lora["W"]))
        without = torch.tensor([float(a_nd["pc"].mean()), float(a_nd["pr"].mean())])
        effect = float((with_mask - without).abs().min())
        print(f"    without the PLoRA mask: loss {float(l_nd):.6f} log-probs {without.tolist()}")
        assert effect > 3 * err, (err, effect)
    within(tag + ".loss", abs(float(loss) - float(l16)), default=TOL_LOSS_BF16 + 2e-3)
    l16.backward()
    # adapter gradients under checkpoint names / row order
    eng.lv, keep = eng.lgv, eng.lv
    try:
        gsd = eng.lora_state_dict()
    finally:
        eng.lv = keep
    worst = 1.0
    for k, v in Wl.items():
        cs = cosine(gsd[k], v.grad)
        worst = min(worst, cs)
        assert cs > 0.98, f"{k}: cosine {cs:.4f}"
    within(tag + ".one_minus_worst_cosine", 1.0 - worst, default=0.02)
    eng.optimizer_step(lr=1e-3, beta1=0.9, beta2=0.95, eps=1e-6, weight_decay=0.1, max_grad_norm=1.0)
    torch.cuda.synchronize()
    assert torch.equal(eng.policy.flat, base_before)
    assert set(model.lora_state_dict()) == set(lora["W"])


def test_internlm_save_and_reload(tmp_path):
    from vlrlhf.models.InternLMXC2 import InternLMXC2ForRL
    from vlrlhf.utils.auto_load import MyAutoModel
    z, cfg, W, W_ref, batch, model, ref, tr = build()
    sd = model.state_dict()
    k = "model.layers.1.attention.wqkv.weight"
    assert torch.equal(sd[k].float().cpu(), W[k])                    # checkpoint row order (with ONE K/V head it equals q | k | v; the
                                                                     # non-trivial permutation is pinned by test_internlm_gqa_row_order_*)
    model.save_pretrained(str(tmp_path))
    m2 = MyAutoModel.from_pretrained(str(tmp_path))
    assert isinstance(m2, InternLMXC2ForRL) and m2.engine.nkv == cfg["kv_heads"]
    assert torch.equal(m2.engine.policy.flat, model.engine.policy.flat)


def test_internlm_gqa_two_kv_heads_vs_oracle():
    """the fixture has ONE K/V head, where the checkpoint's wqkv row order [q_0..q_{g-1} | k | v] per K/V head already is q | k | v.
    Two K/V heads (4 query heads, hidden 512) make the load-time permutation non-trivial: random weights, HIP loss vs the fp32 oracle
    (whose split of the fused projection is the reference's rearrange, pinned on CPU by test_internlm_gqa_row_order_matches_reference)."""
    from vlrlhf.models.InternLMXC2 import InternLMXC2DPOTrainer, InternLMXC2ForRL
    z, cfg0, _, _, batch, _ = load_case("internlmxc2_small")
    cfg = dict(cfg0, hidden=512, inter=256, heads=4, kv_heads=2, layers=1, plora_dropout=0.0)
    g = torch.Generator().manual_seed(11)
    model = InternLMXC2ForRL(cfg)
    W = {}
    for hf, name, r0, rows in model.engine.layout.hf_names():
        shp = model.engine.layout.shape[name]
        full = (rows,) + tuple(shp[1:]) if len(shp) > 1 else shp
        W[hf] = ((1.0 + 0.1 * torch.randn(full, generator=g)) if "norm" in hf else torch.randn(full, generator=g) * (0.03 if "Plora" in hf else 0.06)).bfloat16().float()
    _, _, W0, _, _, _ = load_case("internlmxc2_small")
    W.update({k: v for k, v in W0.items() if k.startswith("vit.")})
    W["vision_proj.0.weight"], W["vision_proj.0.bias"] = torch.randn(512, cfg["vit_hidden"], generator=g).mul(0.06).bfloat16().float(), torch.zeros(512)
    W["vision_proj.2.weight"], W["vision_proj.2.bias"] = torch.randn(512, 512, generator=g).mul(0.06).bfloat16().float(), torch.zeros(512)
    W_ref = {k: (v + 0.02 * torch.randn(v.shape, generator=g)).bfloat16().float() if k.startswith("model.layers") and k.endswith(".weight") and "norm" not in k else v for k, v in W.items()}
    model.engine.load_state_dict(W)
    ref = model.create_reference_model()
    ref.weights.load_state_dict(W_ref)
    tr = InternLMXC2DPOTrainer(model, ref, 0.1, 0, "sigmoid", SimpleNamespace(gradient_accumulation_steps=1), None, -100, cfg["model_pad_token_id"])
    loss = tr.training_step(model, batch)
    with torch.no_grad():
        l32, _ = IL.compute_loss(W, W_ref, cfg, batch, 0.1)
    assert abs(float(loss) - float(l32)) < TOL_LOSS_FP32, (float(loss), float(l32))
    assert not torch.equal(model.engine.policy.v["l0.wqkv"].float().cpu(), W["model.layers.0.attention.wqkv.weight"])
    assert torch.equal(model.state_dict()["model.layers.0.attention.wqkv.weight"].float().cpu(), W["model.layers.0.attention.wqkv.weight"])


def test_internlm_gradient_checkpointing_is_bit_identical():
    """full fine-tune with PLoRA dropout under --gradient_checkpointing: the recompute re-runs the C layer pass with the pass's own dropout
    seed - loss and every gradient (base AND PLoRA) bit-identical to the run that keeps the activations; the fp32 residual stream is on"""
    outs = []
    for ckpt in (False, True):
        z, cfg, W, W_ref, batch, model, ref, tr = build(plora_dropout=0.25)
        assert model.engine.resid_f32
        if ckpt:
            model.gradient_checkpointing_enable()
        model.train()
        loss = tr.training_step(model, batch)
        torch.cuda.synchronize()
        assert model._last_ctx["ckpt"] == ckpt
        outs.append((float(loss), model.engine.grads.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].float().abs().sum()) > 0


def test_internlm_lora_gradient_checkpointing_is_bit_identical():
    """peft LoRA over PLoRA (both dropouts on) under --gradient_checkpointing: the recompute re-runs vlr_decoder_layer_fwd_lora2 with the
    pass's two dropout seeds - loss and every adapter gradient bit-identical to the run that keeps the activations"""
    outs = []
    for ckpt in (False, True):
        pc = dict(r=8, lora_alpha=8, lora_dropout=0.25, target_modules="auto", bias="none", seed=5)
        z, cfg, W, W_ref, batch, model, ref, tr = build(lora=pc, plora_dropout=0.25)
        lora = IL.random_lora(cfg, r=8, alpha=8, seed=3, b_std=0.05, dropout=0.25)
        model.engine.load_lora_state_dict({k: v.bfloat16().float() for k, v in lora["W"].items()})
        assert model.engine.lora_fused and model.engine.resid_f32
        if ckpt:
            model.gradient_checkpointing_enable()
        model.train()
        loss = tr.training_step(model, batch)
        torch.cuda.synchronize()
        assert model._last_ctx["ckpt"] == ckpt
        outs.append((float(loss), model.engine.lora_grads.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].float().abs().sum()) > 0
