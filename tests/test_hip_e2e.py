"""End-to-end parity of the MI355X DPO step (model wrapper + trainer + optimizer, everything through the C ABI) against
(1) the golden vectors generated from the reference's own functions and (2) the CPU oracle on seeded inputs.
Needs a real MI355X:  pytest -m gpu"""
import copy
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import llava_dpo_oracle as O  # noqa: E402  (checker only)
from tests.golden_util import load_case, t, within  # noqa: E402

# Tolerances.  The HIP path stores weights/activations/gradients in bf16 (fp32 accumulation, fp32 softmax / norms /
# log-softmax / loss); the golden vectors are fp32.  On the toy fixture the bf16-emulated ORACLE itself sits 2.3e-3
# (relative) from the fp32 loss, so: vs fp32 golden  |dloss| <= 6e-3, |dlogps| <= 0.25 (values ~ -75..-150);
# vs the bf16-emulated oracle |dloss| <= 3e-3.  north_star's rtol = 1e-3 is checked where bf16 allows it: against the
# bf16-emulated oracle on the 7B-shaped single-layer case below (test_true_width_layer_loss).
# measured on the toy fixtures (round 3, fp32 residual stream): |loss - fp32 golden| <= 2.5e-3 over all model families (InternLM, on the bf16
# stream, is the largest), per-sequence log-probs within 0.1, loss within 2e-3 of the oracle run with the path's own rounding (EMU)
TOL_LOSS_FP32, TOL_LOGPS_FP32, TOL_LOSS_BF16 = 3e-3, 0.1, 2e-3
EMU = O.HIP_ROUNDING      # the oracle's model of what the fp32-residual-stream HIP path rounds to bf16 (oracle/llava_dpo_oracle.py)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vlrlhf import _hip
    _hip.lib()
    return torch.device("cuda")


def relmax(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


def cosine(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def build(cfg, W, W_ref):
    from vlrlhf.models.Llava import LlavaForRL
    model = LlavaForRL.from_state_dict(cfg, W)
    ref = model.create_reference_model()
    ref.weights.load_state_dict(W_ref)
    return model, ref


def make_trainer(model, ref, cfg, loss_type="sigmoid", **kw):
    from vlrlhf.models.Llava import LlavaDPOTrainer
    args = SimpleNamespace(gradient_accumulation_steps=1, per_device_train_batch_size=2, learning_rate=cfg["optim"]["lr"],
                           adam_beta1=cfg["optim"]["beta1"], adam_beta2=cfg["optim"]["beta2"], adam_epsilon=cfg["optim"]["eps"],
                           weight_decay=cfg["optim"]["weight_decay"], max_grad_norm=cfg["optim"]["max_grad_norm"], seed=0)
    tr = LlavaDPOTrainer(model, ref, cfg["beta"], kw.pop("label_smoothing", 0), loss_type, args, None, -100, 0, "keep_end",
                         None, None, None, **kw)
    return tr


def test_forward_matches_golden(gpu):
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    model, ref = build(cfg, W, W_ref)
    tr = make_trainer(model, ref, cfg)
    cb = tr.concatenated_inputs(batch, device=gpu)
    model.eval()
    with torch.no_grad():
        out = model(input_ids=cb["concatenated_input_ids"], attention_mask=cb["concatenated_attention_mask"],
                    labels=cb["concatenated_labels"], use_cache=False, **cb["concatenated_img_input_dict"])
    assert torch.equal(out.labels.cpu(), t(z, "merged_labels"))
    assert torch.equal(out.image_position_map.cpu(), t(z, "image_position_map"))
    c = out.logits.c
    assert torch.equal(c["mask"].cpu().long(), t(z, "merged_mask"))
    assert torch.equal(c["pos"].cpu().long(), t(z, "merged_pos"))
    B = batch["chosen_input_ids"].shape[0]
    assert relmax(c["vit_feat"].reshape(B, -1, cfg["vit_hidden"]), t(z, "vit_feat")) < 3e-2
    assert relmax(c["feats"].reshape(B, -1, cfg["hidden"]), t(z, "image_features")) < 3e-2
    valid = t(z, "merged_mask").bool()
    S = c["S"]
    hid = c["hidden"].float().cpu().reshape(2 * B, S, -1)
    assert relmax(hid[valid], t(z, "hidden_last")[valid]) < 4e-2
    logits = out.logits.materialize().cpu()
    assert tuple(out.logits.shape) == tuple(z["logits"].shape)
    assert relmax(logits[valid], t(z, "logits")[valid]) < 4e-2
    lp = tr.get_batch_logps(out.logits, out.labels)
    assert float((lp.cpu() - t(z, "policy_logps")).abs().max()) < TOL_LOGPS_FP32
    # same op on the materialised tensor (reference A4 path)
    lp2 = tr.get_batch_logps(out.logits.materialize(), out.labels)
    assert float((lp2.cpu() - lp.cpu()).abs().max()) < 2e-3
    lpa = tr.get_batch_logps(out.logits, out.labels, average_log_prob=True)
    assert float((lpa.cpu() - t(z, "policy_logps_avg")).abs().max()) < 2e-2
    lpd = tr.get_batch_logps(out.logits, out.labels, mask_shared_tokens=True)
    assert float((lpd.cpu() - t(z, "policy_logps_ddpo")).abs().max()) < TOL_LOGPS_FP32
    # logits/* metric without materialising
    assert abs(float(out.logits[:B].mean()) - float(t(z, "logits")[:B].mean())) < 2e-2 * float(t(z, "logits").abs().mean()) + 1e-3
    with pytest.raises(ValueError):
        tr.get_batch_logps(out.logits, out.labels[:, :-1])


@pytest.mark.parametrize("loss_type", ["sigmoid", "hinge", "ipo", "kto_pair", "ddpo"])
def test_losses_match_golden(gpu, loss_type):
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    model, ref = build(cfg, W, W_ref)
    tr = make_trainer(model, ref, cfg, loss_type)
    model.eval()
    with torch.no_grad():
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
    key = "ref_logps_ddpo" if loss_type == "ddpo" else "ref_logps"
    assert float((torch.cat([rc, rr]).cpu() - t(z, key)).abs().max()) < TOL_LOGPS_FP32
    losses, cr, rw = tr.dpo_loss(pc, pr, rc, rr)
    exp = t(z, f"loss_{loss_type}")
    within(f"llava.losses.{loss_type}", (losses.cpu() - exp).abs().max())
    within(f"llava.chosen_rewards.{loss_type}", (cr.cpu() - t(z, f"chosen_rewards_{loss_type}")).abs().max())
    tr2 = make_trainer(model, ref, cfg, "nope")
    with pytest.raises(ValueError, match="Unknown loss type"):
        tr2.dpo_loss(pc, pr, rc, rr)


def test_train_step_matches_golden(gpu):
    """compute_loss -> backward -> clip -> AdamW on the fixture the reference functions generated."""
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    model, ref = build(cfg, W, W_ref)
    tr = make_trainer(model, ref, cfg)
    eng = model.engine
    eng.init_optimizer()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    exp_loss = float(z["loss_mean_sigmoid"])
    assert abs(float(loss) - exp_loss) < TOL_LOSS_FP32, (float(loss), exp_loss)
    # bf16-emulated oracle
    with torch.no_grad():
        Wp = {k: v.bfloat16().float() for k, v in W.items()}
        Wq = {k: v.bfloat16().float() for k, v in W_ref.items()}
        l16, m16 = O.compute_loss(Wp, Wq, cfg, batch, cfg["beta"], emulate_bf16=EMU)
    assert abs(float(loss) - float(l16)) < TOL_LOSS_BF16, (float(loss), float(l16))
    # the eight metrics of trl's get_batch_loss_metrics
    logs = tr.log({"loss": float(loss)})
    l32, m32 = None, None
    with torch.no_grad():
        l32, m32 = O.compute_loss(W, W_ref, cfg, batch, cfg["beta"])
    for k, v in m32.items():
        assert k in logs, k
        assert abs(logs[k] - float(v)) < (0.3 if k.startswith("logps") else 3e-2), (k, logs[k], float(v))
    # gradients vs the golden fp32 gradients
    worst = 0.0
    named = dict(model.named_parameters())
    n = 0
    for k in z.files:
        if not k.startswith("grad."):
            continue
        name = k[5:]
        g = named[name].grad
        exp = t(z, k)
        cs = cosine(g, exp)
        rm = relmax(g, exp)
        worst = max(worst, rm)
        assert cs > 0.995, (name, cs, rm)
        assert rm < 8e-2, (name, cs, rm)
        n += 1
    assert n == len(named)
    # clip + AdamW
    hp = cfg["optim"]
    out = eng.optimizer_step(lr=hp["lr"], beta1=hp["beta1"], beta2=hp["beta2"], eps=hp["eps"], weight_decay=hp["weight_decay"],
                             max_grad_norm=hp["max_grad_norm"])
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(z["grad_norm"])) < 2e-2 * float(z["grad_norm"])
    after = model.state_dict()
    sq = 0.0
    for k in z.files:
        if k.startswith("after_step."):
            name = k[len("after_step."):]
            upd_exp = t(z, k) - W[name]
            # the fp32 master copy is the optimizer's truth; the bf16 copy is its rounding
            off = None
            for hf, nm, r0, rows_ in eng.layout.hf_names():
                if hf == name:
                    base = eng.layout.offset[nm]
                    shape = eng.layout.shape[nm]
                    cols = shape[1] if len(shape) == 2 else 1
                    off = (base + r0 * cols, rows_ * cols if len(shape) == 2 else shape[0])
            m_new = eng.master[off[0]: off[0] + off[1]].cpu().reshape(W[name].shape)
            upd = m_new - W[name].bfloat16().float()
            # Adam's first step is +-lr wherever |g| >> eps: compare the update direction, not ulps
            assert cosine(upd, upd_exp) > 0.9, (name, cosine(upd, upd_exp))
            assert float(upd.abs().max()) <= 1.06 * hp["lr"] * (1 + hp["weight_decay"] * float(W[name].abs().max()) * 50)
            assert torch.equal(after[name].cpu(), m_new.bfloat16())


def test_second_step_and_grad_accumulation(gpu):
    """two micro-steps accumulate into the bf16 gradient buffer exactly like 2x one micro-step's gradient."""
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    model, ref = build(cfg, W, W_ref)
    tr = make_trainer(model, ref, cfg)
    eng = model.engine
    tr.training_step(model, batch)
    g1 = eng.grads.float().clone()
    tr.training_step(model, batch)        # no zero_grad in between -> accumulate
    torch.cuda.synchronize()
    g2 = eng.grads.float()
    assert relmax(g2, 2 * g1) < 2e-2
    eng.zero_grad()
    tr.training_step(model, batch)
    torch.cuda.synchronize()
    assert relmax(eng.grads.float(), g1) < 1e-6 or cosine(eng.grads.float(), g1) > 0.9999


MID = dict(vit_hidden=128, vit_mlp=256, vit_layers=3, vit_heads=2, image_size=56, patch_size=14, hidden=256, inter=512,
           layers=2, heads=2, vocab=512, image_token=500, model_pad_token_id=501, beta=0.1,
           optim=dict(lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.05, max_grad_norm=1.0))


def test_multi_head_ragged_vs_oracle(gpu):
    """2 heads / 2 ViT heads, ragged right-padded batch, policy != reference, against the CPU oracle."""
    cfg = dict(MID)
    W_ref = O.random_weights(cfg, seed=3, std=0.05)
    g = torch.Generator().manual_seed(9)
    W = {k: (v + 0.02 * v.abs().mean() * torch.randn(v.shape, generator=g)) if not k.startswith("vision_tower.") else v
         for k, v in W_ref.items()}
    batch = O.synthetic_batch(3, 40, cfg["image_token"], 480, cfg["image_size"], seed=5, ragged=True)
    model, ref = build(cfg, W, W_ref)
    tr = make_trainer(model, ref, cfg)
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    Wp = {k: v.bfloat16().float() for k, v in W.items()}
    Wq = {k: v.bfloat16().float() for k, v in W_ref.items()}
    with torch.no_grad():
        l16, _ = O.compute_loss(Wp, Wq, cfg, batch, cfg["beta"], emulate_bf16=EMU)
    l32, m32 = O.compute_loss({k: v.clone().requires_grad_(not k.startswith("vision_tower.")) for k, v in W.items()}, W_ref, cfg, batch, cfg["beta"])
    assert abs(float(loss) - float(l16)) < TOL_LOSS_BF16, (float(loss), float(l16), float(l32))
    assert abs(float(loss) - float(l32)) < TOL_LOSS_FP32


def test_true_width_layer_loss(gpu):
    """LLaVA-1.5-7B WIDTHS (H=4096, 32 heads x 128, I=11008, V=32064) at depth 1: catches tiling / edge bugs of the
    256x256 GEMM tiles, the fused lm-head and the attention kernels at the real shapes.  north_star asks for loss parity
    at rtol=1e-3: checked here against the bf16-emulated CPU oracle (policy != reference, ragged batch)."""
    cfg = dict(vit_hidden=128, vit_mlp=256, vit_layers=3, vit_heads=2, image_size=56, patch_size=14, hidden=4096, inter=11008,
               layers=1, heads=32, vocab=32064, image_token=32000, model_pad_token_id=32001, beta=0.1,
               optim=dict(lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.0, max_grad_norm=1.0))
    W_ref = O.random_weights(cfg, seed=11, std=0.02)
    g = torch.Generator().manual_seed(12)
    W = {k: (v + 0.05 * v.abs().mean() * torch.randn(v.shape, generator=g)) if not k.startswith("vision_tower.") else v
         for k, v in W_ref.items()}
    batch = O.synthetic_batch(2, 48, cfg["image_token"], 32000, cfg["image_size"], seed=13, ragged=True)
    model, ref = build(cfg, W, W_ref)
    tr = make_trainer(model, ref, cfg)
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    Wp = {k: v.bfloat16().float() for k, v in W.items()}
    Wq = {k: v.bfloat16().float() for k, v in W_ref.items()}
    with torch.no_grad():
        l16, m16 = O.compute_loss(Wp, Wq, cfg, batch, cfg["beta"], emulate_bf16=EMU)
        l32, _ = O.compute_loss(W, W_ref, cfg, batch, cfg["beta"])
    rel16 = abs(float(loss) - float(l16)) / abs(float(l16))
    rel32 = abs(float(loss) - float(l32)) / abs(float(l32))
    print(f"true-width loss: hip {float(loss):.6f} oracle-bf16 {float(l16):.6f} oracle-fp32 {float(l32):.6f} rel {rel16:.2e} / {rel32:.2e}")
    assert rel16 < 1e-3, (float(loss), float(l16), rel16)
    assert rel32 < 5e-3, (float(loss), float(l32), rel32)
    # gradient of the fused lm-head and the big GEMMs against fp32 autograd on the oracle
    leaves = {k: v.clone().requires_grad_(not k.startswith("vision_tower.")) for k, v in W.items()}
    lg, _ = O.compute_loss(leaves, W_ref, cfg, batch, cfg["beta"])
    lg.backward()
    named = dict(model.named_parameters())
    for name in ("language_model.lm_head.weight", "language_model.model.layers.0.mlp.down_proj.weight",
                 "language_model.model.layers.0.self_attn.q_proj.weight", "language_model.model.layers.0.mlp.gate_proj.weight",
                 "multi_modal_projector.linear_2.weight"):
        cs = cosine(named[name].grad, leaves[name].grad)
        assert cs > 0.99, (name, cs)


def test_reference_logps_from_batch_and_reference_free(gpu):
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    model, ref = build(cfg, W, W_ref)
    tr = make_trainer(model, ref, cfg)
    B = batch["chosen_input_ids"].shape[0]
    b2 = dict(batch)
    b2["reference_chosen_logps"] = t(z, "ref_logps")[:B]
    b2["reference_rejected_logps"] = t(z, "ref_logps")[B:]
    l_pre = tr.training_step(model, b2)            # precomputed reference log-probs: no reference forward
    model.engine.zero_grad()
    l_run = tr.training_step(model, batch)
    torch.cuda.synchronize()
    assert abs(float(l_pre) - float(l_run)) < 3e-3
    tr3 = make_trainer(model, None, cfg, reference_free=True)
    pc = torch.tensor([-1.0, -2.0], device=gpu)
    pr = torch.tensor([-2.5, -1.0], device=gpu)
    losses, _, _ = tr3.dpo_loss(pc, pr, pc * 0 - 7, pr * 0 - 9)
    exp, _, _ = O.dpo_loss(pc.cpu(), pr.cpu(), pc.cpu() * 0 - 7, pr.cpu() * 0 - 9, cfg["beta"], reference_free=True)
    assert torch.allclose(losses.cpu(), exp, atol=1e-6)


def test_wrong_image_count_raises(gpu):
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    model, ref = build(cfg, W, W_ref)
    ids = batch["chosen_input_ids"].clone()
    ids[0, 5] = cfg["image_token"]           # a second <image> in row 0 but only one image per row supplied
    with torch.no_grad(), pytest.raises(ValueError, match="number of image"):
        model(input_ids=ids, attention_mask=batch["chosen_attention_mask"], labels=batch["chosen_labels"],
              pixel_values=batch["img_input_dict"]["pixel_values"])


# ------------------------------------------------------------------------------------------------------------ LoRA
PEFT = dict(r=8, lora_alpha=16, lora_dropout=0.0, target_modules="auto", bias="none")


def _lora_oracle_grads(W, cfg, batch, lora, emulate=EMU):
    leaves = {k: v.clone().requires_grad_(True) for k, v in lora["W"].items()}
    l2 = dict(lora)
    l2["W"] = leaves
    Wb = {k: v.bfloat16().float() for k, v in W.items()} if emulate else W
    loss, metrics = O.compute_loss(Wb, Wb, cfg, batch, cfg["beta"], emulate_bf16=emulate, lora=l2)
    loss.backward()
    return loss.detach(), metrics, {k: leaves[k].grad for k in leaves}


def test_dropout_mask_matches_restatement(gpu):
    from vlrlhf import _hip
    for seed, p, n in ((1, 0.05, 4096), ((5 << 40) + (3 << 16) + 8 * 7 + 2, 0.25, 1 << 16), (99, 0.0, 64)):
        m = torch.empty(n, dtype=torch.uint8, device=gpu)
        _hip.call("vlr_dropout_mask", m, n, p, seed)
        assert torch.equal(m.cpu(), O.dropout_mask(seed, n, p)), (seed, p)
    x = torch.randn(4096, device=gpu).bfloat16()
    y = torch.empty_like(x)
    _hip.call("vlr_dropout", x, y, 4096, 0.25, 1, 1.0, 0)
    keep = O.dropout_mask(1, 4096, 0.25).bool()
    exp = torch.where(keep, x.float().cpu() / 0.75, torch.zeros(())).bfloat16()
    assert torch.equal(y.cpu(), exp)
    acc = torch.ones_like(x)
    _hip.call("vlr_dropout", x, acc, 4096, 0.25, 1, 2.0, 1)
    exp2 = (1.0 + torch.where(keep, x.float().cpu() * (2.0 / 0.75), torch.zeros(()))).bfloat16()
    assert relmax(acc, exp2) < 1e-2
    with pytest.raises(ValueError):
        _hip.call("vlr_dropout", x, y, 4095, 0.25, 1, 1.0, 0)


@pytest.mark.parametrize("dropout", [0.0, 0.25])
def test_lora_step_matches_oracle(gpu, dropout):
    """peft_config path of the trainer: adapters on the seven decoder linears, frozen base, reference = adapters disabled."""
    from vlrlhf.models.Llava import LlavaForRL
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    lora = O.random_lora(cfg, r=8, alpha=16, seed=3, b_std=0.05, dropout=dropout)
    lora["W"] = {k: v.bfloat16().float() for k, v in lora["W"].items()}
    model = LlavaForRL.from_state_dict(cfg, W)
    pc = dict(PEFT, lora_dropout=dropout, seed=5)
    tr = make_trainer(model, None, cfg, peft_config=pc)
    assert tr.ref_model is None and tr.is_peft_model
    eng = model.engine
    eng.load_lora_state_dict(lora["W"])
    names = [n for n, _ in model.named_parameters()]
    assert len(names) == 2 * 7 * cfg["layers"] and all(".lora_" in n for n in names)
    base_before = eng.policy.flat.clone()
    eng.init_optimizer()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    seed = (5 << 40) + (eng._lora_calls << 16)           # the policy pass is the engine's latest adapter forward
    lora["seed"] = seed
    l16, m16, g16 = _lora_oracle_grads(W, cfg, batch, lora)
    within(f"llava.lora.loss.p{dropout}", abs(float(loss) - float(l16)), default=TOL_LOSS_BF16)
    # reference pass = base weights only: rewards are relative to the adapter-free policy
    logs = tr.log({"loss": float(loss)})
    within(f"llava.lora.margin.p{dropout}", abs(logs["rewards/margins"] - float(m16["rewards/margins"])), default=3e-2)
    named = dict(model.named_parameters())
    worst = 1.0
    for k, g in g16.items():
        hip = named[k.replace(".weight", ".default.weight")].grad
        assert tuple(hip.shape) == tuple(g.shape)
        c = cosine(hip, g)
        worst = min(worst, c)
        assert c > 0.97, (k, c)
        assert abs(float(hip.float().norm().cpu()) / float(g.norm()) - 1) < 0.08, k
    within(f"llava.lora.one_minus_worst_cosine.p{dropout}", 1.0 - worst, default=0.03)
    # optimizer: only the adapters move
    o = cfg["optim"]
    eng.optimizer_step(o["lr"], o["beta1"], o["beta2"], o["eps"], o["weight_decay"], o["max_grad_norm"])
    torch.cuda.synchronize()
    assert torch.equal(eng.policy.flat, base_before)
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in g16.values()))
    assert abs(eng.grad_norm() - total) < 0.05 * total
    state = {}
    Wl = {k: v.clone() for k, v in lora["W"].items()}
    gc = {k: v.clone() for k, v in g16.items()}
    O.clip_grad_norm_(gc, o["max_grad_norm"])
    O.adamw_step(Wl, gc, state, o["lr"], o["beta1"], o["beta2"], o["eps"], o["weight_decay"])
    new = eng.lora_state_dict()
    num = den = 0.0
    for k in Wl:
        du_h = new[k].float().cpu() - lora["W"][k]
        du_o = Wl[k] - lora["W"][k]
        num += float(((du_h - du_o) ** 2).sum())
        den += float((du_o ** 2).sum())
    assert math.sqrt(num / den) < 0.25, math.sqrt(num / den)      # first Adam step = lr * sign-like update, bf16 params


def test_lora_init_is_reference_and_merge(gpu):
    from vlrlhf.models.Llava import LlavaForRL
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    model = LlavaForRL.from_state_dict(cfg, W)
    tr = make_trainer(model, None, cfg, peft_config=dict(PEFT))
    # peft init (B = 0): the policy IS the reference -> loss = ln 2 exactly, rewards 0
    loss = tr.training_step(model, batch)
    assert abs(float(loss) - math.log(2.0)) < 1e-6
    eng = model.engine
    lora = O.random_lora(cfg, r=8, alpha=16, seed=11, b_std=0.05)
    eng.load_lora_state_dict(lora["W"])
    model.eval()
    with torch.no_grad():
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
        with model.disable_adapter():
            bc, br, _, _ = tr.concatenated_forward(model, batch)
    assert float((pc - bc).abs().max()) > 1e-2
    ref_lp = t(z, "policy_logps")          # fixture policy == these base weights
    assert float((torch.cat([bc, br]).cpu() - ref_lp).abs().max()) < TOL_LOGPS_FP32
    # merge_and_unload: a plain model on the merged weights reproduces the adapter forward
    merged = LlavaForRL.from_state_dict(cfg, {**W, **{k: v.float().cpu() for k, v in model.merge_and_unload().items()}})
    merged.eval()
    tr2 = make_trainer(merged, None, cfg, reference_free=True)
    with torch.no_grad():
        mc, mr, _, _ = tr2.concatenated_forward(merged, batch)
    assert float((mc.cpu() - pc.cpu()).abs().max()) < 0.3 and float((mr.cpu() - pr.cpu()).abs().max()) < 0.3
    with pytest.raises(NotImplementedError):
        LlavaForRL.from_state_dict(cfg, W).apply_lora(dict(PEFT, target_modules=["q_proj", "v_proj"]))
    with pytest.raises(ValueError):
        LlavaForRL.from_state_dict(cfg, W).apply_lora(dict(PEFT, r=6))


def test_precompute_ref_log_probs_prepass(gpu):
    """trl's precompute_ref_log_probs: one no-grad pass stores the reference log-probs on the dataset rows; the training
    steps then run without a reference forward and see the same loss as with a live reference model."""
    from vlrlhf.models.Llava import LlavaDPODataCollatorWithPadding, LlavaDPOTrainer, LlavaForRL
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    px = t(z, "batch.pixel_values")
    ds = [dict(r, img_path=px[i]) for i, r in enumerate(rows)]
    model = LlavaForRL.from_state_dict(cfg, W)
    o = cfg["optim"]
    args = SimpleNamespace(gradient_accumulation_steps=1, per_device_train_batch_size=2, learning_rate=o["lr"], adam_beta1=o["beta1"],
                           adam_beta2=o["beta2"], adam_epsilon=o["eps"], weight_decay=o["weight_decay"], max_grad_norm=o["max_grad_norm"],
                           seed=0, max_steps=2, logging_steps=1, lr_scheduler_type="constant")
    coll = LlavaDPODataCollatorWithPadding(pad_token_id=0, label_pad_token_id=-100)
    tr = LlavaDPOTrainer(model, None, cfg["beta"], 0, "sigmoid", args, coll, -100, 0, "keep_end", ds, None, None,
                         precompute_ref_log_probs=True)
    assert tr.ref_model is None
    tr.train()
    got = torch.tensor([[r["reference_chosen_logps"] for r in tr.train_dataset], [r["reference_rejected_logps"] for r in tr.train_dataset]])
    exp = t(z, "policy_logps").view(2, -1)             # the policy's initial weights are the reference
    assert float((got - exp).abs().max()) < TOL_LOGPS_FP32
    # first logged loss: policy == reference -> ln 2 (up to the bf16 noise between two evaluations of the same weights)
    assert abs(tr.log_history[0]["loss"] - math.log(2.0)) < 2e-3, tr.log_history[0]
    assert len(tr.log_history) == 2 and all(math.isfinite(h["loss"]) for h in tr.log_history)


def test_step_is_bit_reproducible(gpu):
    """two evaluations of the same step give bit-identical loss and gradients - including the embedding gradient, whose
    duplicate-token contributions are summed in position order (no atomics anywhere on the path)."""
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    # repeat tokens inside and across the sequences so that many positions share an embedding row
    b2 = dict(batch)
    for side in ("chosen", "rejected"):
        ids = batch[f"{side}_input_ids"].clone()
        ids[:, -6:] = ids[:, -7:-6]          # the tail is text (the <image> token sits in the prompt)
        b2[f"{side}_input_ids"] = ids
    outs = []
    for rep in range(2):
        model, ref = build(cfg, W, W_ref)
        tr = make_trainer(model, ref, cfg)
        loss = tr.training_step(model, b2)
        torch.cuda.synchronize()
        outs.append((float(loss), model.engine.grads.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    emb = dict(model.named_parameters())["language_model.model.embed_tokens.weight"].grad
    assert float(emb.float().abs().sum()) > 0


@pytest.mark.parametrize("lora", [False, True])
def test_gradient_checkpointing_is_bit_identical(gpu, lora):
    """--gradient_checkpointing True (reference dpo.py:99, every shipped script): the engine keeps only the layer inputs and re-runs each
    layer's forward before its backward - loss and EVERY gradient bit-identical to the run that keeps all activations (LoRA: with
    lora_dropout, whose counter-based masks the recompute regenerates)."""
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    outs = []
    for ckpt in (False, True):
        model, ref = build(cfg, W, W_ref)
        kw = dict(peft_config=dict(PEFT, lora_dropout=0.25, seed=5)) if lora else {}
        tr = make_trainer(model, None if lora else ref, cfg, **kw)
        if lora:
            for k, t in model.engine.lv.items():          # peft init has B = 0: give the adapters something to do
                if ".b_" in k:
                    t.copy_(torch.randn(t.shape, generator=torch.Generator().manual_seed(len(k))).mul(0.02))
        if ckpt:
            model.gradient_checkpointing_enable()
        assert model.is_gradient_checkpointing == ckpt
        model.train()
        loss = tr.training_step(model, batch)
        torch.cuda.synchronize()
        g = model.engine.lora_grads if lora else model.engine.grads
        outs.append((float(loss), g.clone()))
        if ckpt:          # the per-layer activation sets were never allocated
            assert not any(isinstance(k, tuple) and len(k) == 4 and k[0] == "policy" and isinstance(k[1], int) for k in model.engine._ws)
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].float().abs().sum()) > 0


def _shifted(batch, k):
    """a second / third batch of the same shapes: the text tail tokens rotated by k"""
    b = dict(batch)
    for side in ("chosen", "rejected"):
        ids = batch[f"{side}_input_ids"].clone()
        ids[:, -5:] = torch.roll(ids[:, -5:], k, dims=1)
        b[f"{side}_input_ids"] = ids
    return b


@pytest.mark.parametrize("lora", [False, True])
def test_reference_pipeline_is_bit_identical(gpu, lora):
    """VLDPOTrainer.prefetch_reference: the frozen reference forward of the NEXT batch issued before the optimizer step of the
    current one (so clip + AdamW run under it) gives the same losses, metrics and weights, bit for bit, as computing it inside
    its own step - full fine-tune (separate frozen copy) and LoRA (base weights with the adapters disabled)."""
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    o = cfg["optim"]
    runs = []
    for pipeline in (False, True):
        model, ref = build(cfg, W, W_ref)
        tr = make_trainer(model, None if lora else ref, cfg, **(dict(peft_config=dict(PEFT, seed=5)) if lora else {}))
        tr.ref_pipeline = pipeline
        eng = model.engine
        if lora:
            for k, t_ in eng.lv.items():
                if ".b_" in k:
                    t_.copy_(torch.randn(t_.shape, generator=torch.Generator().manual_seed(len(k))).to(t_) * 0.05)
        eng.init_optimizer()
        bs = [tr._prepare_inputs(_shifted(batch, k)) for k in range(3)]
        losses = []
        for i in range(4):
            losses.append(tr.training_step(model, bs[i % 3]))
            assert tr._ref_pending is None                     # a prefetched result is consumed by the step it was made for
            tr.prefetch_reference(bs[(i + 1) % 3])
            assert (tr._ref_pending is not None) == pipeline
            eng.optimizer_step(o["lr"], o["beta1"], o["beta2"], o["eps"], o["weight_decay"], o["max_grad_norm"])
        torch.cuda.synchronize()
        margins = [float(x) for x in tr._stored_metrics["train"]["rewards/margins"]]
        flat = (eng.lora_flat if lora else eng.policy.flat).clone()
        runs.append(([float(x) for x in losses], margins, flat))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert runs[0][1] == runs[1][1]
    assert torch.equal(runs[0][2], runs[1][2])
    assert len(set(runs[0][0])) > 1            # the batches differ and the weights move


def test_training_trajectory_tracks_oracle(gpu):
    """four optimizer steps on the same batch with a large learning rate (the loss must MOVE): the HIP trajectory follows the
    oracle trajectory computed the way the HIP path stores things - fp32 master weights updated by the restated clip + AdamW,
    forward / backward on their bf16 rounding with bf16-rounded activations.  (Adam's first steps are sign-like, so tiny
    gradient elements flip between any two arithmetic paths: the per-step tolerance is 3e-2 on a 0.69 -> 0.15 trajectory.)"""
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    o = dict(cfg["optim"], lr=2e-4)
    model, ref = build(cfg, W, W_ref)
    tr = make_trainer(model, ref, cfg)
    eng = model.engine
    eng.init_optimizer()
    hip_losses = []
    for _ in range(4):
        eng.zero_grad()
        hip_losses.append(float(tr.training_step(model, batch)))
        eng.optimizer_step(o["lr"], o["beta1"], o["beta2"], o["eps"], o["weight_decay"], o["max_grad_norm"])
    torch.cuda.synchronize()
    names = O.trainable_names(W)
    master = {k: v.bfloat16().float() for k, v in W.items()}           # the engine's master copy starts from the bf16 weights
    Wr16 = {k: v.bfloat16().float() for k, v in W_ref.items()}
    state, ora_losses = {}, []
    for _ in range(4):
        leaves = {k: master[k].bfloat16().float().requires_grad_(True) for k in names}
        Wp = {k: v.bfloat16().float() for k, v in master.items()}
        Wp.update(leaves)
        loss, _ = O.compute_loss(Wp, Wr16, cfg, batch, cfg["beta"], emulate_bf16=EMU)
        loss.backward()
        grads = {k: leaves[k].grad for k in names if leaves[k].grad is not None}
        O.clip_grad_norm_(grads, o["max_grad_norm"])
        with torch.no_grad():
            O.adamw_step(master, grads, state, o["lr"], o["beta1"], o["beta2"], o["eps"], o["weight_decay"])
        ora_losses.append(float(loss))
    assert ora_losses[0] - ora_losses[-1] > 0.3, ora_losses             # the trajectory does move
    assert all(a > b for a, b in zip(hip_losses, hip_losses[1:])), hip_losses
    for h, r in zip(hip_losses, ora_losses):
        assert abs(h - r) < 3e-2, (hip_losses, ora_losses)
    new = model.state_dict()
    num = den = dot = 0.0
    for k in names:
        dh = new[k].float().cpu() - W[k].bfloat16().float()
        dr = master[k] - W[k].bfloat16().float()
        num += float((dh * dh).sum()); den += float((dr * dr).sum()); dot += float((dh * dr).sum())
    cos = dot / math.sqrt(num * den)
    assert cos > 0.9 and 0.8 < math.sqrt(num / den) < 1.25, (cos, math.sqrt(num / den))


def test_evaluate_reports_eval_metrics(gpu):
    """HF-style evaluation on the DPO objective: eval_loss + the eight eval_ metrics, no gradients, policy untouched."""
    from vlrlhf.models.Llava import LlavaDPODataCollatorWithPadding, LlavaDPOTrainer
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    px = t(z, "batch.pixel_values")
    ds = [dict(r, img_path=px[i]) for i, r in enumerate(rows)]
    model, ref = build(cfg, W, W_ref)
    args = SimpleNamespace(gradient_accumulation_steps=1, per_device_train_batch_size=2, per_device_eval_batch_size=2, seed=0)
    tr = LlavaDPOTrainer(model, ref, cfg["beta"], 0, "sigmoid", args, LlavaDPODataCollatorWithPadding(pad_token_id=0, label_pad_token_id=-100),
                         -100, 0, "keep_end", None, ds, None)
    before = model.engine.policy.flat.clone()
    out = tr.evaluate()
    assert abs(out["eval_loss"] - float(z["loss_mean_sigmoid"])) < TOL_LOSS_FP32
    for k in ("rewards/chosen", "rewards/rejected", "rewards/accuracies", "rewards/margins", "logps/chosen", "logps/rejected",
              "logits/chosen", "logits/rejected"):
        assert f"eval_{k}" in out, k
    assert torch.equal(before, model.engine.policy.flat) and model.training
    # generate_during_eval (trl 0.8.1 evaluation_loop): one random eval batch is sampled from policy and reference and logged as a table
    tr.generate_during_eval = True
    tr.max_length = max(len(r["prompt_input_ids"]) for r in rows) + 3
    tr.tokenizer = SimpleNamespace(pad_token_id=0, eos_token_id=10 ** 6,
                                   batch_decode=lambda t_, skip_special_tokens=True: [" ".join(map(str, (r.tolist() if hasattr(r, "tolist") else r))) for r in t_])
    n_before = len(tr.log_history)
    out2 = tr.evaluate()
    assert "eval_loss" in out2
    games = [e for e in tr.log_history[n_before:] if "game_log" in e]
    assert len(games) == 1 and games[0]["game_log"]["columns"] == ["Prompt", "Policy", "Ref Model"] and len(games[0]["game_log"]["rows"]) == 2
    assert torch.equal(before, model.engine.policy.flat) and model.training


def test_generate_and_get_batch_samples(gpu):
    """reference base/trainer.py:310-360 (`generate_during_eval`): `model.generate` on the left-padded prompts of the batch - greedy tokens
    judged against the fp32 oracle's next-token logits on the SAME prefix (teacher-forced with the HIP tokens: the chosen token is the
    oracle's argmax up to bf16 noise), sampling is reproducible under a generator, finished rows keep receiving the pad id, and
    `get_batch_samples` returns decoded policy / reference strings padded to max_length."""
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    model, ref = build(cfg, W, W_ref)
    pix = batch["img_input_dict"]["pixel_values"]
    ids0, m0 = batch["prompt_input_ids"], batch["prompt_attention_mask"]
    assert int(m0[1, 0]) == 0 and int(m0[1, -1]) == 1                         # the fixture's prompts are left-padded
    new = 5
    out = model.generate(input_ids=ids0.to(gpu), attention_mask=m0.to(gpu), max_new_tokens=new, do_sample=False, pad_token_id=0,
                         eos_token_id=10 ** 6, pixel_values=pix.to(gpu)).cpu()
    assert out.shape == (2, ids0.shape[1] + new) and torch.equal(out[:, :ids0.shape[1]], ids0)
    ids, mask = ids0.clone(), m0.clone()
    for k in range(new):
        logits, _, aux = O.llava_forward(W, cfg, ids, mask, torch.full_like(ids, -100), pix, dedupe_images=False)
        for b in range(2):
            last = int(torch.nonzero(aux["mask"][b]).max())
            row = logits[b, last]
            tok = int(out[b, ids0.shape[1] + k])
            assert float(row.max() - row[tok]) <= 0.06 * float(row.max() - row.min()), (k, b, tok, int(row.argmax()))
        ids = torch.cat([ids, out[:, ids0.shape[1] + k: ids0.shape[1] + k + 1]], 1)
        mask = torch.cat([mask, torch.ones(2, 1, dtype=mask.dtype)], 1)
    # sampling: reproducible under a generator, every sampled token inside the top-k set of the step
    g = lambda: torch.Generator(device=gpu).manual_seed(7)      # noqa: E731
    kw = dict(input_ids=ids0.to(gpu), attention_mask=m0.to(gpu), max_length=ids0.shape[1] + 4, do_sample=True, top_k=5, pad_token_id=0,
              eos_token_id=10 ** 6, pixel_values=pix.to(gpu))
    s1, s2 = model.generate(generator=g(), **kw).cpu(), model.generate(generator=g(), **kw).cpu()
    assert torch.equal(s1, s2) and s1.shape[1] == ids0.shape[1] + 4
    # eos: a row that emits it is finished and receives the pad id from then on
    first = int(out[0, ids0.shape[1]])
    e = model.generate(input_ids=ids0.to(gpu), attention_mask=m0.to(gpu), max_new_tokens=4, do_sample=False, pad_token_id=191,
                       eos_token_id=first, pixel_values=pix.to(gpu)).cpu()
    assert int(e[0, ids0.shape[1]]) == first and bool((e[0, ids0.shape[1] + 1:] == 191).all())
    # the trainer's wrapper
    tr = make_trainer(model, ref, cfg)
    tr.max_length = ids0.shape[1] + 3
    tr.tokenizer = SimpleNamespace(pad_token_id=0, batch_decode=lambda t_, skip_special_tokens=True: [" ".join(map(str, r.tolist())) for r in t_])
    pol, refs = tr.get_batch_samples(model, {k: (v.to(gpu) if isinstance(v, torch.Tensor) else v) for k, v in batch.items() if k != "img_input_dict"}
                                     | dict(img_input_dict=dict(pixel_values=pix.to(gpu))))
    assert len(pol) == len(refs) == 2 and all(len(p_.split()) == tr.max_length for p_ in pol + refs)
