"""ONE decoder layer at the REAL widths of the three other model families against the CPU oracle - the fp32 oracle and its model of what
the HIP path rounds to bf16 for the loss / log-probs, fp32 autograd for the gradients (VERDICT r05 missing 2: until round 6 these widths
were met only by the property tests of tests/test_hip_fullsize*.py, which a wrong-but-symmetric tile path passes).

  * LLaVA-Next-Mistral : H 4096, I 14336, 32 query / 8 K-V heads (grouped-query), V 32064, anyres tiles, variable-length merge
  * Qwen-VL            : H 4096, I 11008, biased fused c_attn, resampler output 4096, lm-head over a 23 936-column slice of V = 151 936
                         (both = 128 mod 256: the same half-filled last tile column)
  * InternLM-XComposer2: H 4096, I 14336, fused grouped-query wqkv (32 / 8), PLoRA r 256 on the image rows + peft LoRA r 64 (the shipped
                         configuration of scripts/dpo_internlmxc2vl7b.sh), lm-head over a 15 744-column slice of V = 92 544 (= 128 mod 256)
Two lengths each: `short` (a few dozen tokens per sequence, as test_true_width_layer_loss: north_star's loss rtol 1e-3 against the
emulation) and `tiles` (>= 3072 token rows: the persistent 256x256 GEMMs with their fused epilogues and adapter segments, not the
small-grid kernel; judged on the per-sequence log-probs against the floor model, see _judge).  Gradient cosines > 0.99 at both.
Reference anchors: /root/reference src/vlrlhf/models/QwenVL/modeling_qwen.py:153-188,310-323, InternLMXC2/modeling_internlm2.py:318-369,
InternLMXC2/build_mlp.py:158-203, LlavaNext/__init__.py:205-265.  The vision towers keep the fixture widths (frozen, outside the decoder).
Weights are random (seeded) at std 0.02, policy != reference.  Margins: profiles/r06_parity_margins.txt (VLR_MARGINS=<file> pytest -m gpu)."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import internlm_oracle as IL  # noqa: E402  (checker only)
from oracle import llava_dpo_oracle as O  # noqa: E402  (checker only)
from oracle import qwenvl_oracle as Q  # noqa: E402  (checker only)
from tests.golden_util import load_case, within  # noqa: E402
from tests.test_hip_e2e import cosine  # noqa: E402


def _is_gain(name):
    return name.endswith(".weight") and any(s in name for s in ("norm", ".ln_", "ln_f", "ln_q", "ln_kv", "ln_post", "layernorm"))


def wide_weights(cfg, W0, seed, regenerate=None, std=0.02):
    """checkpoint-named weights for `cfg`: every tensor of the trainable layout drawn at its new shape (gains 1 + 0.05 n, biases and matrices
    std n), the frozen vision tower taken from the small fixture, `regenerate` = {name: shape} for frozen tensors whose shape follows the
    decoder width.  Everything bf16-representable (the engine stores bf16)."""
    from vlrlhf.engine import ParamLayout
    lay = ParamLayout(cfg)
    g = torch.Generator().manual_seed(seed)
    W = {}

    def draw(name, shape):
        n = torch.randn(*shape, generator=g)
        return ((1.0 + 0.05 * n) if _is_gain(name) else std * n).bfloat16().float()
    for hf, name, r0, rows in lay.hf_names():
        shp = tuple(lay.shape[name])
        W[hf] = draw(hf, ((rows,) + shp[1:]) if len(shp) > 1 else shp)
    for k, shape in (regenerate or {}).items():
        W[k] = draw(k, shape)
    for k, v in W0.items():
        if k not in W:
            W[k] = v
    return W, lay


def perturbed(W, lay, seed, rel=0.05):
    """the reference model: the policy's decoder matrices moved by rel x their mean magnitude"""
    g = torch.Generator().manual_seed(seed)
    names = {hf for hf, _, _, _ in lay.hf_names()}
    out = {}
    for k, v in W.items():
        if k in names and v.dim() == 2 and not _is_gain(k):
            out[k] = (v + rel * v.abs().mean() * torch.randn(v.shape, generator=g)).bfloat16().float()
        else:
            out[k] = v
    return out


def _args():
    return SimpleNamespace(gradient_accumulation_steps=1)


def _judge(tag, short, loss, hip_lp, emu_lp, f32_lp, l16, l32, beta=0.1):
    """The per-sequence log-prob SUMS are -170 ... -450 (`short`) or -3000 ... -6000 (`tiles`), and any two bf16 roundings of the step (the
    HIP kernels, the oracle's model of them) sit 0.04 - 0.08 resp. 0.2 - 0.6 from fp32 AND from each other (5e-5 ... 3e-4 relative) - the
    DPO loss is beta / 2 times a difference of four such sums, so its error is a SAMPLE of sigma = beta x rms / sqrt(pairs) = 3e-3 ... 1.6e-2
    for ANY bf16-MFMA pipeline (the long LLaVA-Next case landed 2e-6 from fp32 and 2e-2 from the emulation in the same run).  As in
    tests/test_hip_depth.py the parity statement is therefore on the log-probs: the HIP path is no further from fp32 than the oracle's
    model of its rounding allows (max and rms), the loss within 3 sigma; north_star's rtol 1e-3 against the emulation is asserted where
    sigma permits it."""
    import math
    rel16 = abs(float(loss) - float(l16)) / abs(float(l16))
    rel32 = abs(float(loss) - float(l32)) / abs(float(l32))
    e_hip, e_emu = (hip_lp - f32_lp).double(), (emu_lp - f32_lp).double()
    mx, rms = float(e_hip.abs().max()), float(e_hip.pow(2).mean().sqrt())
    fmx, frms = float(e_emu.abs().max()), float(e_emu.pow(2).mean().sqrt())
    sigma = beta * frms / math.sqrt(hip_lp.numel() // 4)
    print(f"{tag}{' (short)' if short else ''}: loss hip {float(loss):.6f} oracle rounding what the HIP path rounds {float(l16):.6f} oracle fp32 {float(l32):.6f} "
          f"rel {rel16:.2e} / {rel32:.2e} | log-prob sums ~{float(f32_lp.mean()):.0f}: HIP - fp32 max {mx:.3f} rms {rms:.3f}, emulation - fp32 max {fmx:.3f} "
          f"rms {frms:.3f} | sigma(loss) {sigma:.2e}")
    name = f"true_width.{tag}.{'short' if short else 'tiles'}"
    # the floor factor is 2 here (1.3 in tests/test_hip_depth.py): both rms are estimated from EIGHT sums - the ratio of two such sample rms
    # scatters over [0.5, 2] at 95 % for identical distributions (measured on one box: 0.5 ... 1.7 over the six cases); a tile or edge bug moves
    # a log-prob sum by tens to hundreds and the gradient cosines (asserted > 0.99, measured > 0.9997) far below 0.99
    within(name + ".dlogp_max_over_floor", mx / (2.0 * fmx + 0.02), default=1.0)
    within(name + ".dlogp_rms_over_floor", rms / (2.0 * frms + 0.01), default=1.0)
    within(name + ".loss_err_over_3sigma", abs(float(loss) - float(l32)) / (3 * sigma), default=1.0)
    # north_star's rtol 1e-3 against the emulation, where the sampling noise of the two roundings allows it (short LLaVA-Next: 9.1e-4);
    # else 3 sigma of the difference of two samples
    within(name + ".loss_rel_vs_emulation", rel16 / max(1e-3, 3 * math.sqrt(2.0) * sigma / abs(float(l16))), default=1.0)


LENGTHS = [pytest.param(True, id="short"), pytest.param(False, id="tiles")]


@pytest.mark.parametrize("short", LENGTHS)
def test_llava_next_mistral_true_width_layer(short):
    from vlrlhf.models.LlavaNext import LlavaNextDPOTrainer, LlavaNextForRL
    from vlrlhf.utils.synthetic import synthetic_batch_anyres
    _, cfg0, W0, _, _, _ = load_case("llavanext_small")
    cfg = dict(cfg0, hidden=4096, inter=14336, heads=32, kv_heads=8, layers=1, vocab=32064, image_token=32000, model_pad_token_id=32001)
    W, lay = wide_weights(cfg, W0, seed=21)
    W_ref = perturbed(W, lay, seed=22)
    batch = synthetic_batch_anyres(2, 48 if short else 1100, cfg["image_token"], 32000, cfg["image_size"], seed=23, image_hw=(40, 75),
                                   grid_pinpoints=cfg["image_grid_pinpoints"], ragged=True)
    model = LlavaNextForRL.from_state_dict(cfg, W)
    ref = model.create_reference_model()
    ref.weights.load_state_dict(W_ref)
    tr = LlavaNextDPOTrainer(model, ref, 0.1, 0, "sigmoid", _args(), None, -100, 0)
    with torch.no_grad():
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
    hip_lp = torch.cat([pc, pr, rc, rr]).cpu()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    if not short:
        assert 4 * int(model._last_ctx["S"]) >= 3072, model._last_ctx["S"]      # >= 192 tiles of 256 x 256 at N = 4096: the persistent kernels
    with torch.no_grad():
        a, b, _, _ = O.concatenated_forward(W, cfg, batch, "sigmoid", O.HIP_ROUNDING)
        c, d, _, _ = O.concatenated_forward(W_ref, cfg, batch, "sigmoid", O.HIP_ROUNDING)
    emu_lp = torch.cat([a, b, c, d])
    l16 = O.dpo_loss(a, b, c, d, 0.1)[0].mean()
    leaves = {k: v.clone().requires_grad_(not k.startswith("vision_tower.")) for k, v in W.items()}
    a, b, _, _ = O.concatenated_forward(leaves, cfg, batch, "sigmoid", False)
    with torch.no_grad():
        c, d, _, _ = O.concatenated_forward(W_ref, cfg, batch, "sigmoid", False)
    l32 = O.dpo_loss(a, b, c, d, 0.1)[0].mean()
    l32.backward()
    f32_lp = torch.cat([a.detach(), b.detach(), c, d])
    named = dict(model.named_parameters())
    worst = 1.0
    for name in ("language_model.lm_head.weight", "language_model.model.layers.0.self_attn.q_proj.weight",
                 "language_model.model.layers.0.self_attn.k_proj.weight", "language_model.model.layers.0.self_attn.v_proj.weight",
                 "language_model.model.layers.0.self_attn.o_proj.weight", "language_model.model.layers.0.mlp.gate_proj.weight",
                 "language_model.model.layers.0.mlp.down_proj.weight", "language_model.model.layers.0.input_layernorm.weight",
                 "multi_modal_projector.linear_2.weight", "image_newline"):
        cs = cosine(named[name].grad, leaves[name].grad)
        worst = min(worst, cs)
        assert cs > 0.99, (name, cs)
    within(f"true_width.llava_next_mistral.{'short' if short else 'tiles'}.one_minus_worst_cosine", 1.0 - worst, default=1e-2)
    _judge("llava_next_mistral", short, loss, hip_lp, emu_lp, f32_lp, l16, l32.detach())


@pytest.mark.parametrize("short", LENGTHS)
def test_qwen_vl_true_width_layer(short):
    from vlrlhf.models.QwenVL import QwenVLDPOTrainer, QwenVLForRL
    from vlrlhf.utils.synthetic import synthetic_batch_qwen
    _, cfg0, W0, _, _, _ = load_case("qwenvl_small")
    V = 23936                                                  # = 151 936 - 500 x 256: the checkpoint's last (half) tile column, fewer whole ones
    E = 4096
    # (36 = 6 x 6 resampler queries: an <img> slot must hold the 18 bytes of the synthetic image path; the query grid is a square)
    cfg = dict(cfg0, hidden=4096, inter=11008, heads=32, layers=1, vocab=V, image_start_id=V - 100, pad_token_id=V - 90, im_start_id=V - 89,
               im_end_id=V - 88, visual=dict(cfg0["visual"], output_dim=E, n_queries=36))
    vp = "transformer.visual."
    W, lay = wide_weights(cfg, W0, seed=31, regenerate={vp + "proj": (E, E), vp + "attn_pool.pos_embed": (cfg["visual"]["n_queries"], E),
                                                        vp + "ln_post.weight": (E,), vp + "ln_post.bias": (E,)})
    W_ref = perturbed(W, lay, seed=32)
    batch = synthetic_batch_qwen(2, 96 if short else 800, cfg, seed=33)
    model = QwenVLForRL.from_state_dict(cfg, W)
    ref = model.create_reference_model()
    ref.weights.load_state_dict(W_ref)
    tr = QwenVLDPOTrainer(model, ref, 0.1, 0, "sigmoid", _args(), None, -100, cfg["pad_token_id"])
    model.engine.init_optimizer()
    with torch.no_grad():
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
    hip_lp = torch.cat([pc, pr, rc, rr]).cpu()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    if not short:
        assert 4 * int(model._last_ctx["S"]) >= 3072
    qb = dict(batch, pixel_values=batch["img_input_dict"]["pixel_values"])
    emu = O.HIP_ROUNDING | {"vit"}
    with torch.no_grad():
        a, b, _, _ = Q.concatenated_forward(W, cfg, qb, "sigmoid", emu)
        c, d, _, _ = Q.concatenated_forward(W_ref, cfg, qb, "sigmoid", emu)
    emu_lp = torch.cat([a, b, c, d])
    l16 = O.dpo_loss(a, b, c, d, 0.1)[0].mean()
    frozen = lambda k: k.startswith(vp) and "attn_pool" not in k       # noqa: E731  (ViT trunk frozen; resampler + language model trained)
    leaves = {k: v.clone().requires_grad_(not frozen(k)) for k, v in W.items()}
    a, b, _, _ = Q.concatenated_forward(leaves, cfg, qb, "sigmoid", False)
    with torch.no_grad():
        c, d, _, _ = Q.concatenated_forward(W_ref, cfg, qb, "sigmoid", False)
    l32 = O.dpo_loss(a, b, c, d, 0.1)[0].mean()
    l32.backward()
    f32_lp = torch.cat([a.detach(), b.detach(), c, d])
    named = dict(model.named_parameters())
    worst = 1.0
    for name in ("lm_head.weight", "transformer.h.0.attn.c_attn.weight", "transformer.h.0.attn.c_attn.bias", "transformer.h.0.attn.c_proj.weight",
                 "transformer.h.0.mlp.w1.weight", "transformer.h.0.mlp.w2.weight", "transformer.h.0.mlp.c_proj.weight", "transformer.h.0.ln_1.weight",
                 "transformer.visual.attn_pool.kv_proj.weight"):
        cs = cosine(named[name].grad, leaves[name].grad)
        worst = min(worst, cs)
        assert cs > (0.98 if "attn_pool" in name else 0.99), (name, cs)
    within(f"true_width.qwen_vl.{'short' if short else 'tiles'}.one_minus_worst_cosine", 1.0 - worst, default=2e-2)
    _judge("qwen_vl", short, loss, hip_lp, emu_lp, f32_lp, l16, l32.detach())


@pytest.mark.parametrize("short", LENGTHS)
def test_internlm_xcomposer2_true_width_layer_lora_over_plora(short):
    from vlrlhf.models.InternLMXC2 import InternLMXC2DPOTrainer, InternLMXC2ForRL
    from vlrlhf.utils.synthetic import synthetic_batch
    _, cfg0, W0, _, _, _ = load_case("internlmxc2_small")
    V = 15744                                                  # = 92 544 - 300 x 256
    cfg = dict(cfg0, hidden=4096, inter=14336, heads=32, kv_heads=8, layers=1, vocab=V, plora_dropout=0.0)
    W, lay = wide_weights(cfg, W0, seed=41)
    batch = synthetic_batch(2, 64 if short else 1500, cfg["image_token"], V - 8, cfg["image_size"], seed=43, ragged=True)
    for k in ("chosen", "rejected", "prompt"):                 # the random ids must not repeat the <image> id
        ids = batch[f"{k}_input_ids"]
        stray = ids == cfg["image_token"]
        stray[:, 4] = False
        ids[stray] = cfg["image_token"] + 1
        if f"{k}_labels" in batch:
            lab = batch[f"{k}_labels"]
            lab[stray & (lab != -100)] = cfg["image_token"] + 1
    pc_ = dict(r=64, lora_alpha=64, lora_dropout=0.0, target_modules="auto", bias="none", seed=5)
    model = InternLMXC2ForRL.from_state_dict(cfg, W)
    tr = InternLMXC2DPOTrainer(model, None, 0.1, 0, "sigmoid", _args(), None, -100, cfg["model_pad_token_id"], peft_config=pc_)
    lora = IL.random_lora(cfg, r=64, alpha=64, seed=3, b_std=0.02, dropout=0.0)
    lora["W"] = {k: v.bfloat16().float() for k, v in lora["W"].items()}
    eng = model.engine
    eng.load_lora_state_dict(lora["W"])
    assert eng.lora_fused and eng.resid_f32                    # the two-adapter C layer passes (vlr_decoder_layer_fwd_lora2 / bwd_lora2)
    eng.init_optimizer()
    with torch.no_grad():
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
        with tr.null_ref_context():
            rc, rr, _, _ = tr.concatenated_forward(model, batch)
    hip_lp = torch.cat([pc, pr, rc, rr]).cpu()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    if not short:
        assert 4 * int(model._last_ctx["S"]) >= 3072
    lora["seed"] = (5 << 40) + (eng._lora_calls << 16)
    with torch.no_grad():
        a, b, _, _ = IL.concatenated_forward(W, cfg, batch, "sigmoid", O.HIP_ROUNDING, lora=dict(lora, W=lora["W"]))
        c, d, _, _ = IL.concatenated_forward(W, cfg, batch, "sigmoid", O.HIP_ROUNDING)
    emu_lp = torch.cat([a, b, c, d])
    l16 = O.dpo_loss(a, b, c, d, 0.1)[0].mean()
    Wl = {k: v.clone().requires_grad_(True) for k, v in lora["W"].items()}
    a, b, _, _ = IL.concatenated_forward(W, cfg, batch, "sigmoid", False, lora=dict(lora, W=Wl))
    with torch.no_grad():
        c, d, _, _ = IL.concatenated_forward(W, cfg, batch, "sigmoid", False)
    l32 = O.dpo_loss(a, b, c, d, 0.1)[0].mean()
    l32.backward()
    f32_lp = torch.cat([a.detach(), b.detach(), c, d])
    eng.lv, keep = eng.lgv, eng.lv                             # adapter gradients under checkpoint names / row order
    try:
        gsd = eng.lora_state_dict()
    finally:
        eng.lv = keep
    worst = 1.0
    for k, v in Wl.items():
        cs = cosine(gsd[k], v.grad)
        worst = min(worst, cs)
        assert cs > 0.99, f"{k}: cosine {cs:.4f}"
    within(f"true_width.internlm_xc2_lora_over_plora.{'short' if short else 'tiles'}.one_minus_worst_cosine", 1.0 - worst, default=1e-2)
    _judge("internlm_xc2_lora_over_plora", short, loss, hip_lp, emu_lp, f32_lp, l16, l32.detach())
