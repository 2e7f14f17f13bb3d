"""Size-independent properties of the DPO step at FULL size for the Qwen-VL-Chat (BASELINE.json configs[2]) and
InternLM-XComposer2-VL-7B (configs[4]) models - each in its own test so that the previous 7-9 B model is released first.
Needs a real MI355X with ~250 GB free:  pytest -m gpu"""
import math
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def _permuted(batch, perm):
    b2 = {}
    for k, v in batch.items():
        if k.startswith("_"):                     # per-batch memos of the trainer (keyed on the original tensors)
            continue
        if isinstance(v, torch.Tensor):
            b2[k] = v[perm]
        elif isinstance(v, dict):
            b2[k] = {kk: vv[perm] for kk, vv in v.items() if isinstance(vv, torch.Tensor)}
        elif isinstance(v, list):
            b2[k] = [v[i] for i in perm]
    return b2


def test_qwen_vl_chat_full_size_lora_properties():
    """the shipped configuration (scripts/dpo_qwenvl.sh): LoRA r 64 on c_attn / attn.c_proj / w1 / w2 of the 7.7 B decoder over the frozen
    1.9 B vision tower (48 layers, head_dim 104, resampler).  peft init (B = 0) => the policy IS the reference: loss = ln 2, yet the
    adapter gradient is finite and non-zero; a permutation of the pairs permutes the log-probs."""
    if not torch.cuda.is_available() or torch.cuda.mem_get_info()[1] < 250 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    from vlrlhf.models.QwenVL import QWEN_VL_CHAT, QwenVLDPOTrainer, QwenVLForRL
    from vlrlhf.utils.synthetic import init_hashed_qwen, synthetic_batch_qwen
    cfg = dict(QWEN_VL_CHAT)
    model = QwenVLForRL(cfg)
    init_hashed_qwen(model, seed=0, std=0.02, with_reference=False)
    tr = QwenVLDPOTrainer(model, None, 0.1, 0, "sigmoid", SimpleNamespace(gradient_accumulation_steps=1), None, -100, cfg["pad_token_id"],
                          peft_config=dict(r=64, lora_alpha=16, lora_dropout=0.05, target_modules="auto", bias="none", seed=1))
    batch = tr._prepare_inputs(synthetic_batch_qwen(4, 1024, cfg, seed=9))      # configs[2]: per-device batch 4, max_length 1024, lora_dropout 0.05
    model.engine.init_optimizer()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    assert model._last_ctx["S"] == 1024 and int(model._last_ctx["img_map"].sum()) == 8 * 256
    assert abs(float(loss) - math.log(2.0)) < 1e-6, float(loss)
    model.engine.optimizer_step(1e-5, 0.9, 0.98, 1e-6, 0.05, 1.0)
    norm = model.engine.grad_norm()
    assert math.isfinite(norm) and norm > 1e-6, norm
    model.eval()
    with torch.no_grad(), tr.null_ref_context():
        c1, r1, _, _ = tr.concatenated_forward(model, batch)
        c2, r2, _, _ = tr.concatenated_forward(model, _permuted(batch, [1, 0, 3, 2]))
    torch.cuda.synchronize()
    assert float((c1[[1, 0, 3, 2]] - c2).abs().max()) < 2e-3 * float(c1.abs().max())
    assert float((r1[[1, 0, 3, 2]] - r2).abs().max()) < 2e-3 * float(r1.abs().max())
    del model, tr
    torch.cuda.empty_cache()


def test_internlm_xcomposer2_7b_full_size_properties():
    """full fine-tune of the InternLM2-7B decoder incl. its PLoRA pairs (32 / 8 grouped-query heads, I = 14336, 1225 image rows per
    sequence), PLoRA dropout off: identical reference => ln 2, finite non-zero gradients for base AND PLoRA weights, the frozen projector
    untouched by the update, pair permutation."""
    if not torch.cuda.is_available() or torch.cuda.mem_get_info()[1] < 250 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    from vlrlhf.models.InternLMXC2 import INTERNLM_XC2_VL_7B, InternLMXC2DPOTrainer, InternLMXC2ForRL
    from vlrlhf.utils.synthetic import init_random_model, synthetic_batch
    cfg = dict(INTERNLM_XC2_VL_7B, plora_dropout=0.0)
    model = InternLMXC2ForRL(cfg)
    ref = init_random_model(model, seed=0, std=0.02, policy_delta=0.0)
    tr = InternLMXC2DPOTrainer(model, ref, 0.1, 0, "sigmoid", SimpleNamespace(gradient_accumulation_steps=1), None, -100, cfg["model_pad_token_id"])
    # configs[4]: per-device batch 4; T = 1024 -> S = 2248 (the script's max_length 2048 needs activation recompute for a full fine-tune
    # of this 8.6 B model; the composed PLoRA layer keeps its activations)
    batch = tr._prepare_inputs(synthetic_batch(4, 1024, cfg["image_token"], 32000, cfg["image_size"], seed=9, ragged=True))
    eng = model.engine
    eng.init_optimizer()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    assert int(model._last_ctx["extra"]["R"]) == 8 * 1225
    assert abs(float(loss) - math.log(2.0)) < 1e-6, float(loss)
    for k in ("l0.pa_qkv", "l0.pb_gu", "l30.pb_o", "l5.wqkv", "l31.wdown"):
        g = eng.gv[k].float()
        assert torch.isfinite(g).all() and (float(g.abs().max()) > 0) == (k != "never"), k
    proj_before = eng.policy.v["proj.w2"].clone()
    eng.optimizer_step(1e-6, 0.9, 0.95, 1e-6, 0.1, 1.0)
    norm = eng.grad_norm()
    assert math.isfinite(norm) and norm > 1e-4, norm
    assert torch.equal(eng.policy.v["proj.w2"], proj_before)
    model.eval()
    with torch.no_grad():
        c1, r1, _, _ = tr.concatenated_forward(ref, batch)
        c2, r2, _, _ = tr.concatenated_forward(ref, _permuted(batch, [1, 0, 3, 2]))
    torch.cuda.synchronize()
    assert float((c1[[1, 0, 3, 2]] - c2).abs().max()) < 2e-3 * float(c1.abs().max())
    assert float((r1[[1, 0, 3, 2]] - r2).abs().max()) < 2e-3 * float(r1.abs().max())
    del model, ref, tr
    torch.cuda.empty_cache()


def test_internlm_xcomposer2_7b_full_size_lora_properties():
    """the SHIPPED configuration at full width (scripts/dpo_internlmxc2vl7b.sh: --use_lora True - peft LoRA r 64 / alpha 64 / dropout 0.05 stacked
    on the frozen PLoRA r 256 of models/InternLMXC2/build_mlp.py:158-203): the two-adapter layer passes
    (vlr_decoder_layer_fwd_lora2 / bwd_lora2: the [B_lora 64 | B_plora 256] K segment at H = 4096 / I = 14336, the PLoRA row-set skips)
    at 7B widths.  peft init (lora_B = 0) => the policy IS the reference although PLoRA is active in both passes: loss = ln 2 with a finite
    NON-zero adapter gradient; the update moves only the adapters (base + PLoRA weights untouched); a permutation of the pairs permutes the
    log-probs (PLoRA's row set moves with the images)."""
    if not torch.cuda.is_available() or torch.cuda.mem_get_info()[1] < 250 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    from vlrlhf.models.InternLMXC2 import INTERNLM_XC2_VL_7B, InternLMXC2DPOTrainer, InternLMXC2ForRL
    from vlrlhf.utils.synthetic import init_random_model, synthetic_batch
    cfg = dict(INTERNLM_XC2_VL_7B, plora_dropout=0.0)      # (PLoRA's own dropout draws a fresh mask per pass: with it on, policy and reference differ and ln 2 is not exact)
    model = InternLMXC2ForRL(cfg)
    ref = init_random_model(model, seed=0, std=0.02, policy_delta=0.0)
    del ref                                                # under LoRA the frozen base doubles as the reference (adapters disabled)
    tr = InternLMXC2DPOTrainer(model, None, 0.1, 0, "sigmoid", SimpleNamespace(gradient_accumulation_steps=1), None, -100, cfg["model_pad_token_id"],
                               peft_config=dict(r=64, lora_alpha=64, lora_dropout=0.05, target_modules="auto", bias="none", seed=1))
    eng = model.engine
    assert tr.ref_model is None and tr.is_peft_model and eng.lora_fused and eng.resid_f32
    batch = tr._prepare_inputs(synthetic_batch(4, 1024, cfg["image_token"], 32000, cfg["image_size"], seed=9, ragged=True))
    base_before = eng.policy.flat.clone()
    eng.init_optimizer()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    assert int(model._last_ctx["extra"]["R"]) == 8 * 1225 and model._last_ctx["S"] > 2100      # (ragged responses: S = 1225 + the longest text - 1)
    assert abs(float(loss) - math.log(2.0)) < 1e-6, float(loss)
    g = eng.lora_grads.float()
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0, "adapter gradient"
    # lora_B = 0: d lora_A = 0 exactly (it is multiplied by B), d lora_B carries the whole gradient
    ga = {k: float(v.float().abs().max()) for k, v in eng.lgv.items()}
    assert any(v > 0 for k, v in ga.items() if ".b_" in k) and all(v == 0 for k, v in ga.items() if ".a_" in k), "peft init: only lora_B receives a gradient"
    eng.optimizer_step(1e-5, 0.9, 0.95, 1e-6, 0.1, 1.0)
    norm = eng.grad_norm()
    assert math.isfinite(norm) and norm > 1e-6, norm
    assert torch.equal(eng.policy.flat, base_before), "base + PLoRA weights must not move under LoRA"
    model.eval()                                           # eval: no dropout of either adapter
    with torch.no_grad(), tr.null_ref_context():
        c1, r1, _, _ = tr.concatenated_forward(model, batch)
        c2, r2, _, _ = tr.concatenated_forward(model, _permuted(batch, [1, 0, 3, 2]))
    torch.cuda.synchronize()
    assert float((c1[[1, 0, 3, 2]] - c2).abs().max()) < 2e-3 * float(c1.abs().max())
    assert float((r1[[1, 0, 3, 2]] - r2).abs().max()) < 2e-3 * float(r1.abs().max())
    del model, tr
    torch.cuda.empty_cache()
