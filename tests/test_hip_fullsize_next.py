"""Size-independent properties of the DPO step on the LLaVA-Next-Mistral-7B model of BASELINE.json configs[3] at FULL size (its own
module: the 7B LLaVA-1.5 fixture of test_hip_fullsize.py must be released first - two 7B models with optimizer state do not fit).
Needs a real MI355X with ~250 GB free:  pytest -m gpu"""
import math
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_llava_next_mistral_7b_full_size_properties():
    """BASELINE.json configs[3] at its FULL size: Mistral-7B decoder (32 query / 8 K/V heads, I = 14336), 672x672 image -> 5 tiles,
    2928 image features, per-device batch 4 pairs, max_length 2048 -> S = 4975 (M = 39 800 token rows), full fine-tune, DDPO - under
    --gradient_checkpointing (reference scripts/dpo_llavanext.sh:42; without it the 32 layers' activations alone are 177 GB next to
    130 GB of weights / gradients / AdamW state): identical reference => ln 2 and zero rewards, finite non-zero gradient, and a
    permutation of the pairs permutes the log-probs."""
    if not torch.cuda.is_available() or torch.cuda.mem_get_info()[1] < 250 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    from vlrlhf.models.LlavaNext import LLAVA_NEXT_MISTRAL_7B, LlavaNextDPOTrainer, LlavaNextForRL
    from vlrlhf.utils.synthetic import init_random_model, synthetic_batch_anyres
    cfg = dict(LLAVA_NEXT_MISTRAL_7B)
    model = LlavaNextForRL(cfg)
    ref = init_random_model(model, seed=0, std=0.02, policy_delta=0.0)
    tr = LlavaNextDPOTrainer(model, ref, 0.1, 0, "ddpo", SimpleNamespace(gradient_accumulation_steps=1), None, -100, 0)
    batch = tr._prepare_inputs(synthetic_batch_anyres(4, 2048, cfg["image_token"], 32000, cfg["image_size"], seed=9, ragged=True))
    model.gradient_checkpointing_enable()
    model.engine.init_optimizer()
    model.engine.zero_grad()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    c = model._last_ctx
    longest = int(max(batch["chosen_attention_mask"].sum(-1).max(), batch["rejected_attention_mask"].sum(-1).max()))
    assert c["S"] == longest - 1 + 2928 and c["pack"]["feature_lens"].tolist() == [2928] * 4
    assert c["S"] > 4700 and c["ckpt"] and c["Bn"] == 8
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"[fullsize llava-next] S = {c['S']}, M = {c['M']}, peak allocated {peak:.1f} GiB")
    assert abs(float(loss) - math.log(2.0)) < 1e-6, float(loss)
    model.engine.optimizer_step(1e-6, 0.9, 0.98, 1e-6, 0.0, 1.0)
    norm = model.engine.grad_norm()
    assert math.isfinite(norm) and norm > 1e-4, norm
    g_nl = model.engine.gv["image_newline"].float()
    assert torch.isfinite(g_nl).all() and float(g_nl.abs().max()) > 0            # 96 newline slots per sequence feed this one row
    perm = [1, 0, 3, 2]
    b2 = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            b2[k] = v[perm]
        elif isinstance(v, dict):
            b2[k] = {kk: vv[perm] for kk, vv in v.items()}
        elif isinstance(v, list):
            b2[k] = [v[i] for i in perm]
    model.eval()
    with torch.no_grad():
        c1, r1, _, _ = tr.concatenated_forward(ref, batch)
        c2, r2, _, _ = tr.concatenated_forward(ref, b2)
    torch.cuda.synchronize()
    assert float((c1[perm] - c2).abs().max()) < 2e-3 * float(c1.abs().max())
    assert float((r1[perm] - r2).abs().max()) < 2e-3 * float(r1.abs().max())
    del model, ref, tr
    torch.cuda.empty_cache()
