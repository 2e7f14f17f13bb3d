#!/usr/bin/env python3
"""run-to-run determinism probe: one 7B DPO step, prints loss, grad norm and checksums of the gradient regions"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vl-rlhf_amd"))
from types import SimpleNamespace
from vlrlhf.models.Llava import LlavaDPOTrainer, LlavaForRL
from vlrlhf.utils.synthetic import LLAVA_1_5_7B, init_random_model, synthetic_batch
cfg = dict(LLAVA_1_5_7B)
cfg["layers"] = int(os.environ.get("LAYERS", "4"))
model = LlavaForRL(cfg)
ref = init_random_model(model, seed=0, std=0.02, policy_delta=1e-3)
eng = model.engine
eng.init_optimizer()
tr = LlavaDPOTrainer(model, ref, 0.1, 0, "sigmoid", SimpleNamespace(gradient_accumulation_steps=1), None, -100, 0)
tr.ref_on_side_stream = os.environ.get("SIDE", "1") == "1"
batch = tr._prepare_inputs(synthetic_batch(4, 1024, cfg["image_token"], 32000, cfg["image_size"], seed=1234))
for rep in range(3):
    eng.zero_grad()
    loss = tr.training_step(model, batch)
    torch.cuda.synchronize()
    g = eng.grads.float()
    parts = {k: float(eng.gv[k].float().double().sum()) for k in ("lm_head", f"l{cfg['layers']-1}.wdown", "l0.wqkv", "l0.wgu", "proj.w1", "embed", "norm")}
    print(f"rep {rep} loss {float(loss):.9f} |g| {float(g.norm()):.9f} " + " ".join(f"{k}={v:.6e}" for k, v in parts.items()), flush=True)
