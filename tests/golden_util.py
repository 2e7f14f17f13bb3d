"""Helpers to read tests/golden/*.npz (written by oracle/make_golden.py from the reference's own functions)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    cfg = json.loads(bytes(z["config_json"]).decode())
    W = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    W.update({k[4:]: torch.from_numpy(z[k]).view(torch.bfloat16).float() for k in z.files if k.startswith("w16.")})   # bf16 bit patterns
    W_ref = dict(W)
    for k in z.files:
        if k.startswith("ref_w."):
            W_ref[k[6:]] = torch.from_numpy(z[k])
        elif k.startswith("ref_w16."):
            W_ref[k[8:]] = torch.from_numpy(z[k]).view(torch.bfloat16).float()
    batch = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch.") and k != "batch.pixel_values"}
    batch["img_input_dict"] = dict(pixel_values=torch.from_numpy(z["batch.pixel_values"]))
    if "batch.image_sizes" in z.files:
        batch["img_input_dict"]["image_sizes"] = torch.from_numpy(z["batch.image_sizes"])
    batch["img_path"] = ["synthetic"] * batch["chosen_input_ids"].shape[0]
    rows = json.loads(bytes(z["rows_json"]).decode())
    return z, cfg, W, W_ref, batch, rows


def t(z, k):
    return torch.from_numpy(z[k])


# per-pair loss / reward tolerances = 1.5 x the error an MI355X run of the committed binary measured against the reference-generated
# fixtures (profiles/r04_parity_margins.txt, written by `VLR_MARGINS=<file> pytest -m gpu`; VERDICT r03 item 6c: the tests used to
# accept 2.5e-2 on every pair).  The kernels are bit-reproducible, so a run only moves when a kernel's rounding order does: re-measure then.
MEASURED_TOL = {
    "llava.losses.sigmoid": 0.0065,
    "llava.chosen_rewards.sigmoid": 0.0056,
    "llava.losses.hinge": 0.014,
    "llava.chosen_rewards.hinge": 0.0056,
    "llava.losses.ipo": 1.2,
    "llava.chosen_rewards.ipo": 0.0056,
    "llava.losses.kto_pair": 0.002,
    "llava.chosen_rewards.kto_pair": 0.0056,
    "llava.losses.ddpo": 0.0037,
    "llava.chosen_rewards.ddpo": 0.0051,
    "qwenvl.losses.sigmoid": 0.0046,
    "qwenvl.losses.ipo.sqrt": 0.12,
    "qwenvl.losses.ddpo": 0.0046,
    "internlm.losses.sigmoid": 0.0021,
    "llavanext.losses.sigmoid": 0.00092,
    "llavanext.chosen_rewards.sigmoid": 0.0021,
    "llavanext.losses.ddpo": 0.0012,
    "llavanext.chosen_rewards.ddpo": 0.0043,
    "llavanext.losses.ipo": 0.19,
    "llavanext.chosen_rewards.ipo": 0.0021,
    # round 5: the adapter step tests - |loss - oracle|, reward margin, 1 - worst gradient cosine, mean policy log-prob
    # error - at max(1.5 x measured, a floor of 5e-4 / 1e-3 / 5e-4 / 1e-2: a bound below the next rounding change of a kernel would be noise);
    # profiles/r05_parity_margins.txt
    "llava.lora.loss.p0.0": 0.0021,                                          # measured 1.43e-03
    "llava.lora.margin.p0.0": 0.0042,                                        # measured 2.78e-03
    "llava.lora.one_minus_worst_cosine.p0.0": 0.0005,                        # measured 1.68e-04
    "llava.lora.loss.p0.25": 0.0005,                                         # measured 1.04e-04
    "llava.lora.margin.p0.25": 0.001,                                       # measured 3.13e-04
    "llava.lora.one_minus_worst_cosine.p0.25": 0.0005,                       # measured 1.93e-04
    "llavanext.lora.loss": 0.0019,                                           # measured 1.24e-03
    "llavanext.lora.one_minus_worst_cosine": 0.0005,                         # measured 2.19e-04
    "qwenvl.lora.loss.p0.0": 0.0015,                                         # measured 9.74e-04
    "qwenvl.lora.one_minus_worst_cosine.p0.0": 0.0018,                       # measured 1.23e-03
    "qwenvl.lora.loss.p0.25": 0.0005,                                        # measured 6.08e-05
    "qwenvl.lora.one_minus_worst_cosine.p0.25": 0.002,                      # measured 1.34e-03
    "internlm.plora_dropout.loss": 0.0017,                                   # measured 1.12e-03
    "internlm.plora_dropout.mean_logp": 0.045,                              # measured 3.02e-02
    "internlm.lora2.p0.0.pp0.0.mean_logp": 0.057,                           # measured 3.82e-02
    "internlm.lora2.p0.0.pp0.0.loss": 0.0048,                                # measured 3.17e-03
    "internlm.lora2.p0.0.pp0.0.one_minus_worst_cosine": 0.0005,              # measured 2.20e-04
    "internlm.lora2.p0.25.pp0.0.mean_logp": 0.022,                          # measured 1.44e-02
    "internlm.lora2.p0.25.pp0.0.loss": 0.0021,                               # measured 1.39e-03
    "internlm.lora2.p0.25.pp0.0.one_minus_worst_cosine": 0.0005,             # measured 1.77e-04
    "internlm.lora2.p0.25.pp0.5.mean_logp": 0.018,                          # measured 1.22e-02
    "internlm.lora2.p0.25.pp0.5.loss": 0.0039,                               # measured 2.61e-03
    "internlm.lora2.p0.25.pp0.5.one_minus_worst_cosine": 0.0005,             # measured 1.62e-04
    "internlm.lora.composed.p0.25.pp0.0.mean_logp": 0.025,                  # measured 1.69e-02
    "internlm.lora.composed.p0.25.pp0.0.loss": 0.0038,                       # measured 2.55e-03
    "internlm.lora.composed.p0.25.pp0.0.one_minus_worst_cosine": 0.0005,     # measured 2.77e-04
}


def within(name, measured, tol=None, default=None):
    """assert measured < tol; with VLR_MARGINS=<file> the (name, measured, tol) triple is appended to that file first - the tolerances of
    the per-pair loss tests are 1.5 x what an MI355X run of the committed binary measured (profiles/r04_parity_margins.txt,
    profiles/r05_parity_margins.txt); `default`: the bound of a name that has no measured entry yet"""
    measured = float(measured)
    tol = MEASURED_TOL.get(name, default) if tol is None else tol
    assert tol is not None, f"no tolerance for {name}"
    f = os.environ.get("VLR_MARGINS")
    if f:
        with open(f, "a") as fh:
            fh.write(f"{name:60s} measured {measured:.3e}  tolerance {tol:.3e}  ratio {measured / tol:.2f}\n")
    assert measured < tol, (name, measured, tol)


TINY_PROCESSOR = os.path.join(GOLDEN, "tiny_llava_processor")
# a checkpoint-shaped tiny LLaVA whose vocabulary matches tests/golden/tiny_llava_processor (385 tokens -> 392)
TINY_CKPT_CFG = dict(vit_hidden=128, vit_mlp=256, vit_layers=3, vit_heads=2, image_size=28, patch_size=14, hidden=256, inter=512,
                     layers=2, heads=2, vocab=392, image_token=383, model_pad_token_id=384)


def write_tiny_checkpoint(path, cfg=None, seed=5, std=0.05):
    """transformers==4.41.0-layout LLaVA checkpoint directory (config.json + model.safetensors + the tiny processor files)
    with random weights: what `--model_name_or_path` points at.  Returns (cfg, state_dict)."""
    import shutil
    import sys
    from safetensors.torch import save_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import llava_dpo_oracle as O
    cfg = dict(cfg or TINY_CKPT_CFG)
    os.makedirs(path, exist_ok=True)
    W = {k: v.to(torch.bfloat16) for k, v in O.random_weights(cfg, seed=seed, std=std).items()}
    save_file({k: v.contiguous() for k, v in W.items()}, os.path.join(path, "model.safetensors"))
    hf = dict(architectures=["LlavaForConditionalGeneration"], model_type="llava", image_token_index=cfg["image_token"],
              pad_token_id=cfg["model_pad_token_id"], ignore_index=-100, vocab_size=cfg["vocab"],
              text_config=dict(model_type="llama", hidden_size=cfg["hidden"], intermediate_size=cfg["inter"],
                               num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], vocab_size=cfg["vocab"],
                               rms_norm_eps=1e-5, rope_theta=10000.0),
              vision_config=dict(model_type="clip_vision_model", hidden_size=cfg["vit_hidden"], intermediate_size=cfg["vit_mlp"],
                                 num_hidden_layers=cfg["vit_layers"], num_attention_heads=cfg["vit_heads"],
                                 image_size=cfg["image_size"], patch_size=cfg["patch_size"], layer_norm_eps=1e-5))
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf, f)
    for fn in os.listdir(TINY_PROCESSOR):
        shutil.copy(os.path.join(TINY_PROCESSOR, fn), os.path.join(path, fn))
    return cfg, W
