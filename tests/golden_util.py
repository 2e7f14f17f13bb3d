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
    W_ref = dict(W)
    for k in z.files:
        if k.startswith("ref_w."):
            W_ref[k[6:]] = torch.from_numpy(z[k])
    batch = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch.") and k != "batch.pixel_values"}
    batch["img_input_dict"] = dict(pixel_values=torch.from_numpy(z["batch.pixel_values"]))
    batch["img_path"] = ["synthetic"] * batch["chosen_input_ids"].shape[0]
    rows = json.loads(bytes(z["rows_json"]).decode())
    return z, cfg, W, W_ref, batch, rows


def t(z, k):
    return torch.from_numpy(z[k])
