"""Dataset builders -> rows {prompt, chosen, rejected, img_path} (mirror of /root/reference/src/vlrlhf/utils/data.py:
DATASET_MAP :142-147).  The HF-hub datasets (vlfeedback_paired, rlhfv) need network access; `plain_dpo` reads the same
JSON format as the reference (:120-139) and `synthetic` generates VLFeedback-shaped rows for benchmarks."""
import json
import os

import torch


def make_plain_dpo_dataset(script_args):
    with open(script_args.data_path) as f:
        data = json.load(f)
    rows = []
    for d in data:
        img = d["image"] if "image" in d else d["img_path"]
        if getattr(script_args, "image_root", None):
            img = os.path.join(script_args.image_root, img)
        rows.append(dict(prompt=d["prompt"], chosen=d["chosen"], rejected=d["rejected"], img_path=img))
    return rows


def make_synthetic_dataset(script_args):
    n = int(getattr(script_args, "synthetic_rows", 64))
    size = int(getattr(script_args, "synthetic_image_size", 336))
    g = torch.Generator().manual_seed(1234)
    words = ["alpha", "beta", "gamma", "delta", "red", "blue", "cat", "dog", "tree", "car", "sky", "left", "right"]

    def sent(k):
        return " ".join(words[int(i)] for i in torch.randint(0, len(words), (k,), generator=g))
    rows = []
    for _ in range(n):
        rows.append(dict(prompt="What is shown? " + sent(6), chosen=sent(int(torch.randint(4, 24, (1,), generator=g))),
                         rejected=sent(int(torch.randint(4, 24, (1,), generator=g))),
                         img_path=torch.randn(3, size, size, generator=g)))
    return rows


def _needs_hub(name):
    def f(script_args):
        raise RuntimeError(f"dataset '{name}' is downloaded from the HuggingFace hub by the reference; there is no "
                           "network here - export it to the plain_dpo JSON format (prompt/chosen/rejected/image).")
    return f


DATASET_MAP = {
    "vlfeedback_paired": _needs_hub("vlfeedback_paired"),
    "rlhfv": _needs_hub("rlhfv"),
    "vlquery_json": _needs_hub("vlquery_json"),
    "plain_dpo": make_plain_dpo_dataset,
    "synthetic": make_synthetic_dataset,
}
