"""Dataset builders -> rows {prompt, chosen, rejected, img_path} (mirror of /root/reference/src/vlrlhf/utils/data.py:
DATASET_MAP :142-147).  The reference downloads vlfeedback_paired / rlhfv from the HuggingFace hub; there is no network here, so
those two read a local export of the same rows (`--data_path`) and run the reference's pair mining on it; `plain_dpo` and
`vlquery_json` read the same JSON formats as the reference and `synthetic` generates VLFeedback-shaped rows for benchmarks."""
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


def _load_rows(path):
    """a local export of a hub dataset: .json (list), .jsonl, or a directory of such files"""
    files = [os.path.join(path, f) for f in sorted(os.listdir(path))] if os.path.isdir(path) else [path]
    rows = []
    for fn in files:
        if fn.endswith(".jsonl"):
            with open(fn) as f:
                rows += [json.loads(l) for l in f if l.strip()]
        elif fn.endswith(".json"):
            with open(fn) as f:
                d = json.load(f)
            rows += d if isinstance(d, list) else [d]
    return rows


def vlfeedback_pairs(samples, score_margin=-1):
    """reference utils/data.py:16-71 (`make_batch_pairs`): every pair of completions of a sample is ordered by the MEAN of the
    annotators' per-aspect `Rating`s (pairs with an unparsable rating or a tie are skipped); score_margin == -1 keeps only the
    pairs with the LARGEST gap of the sample, otherwise every pair whose gap >= score_margin.  -> rows {prompt, chosen, rejected, img_path}"""
    from itertools import combinations
    out = []
    for s in samples:
        comps = s["completions"]
        by_gap = {}
        for i, j in combinations(range(len(comps["annotations"])), 2):
            a1, a2 = comps["annotations"][i], comps["annotations"][j]
            try:
                s1 = sum(float(a1[k]["Rating"]) for k in a1) / len(a1)
                s2 = sum(float(a2[k]["Rating"]) for k in a2) / len(a2)
            except ValueError:
                continue
            if s1 == s2:
                continue
            c, r = (i, j) if s1 > s2 else (j, i)
            by_gap.setdefault(abs(s1 - s2), []).append(dict(prompt=s["prompt"], chosen=comps["response"][c], rejected=comps["response"][r],
                                                               img_path=s["img_path"]))
        if not by_gap:
            continue
        if score_margin == -1:
            out += by_gap[max(by_gap)]
        else:
            for gap, rows in by_gap.items():       # insertion order = the order the reference's defaultdict saw the gaps
                if gap >= score_margin:
                    out += rows
    return out


def make_vlfeedback_paired_dataset(script_args):
    """reference :11-82.  The hub dataset MMInstruction/VLFeedback cannot be downloaded here: `--data_path` names a local export
    of its rows ({prompt, img_path, completions: {annotations: [...], response: [...]}}; .json / .jsonl / directory)."""
    path = getattr(script_args, "data_path", None)
    if not path:
        raise RuntimeError("dataset 'vlfeedback_paired' is downloaded from the HuggingFace hub by the reference; there is no network "
                           "here - pass --data_path <local export of MMInstruction/VLFeedback rows>")
    rows = _load_rows(path)
    root = getattr(script_args, "image_root", None)
    if root:
        for r in rows:
            r["img_path"] = os.path.join(root, r["img_path"])
    return vlfeedback_pairs(rows, getattr(script_args, "score_margin", -1))


def make_rlhfv_paired_dataset(script_args):
    """reference :100-117: rows {image_path, text: json{question, chosen, rejected}} of a local export of HaoyeZhang/RLHF-V-Dataset"""
    path = getattr(script_args, "data_path", None)
    if not path:
        raise RuntimeError("dataset 'rlhfv' is downloaded from the HuggingFace hub by the reference; pass --data_path <local export>")
    out = []
    for r in _load_rows(path):
        t = json.loads(r["text"]) if isinstance(r["text"], str) else r["text"]
        out.append(dict(img_path=os.path.join(script_args.image_root or "", r["image_path"]), prompt=t["question"], chosen=t["chosen"],
                        rejected=t["rejected"]))
    return out


def build_dataset_from_vlquery_json(script_args):
    """reference :85-97: a JSON list whose rows carry `image`; img_path = image_root / image, everything else passes through"""
    with open(script_args.data_path) as f:
        raw = json.load(f)
    return [dict(d, img_path=os.path.join(script_args.image_root or "", d["image"])) for d in raw]


DATASET_MAP = {
    "vlfeedback_paired": make_vlfeedback_paired_dataset,
    "rlhfv": make_rlhfv_paired_dataset,
    "vlquery_json": build_dataset_from_vlquery_json,
    "plain_dpo": make_plain_dpo_dataset,
    "synthetic": make_synthetic_dataset,
}
