#!/usr/bin/env python3
"""Golden vectors for the Qwen-VL DPO path.  TEST INFRASTRUCTURE - build container only.

    python oracle/make_golden_qwenvl.py      ->  tests/golden/qwenvl_small.npz, tests/golden/qwenvl_tokenize.json

The REFERENCE's own model code runs here: `QWenLMHeadModel` (/root/reference/src/vlrlhf/models/QwenVL/modeling_qwen.py, with
its `VisionTransformer` / `Resampler`, visual.py) is imported under the stubs of oracle/make_golden.py plus a stub for
torchvision (absent from the container; only `image_transform`, which this script bypasses, uses it), instantiated at
kernel-compatible toy widths (decoder head_dim 128; ViT width 208 / 2 heads = head_dim 104 like the real 1664 / 16; 8x8 patches
-> the 16x16 position table is bicubically shrunk, 4x4 = 16 queries -> the resampler's key positions are bicubically grown) and
driven through its real `forward`: the image paths are decoded from the token ids, `visual.encode` (which would open the files) is
replaced by `visual(pixels of that path)`.  Log-probs and losses come from the reference's `VLDPOTrainer.get_batch_logps` /
`dpo_loss`; gradients from autograd through the reference model.  Weights are rounded to bf16 BEFORE the run and stored as bf16
bit patterns, so the HIP path loads exactly the values the reference computed with.

The second file pins the tokenisation side: the reference's `QwenVLProcessor.process_batch_conv` / `format_multimodal_prompt` and
`QwenVLDPOTrainer.tokenize_row` (QwenVL/__init__.py:96-218, 257-347) driven with a deterministic stand-in tokenizer (the real one
needs tiktoken + the Qwen vocabulary file, neither is in the container): input rows + the exact output lists.
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _T:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


_tv = _stub("torchvision")
_tv.transforms = _stub("torchvision.transforms", Compose=_T, Resize=_T, ToTensor=_T, Normalize=_T,
                       InterpolationMode=types.SimpleNamespace(BICUBIC=3))

from vlrlhf.models.QwenVL.modeling_qwen import QWenLMHeadModel  # noqa: E402
from vlrlhf.models.QwenVL.configuration_qwen import QWenConfig  # noqa: E402
import vlrlhf.models.QwenVL as RQ  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

CFG = dict(
    family="qwen_vl", hidden=256, inter=256, layers=2, heads=2, vocab=520, rms_eps=1e-6, rope_theta=10000.0,
    image_start_id=500, pad_token_id=510, im_start_id=511, im_end_id=512,
    visual=dict(width=208, heads=2, layers=2, mlp_ratio=2.0, patch_size=14, image_size=112, output_dim=256, n_queries=16),
    pairs=2, beta=0.1, w_scale=3.0, perturb=0.4,
)
GRAD_KEYS = ["transformer.h.0.attn.c_attn.weight", "transformer.h.0.attn.c_attn.bias", "transformer.h.1.attn.c_proj.weight",
             "transformer.h.0.mlp.w1.weight", "transformer.h.1.mlp.w2.weight", "transformer.h.1.mlp.c_proj.weight",
             "transformer.h.0.ln_1.weight", "transformer.ln_f.weight", "lm_head.weight", "transformer.wte.weight"] + \
            ["transformer.visual.attn_pool." + k for k in ("query", "kv_proj.weight", "attn.in_proj_weight", "attn.in_proj_bias",
                                                           "attn.out_proj.weight", "attn.out_proj.bias", "ln_q.weight", "ln_q.bias",
                                                           "ln_kv.weight", "ln_kv.bias")]


def build(cfg, seed):
    torch.manual_seed(seed)
    v = cfg["visual"]
    qc = QWenConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                    kv_channels=cfg["hidden"] // cfg["heads"], intermediate_size=2 * cfg["inter"], seq_length=2048, bf16=False, fp16=False,
                    fp32=True, use_flash_attn=False, no_bias=True, rotary_emb_base=cfg["rope_theta"], use_dynamic_ntk=True,
                    use_logn_attn=True, layer_norm_epsilon=cfg["rms_eps"], emb_dropout_prob=0.0, attn_dropout_prob=0.0,
                    visual=dict(heads=v["heads"], image_size=v["image_size"], image_start_id=cfg["image_start_id"], layers=v["layers"],
                                mlp_ratio=v["mlp_ratio"], output_dim=v["output_dim"], patch_size=v["patch_size"], width=v["width"],
                                n_queries=v["n_queries"]), tie_word_embeddings=False)
    m = QWenLMHeadModel(qc).float()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2 and "pos_embed" not in n:
                p.mul_(cfg["w_scale"])
            elif n.endswith(".bias"):
                p.copy_(torch.randn_like(p) * 0.05)
            elif "ln" in n and n.endswith(".weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
        m.transformer.visual.attn_pool.query.copy_(torch.randn_like(m.transformer.visual.attn_pool.query) * 0.5)
        for p in m.parameters():                      # the values the HIP path will hold
            p.copy_(p.to(torch.bfloat16).float())
    return m


def encode_image_slot(path: str, cfg):
    """what tokenization_qwen.py:283-294 puts between the markers: the utf-8 bytes of the path, then <imgpad> up to n_queries ids"""
    nq = cfg["visual"]["n_queries"]
    b = list(path.encode("utf-8"))
    assert len(b) < nq
    return [cfg["image_start_id"]] + b + [cfg["image_start_id"] + 2] * (nq - len(b)) + [cfg["image_start_id"] + 1]


def make_batch(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    n = cfg["pairs"]
    s = cfg["visual"]["image_size"]
    paths = [f"im{i}.png" for i in range(n)]
    pixels = {p: torch.randn(3, s, s, generator=g) for p in paths}
    rnd = lambda k: torch.randint(256, cfg["image_start_id"] - 1, (k,), generator=g).tolist()   # noqa: E731  text ids
    rows = []
    for i in range(n):
        prompt = rnd(3) + encode_image_slot(paths[i], cfg) + rnd(4 + 3 * i)
        rows.append(dict(prompt=prompt, chosen=rnd(6 + 5 * i) + [cfg["im_end_id"]], rejected=rnd(11 - 3 * i) + [cfg["im_end_id"]]))
    batch = {}
    for side in ("chosen", "rejected"):
        seqs = [r["prompt"] + r[side] for r in rows]
        T = max(len(x) for x in seqs)
        ids = torch.full((n, T), cfg["pad_token_id"], dtype=torch.long)
        am = torch.zeros(n, T, dtype=torch.long)
        lab = torch.full((n, T), -100, dtype=torch.long)
        for i, x in enumerate(seqs):
            ids[i, :len(x)] = torch.tensor(x)
            am[i, :len(x)] = 1
            lab[i, len(rows[i]["prompt"]):len(x)] = torch.tensor(x[len(rows[i]["prompt"]):])
        batch[f"{side}_input_ids"], batch[f"{side}_attention_mask"], batch[f"{side}_labels"] = ids, am, lab
    return batch, paths, pixels


def ref_forward(model, ids, am, pixels):
    """the reference forward with visual.encode reading `pixels[path]` instead of the file system"""
    vis = model.transformer.visual
    seen = []

    def encode(image_paths):
        seen.extend(image_paths)
        return vis(torch.stack([pixels[p] for p in image_paths], 0))
    vis.encode = encode
    # version drift: the reference pins transformers 4.41.0, whose ModuleUtilsMixin.get_head_mask(None, n) returns [None] * n; the
    # container's 5.x removed the method
    model.transformer.get_head_mask = lambda head_mask, n, *a, **k: [None] * n
    out = model(input_ids=ids, attention_mask=am, return_dict=True, use_cache=False)
    return out, seen


def concat(batch, pad):
    n = max(batch["chosen_input_ids"].shape[1], batch["rejected_input_ids"].shape[1])
    out = {}
    for f, p in (("input_ids", pad), ("attention_mask", 0), ("labels", -100)):
        parts = []
        for side in ("chosen", "rejected"):
            t = batch[f"{side}_{f}"]
            parts.append(torch.cat([t, torch.full((t.shape[0], n - t.shape[1]), p, dtype=t.dtype)], 1) if t.shape[1] < n else t)
        out[f] = torch.cat(parts, 0)
    return out


def gen_model():
    cfg = dict(CFG)
    policy = build(cfg, 0).train()        # training mode as in the DPO step (dropout probabilities are 0; logn scaling is inference-only)
    ref = build(cfg, 0).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(7)
        for (n, p), q in zip(policy.named_parameters(), ref.parameters()):
            if "visual" not in n or ("attn_pool" in n and "pos_embed" not in n):
                # the ViT trunk is shared and frozen; the LLM and the resampler (re-enabled by freeze_vision_tower, :33-37) are trained
                p.copy_((q + cfg["perturb"] * q.abs().mean() * torch.randn(q.shape, generator=g)).to(torch.bfloat16).float())
    batch, paths, pixels = make_batch(cfg, 3)
    cb = concat(batch, cfg["pad_token_id"])
    out, seen = ref_forward(policy, cb["input_ids"], cb["attention_mask"], pixels)
    assert seen == paths + paths, seen
    with torch.no_grad():
        rout, _ = ref_forward(ref, cb["input_ids"], cb["attention_mask"], pixels)
        vis_feat = ref.transformer.visual(torch.stack([pixels[p] for p in paths], 0))
        pol_feat = policy.transformer.visual(torch.stack([pixels[p] for p in paths], 0))
    tr = G.VLDPOTrainer.__new__(G.VLDPOTrainer)
    tr.accelerator = types.SimpleNamespace(device=torch.device("cpu"))
    res = {}
    nc = cfg["pairs"]
    for lt in ("sigmoid", "ipo", "ddpo"):
        tr.loss_type, tr.beta, tr.label_smoothing, tr.reference_free = lt, cfg["beta"], 0.0, False
        tr.label_pad_token_id, tr.is_encoder_decoder = -100, False
        kw = dict(average_log_prob=False, label_pad_token_id=-100, is_encoder_decoder=False)      # as concatenated_forward calls it (base/trainer.py:230)
        if lt == "ddpo":
            kw["mask_shared_tokens"] = True
        lp = G.VLDPOTrainer.get_batch_logps(out.logits, cb["labels"], **kw)
        rlp = G.VLDPOTrainer.get_batch_logps(rout.logits, cb["labels"], **kw)
        losses, cr, rr = tr.dpo_loss(lp[:nc], lp[nc:], rlp[:nc], rlp[nc:])
        res[lt] = dict(logps=lp.detach(), ref_logps=rlp.detach(), losses=losses.detach(), loss=losses.mean())
    policy.zero_grad()
    res["sigmoid"]["loss"].backward()
    grads = {k: p.grad.detach().clone() for k, p in policy.named_parameters() if k in GRAD_KEYS}
    assert len(grads) == len(GRAD_KEYS)
    z = {"config_json": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)}
    bits = lambda t: t.detach().to(torch.bfloat16).view(torch.int16).numpy()   # noqa: E731
    rsd = ref.state_dict()
    for k, v in policy.state_dict().items():
        if "logn" in k or "rotary" in k or "masked_bias" in k:
            continue
        z["w16." + k] = bits(v)
        if not torch.equal(v, rsd[k]):
            z["ref_w16." + k] = bits(rsd[k])
    for k, v in batch.items():
        z["batch." + k] = v.numpy()
    z["batch.pixel_values"] = torch.stack([pixels[p] for p in paths], 0).numpy()
    z["paths_json"] = np.frombuffer(json.dumps(paths).encode(), dtype=np.uint8)
    z["rows_json"] = np.frombuffer(json.dumps([]).encode(), dtype=np.uint8)
    z["visual_features"] = vis_feat.numpy()
    z["policy_visual_features"] = pol_feat.numpy()
    z["image_position_map"] = out.image_position_map.numpy()
    z["logits"] = out.logits.detach().numpy().astype(np.float32)
    for lt, d in res.items():
        for k, v in d.items():
            z[f"{lt}.{k}"] = v.detach().numpy()
    for k, v in grads.items():
        if v.numel() > 40000:            # probe of the big ones: every 17th element + the norm
            z["grad_probe." + k] = v.reshape(-1)[::17].numpy()
            z["grad_norm." + k] = np.array(float(v.norm()))
        else:
            z["grad." + k] = v.numpy()
    path = os.path.join(OUT, "qwenvl_small.npz")
    np.savez_compressed(path, **z)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB; loss", {k: float(v['loss']) for k, v in res.items()})


# ------------------------------------------------------------------------------------------------- tokenisation fixture
sys.path.insert(0, ROOT)
from tests.qwen_standin import StandInTokenizer  # noqa: E402


def gen_tokenize():
    tok = StandInTokenizer()
    proc = RQ.QwenVLProcessor.__new__(RQ.QwenVLProcessor)
    proc._QwenVLProcessor__tokenizer = tok
    proc.train()
    rows = [dict(prompt="What is in the picture?", chosen="A dog on a sofa.", rejected="Two cats.", img_path="imgs/a.jpg"),
            dict(prompt="<image>" + "Describe the image in detail, please. " * 3, chosen="It shows a street " * 8, rejected="no", img_path="b.png"),
            dict(prompt="hi", chosen="yes" * 40, rejected="maybe " * 30, img_path="c/d/e.jpeg")]
    cases = []
    for max_length, max_prompt_length, mode in ((1024, 512, "keep_end"), (300, 280, "keep_end"), (300, 280, "keep_start"), (290, 270, "keep_end")):
        tr = RQ.QwenVLDPOTrainer.__new__(RQ.QwenVLDPOTrainer)
        tr.processor, tr.tokenizer = proc, tok
        tr.max_length, tr.max_prompt_length, tr.truncation_mode, tr.label_pad_token_id = max_length, max_prompt_length, mode, -100
        outs = [tr.tokenize_row(dict(r)) for r in rows]
        cases.append(dict(max_length=max_length, max_prompt_length=max_prompt_length, truncation_mode=mode, out=outs))
    conv = [[{"from": "user", "value": RQ.QwenVLProcessor.format_multimodal_prompt(r["prompt"], r["img_path"])},
             {"from": "assistant", "value": r["chosen"]}] for r in rows]
    pbc = proc.process_batch_conv(conv)
    fmt = [RQ.QwenVLProcessor.format_multimodal_prompt(r["prompt"], r["img_path"]) for r in rows]
    path = os.path.join(OUT, "qwenvl_tokenize.json")
    with open(path, "w") as f:
        json.dump(dict(rows=rows, cases=cases, process_batch_conv=pbc, format_multimodal_prompt=fmt,
                       is_valid=[RQ.QwenVLProcessor.is_multimodal_prompt_valid(x) for x in fmt],
                       removed=[RQ.QwenVLProcessor.remove_image_placeholder(x) for x in fmt]), f)
    print("wrote", path, f"{os.path.getsize(path) / 1e3:.1f} kB")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "model"):
        gen_model()
    if which in ("all", "tokenize"):
        gen_tokenize()
