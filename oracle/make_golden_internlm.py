#!/usr/bin/env python3
"""Golden vectors for the InternLM-XComposer2 DPO path.  TEST INFRASTRUCTURE - build container only.

    python oracle/make_golden_internlm.py      ->  tests/golden/internlmxc2_small.npz

The REFERENCE's own classes run here: `InternLMXC2ForRL` (/root/reference/src/vlrlhf/models/InternLMXC2/__init__.py: the LLaVA-style
merge :33-105 and forward :107-236) on top of the vendored `InternLM2Model` with its `PLoRA` linears (modeling_internlm2.py,
build_mlp.py:158-203), imported under the stubs of oracle/make_golden.py plus a torchvision stub.  Stand-ins this script has to
provide, all outside the arithmetic under test:
  * `build_vision_tower()` downloads openai/clip-vit-large-patch14-336 from the hub - replaced by a randomly initialised
    `CLIPVisionModel` of toy size (same class, `select_layer = -1`, "patch" features, position table already at the target size);
  * `build_vision_projector()` hard-codes 1024 -> 4096 -> 4096 - replaced by the same `mlp2x_gelu` at toy widths;
  * config attributes the real checkpoint's config.json carries (`max_length`, `img_size`, `image_token_index`, `ignore_index`) and
    `rope_scaling = None` are set by hand (transformers 5.x no longer turns unknown constructor kwargs into attributes).
The model runs in eval mode (PLoRA's nn.Dropout draws from torch's RNG - no other implementation can reproduce its mask; the
dropout path is checked against the oracle's restatement with the product's counter-based mask instead).  Weights are rounded to
bf16 before the run and stored as bf16 bit patterns."""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

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
_tv.transforms.functional = _stub("torchvision.transforms.functional", InterpolationMode=types.SimpleNamespace(BICUBIC=3))

from transformers import CLIPVisionConfig, CLIPVisionModel  # noqa: E402
import vlrlhf.models.InternLMXC2.build_mlp as BM  # noqa: E402

CFG = dict(
    family="internlm_xc2", hidden=256, inter=256, layers=2, heads=2, kv_heads=1, vocab=200, rms_eps=1e-5, rope_theta=1000000.0,
    vit_hidden=128, vit_mlp=256, vit_layers=3, vit_heads=2, image_size=42, patch_size=14, vit_feature_layer=-1,
    image_token=190, model_pad_token_id=2, plora_r=256, plora_alpha=256, plora_dropout=0.05,
    pairs=2, prompt_len=(7, 10), resp_len=(6, 14), beta=0.1, w_scale=3.0, perturb=0.3,
)

_VC = CLIPVisionConfig(hidden_size=CFG["vit_hidden"], intermediate_size=CFG["vit_mlp"], num_hidden_layers=CFG["vit_layers"],
                       num_attention_heads=CFG["vit_heads"], image_size=CFG["image_size"], patch_size=CFG["patch_size"],
                       hidden_act="quick_gelu", layer_norm_eps=1e-5)
_VC._attn_implementation = "eager"


class _Proxy:
    """transformers 5.x flattened CLIPVisionModel (no inner `.vision_model`); the reference (4.41.0) reaches through it
    (InternLMXC2/__init__.py:259).  Attribute reads and writes go to the model itself."""

    def __init__(self, m):
        object.__setattr__(self, "_m", m)

    def __getattr__(self, k):
        return getattr(self._m, k)

    def __setattr__(self, k, v):
        setattr(self._m, k, v)


class _Tower(BM.CLIPVisionTower):
    def load_model(self):
        self.vision_tower = CLIPVisionModel(_VC)
        self.vision_tower.requires_grad_(False)
        object.__setattr__(self.vision_tower, "vision_model", _Proxy(self.vision_tower))
        self.is_loaded = True

    def resize_pos(self):           # the toy tower is created at its final size (the real checkpoint stores the resized table too)
        self.is_resize_pos = True


BM.build_vision_tower = lambda: _Tower("toy")
BM.build_vision_projector = lambda: nn.Sequential(nn.Linear(CFG["vit_hidden"], CFG["hidden"]), nn.GELU(), nn.Linear(CFG["hidden"], CFG["hidden"]))
import vlrlhf.models.InternLMXC2.modeling_internlm_xcomposer2 as MX  # noqa: E402
MX.build_vision_tower, MX.build_vision_projector = BM.build_vision_tower, BM.build_vision_projector
import vlrlhf.models.InternLMXC2 as RI  # noqa: E402
from vlrlhf.models.InternLMXC2.configuration_internlm_xcomposer2 import InternLMXcomposer2Config  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
GRAD_KEYS = ["model.layers.0.attention.wqkv.weight", "model.layers.0.attention.wqkv.Plora_A.weight", "model.layers.1.attention.wqkv.Plora_B.weight",
             "model.layers.1.attention.wo.weight", "model.layers.0.attention.wo.Plora_B.weight", "model.layers.0.feed_forward.w1.weight",
             "model.layers.0.feed_forward.w3.Plora_A.weight", "model.layers.1.feed_forward.w2.weight", "model.layers.0.feed_forward.w2.Plora_B.weight",
             "model.layers.0.attention_norm.weight", "model.norm.weight", "output.weight", "model.tok_embeddings.weight"]


def build(cfg, seed):
    torch.manual_seed(seed)
    c = InternLMXcomposer2Config(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], intermediate_size=cfg["inter"],
                                 num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], num_key_value_heads=cfg["kv_heads"],
                                 rms_norm_eps=cfg["rms_eps"], rope_theta=cfg["rope_theta"], max_position_embeddings=4096, bias=False,
                                 attn_implementation="eager", pad_token_id=cfg["model_pad_token_id"])
    c.rope_scaling, c.max_length, c.img_size = None, 4096, cfg["image_size"]
    c.image_token_index, c.ignore_index = cfg["image_token"], -100
    c._attn_implementation = "eager"
    m = RI.InternLMXC2ForRL(c).float()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.mul_(cfg["w_scale"]) if "Plora" not in n else p.copy_(torch.randn_like(p) * (0.04 if "Plora_A" in n else 0.02))
            elif n.endswith("norm.weight") or "layer_norm" in n or "layrnorm" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith(".bias"):
                p.copy_(torch.randn_like(p) * 0.05)
        # nn.Embedding(padding_idx=pad) starts that row at zero; in the trained checkpoint it is an ordinary row (pad = </s> = 2), and an
        # all-zero text embedding would be mistaken for an image slot by the merge (:80-81)
        m.model.tok_embeddings.weight[cfg["model_pad_token_id"]].copy_(torch.randn(cfg["hidden"]) * 0.06)
        for p in m.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    return m.eval()


def make_batch(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    n, s = cfg["pairs"], cfg["image_size"]
    px = torch.randn(n, 3, s, s, generator=g)
    rnd = lambda k: torch.randint(3, cfg["image_token"] - 1, (k,), generator=g).tolist()   # noqa: E731
    rows = []
    for i in range(n):
        prompt = [1] + rnd(2) + [cfg["image_token"]] + rnd(cfg["prompt_len"][0] + 3 * i)
        # chosen and rejected SHARE spans of >= 3 tokens (a common opening, and for the second pair a common middle), as real preference
        # pairs do: the DDPO mask (trainer.py:161-184, min_match_size 3) then removes them and `ddpo` differs from `sigmoid`
        lc, lr = cfg["resp_len"][0] + 4 * i, cfg["resp_len"][1] - 5 * i
        head, mid = rnd(3), rnd(3) if i else []
        k = len(head) + len(mid)
        chosen = head + rnd((lc - k) // 2) + mid + rnd(lc - k - (lc - k) // 2)
        rejected = head + rnd((lr - k) // 2) + mid + rnd(lr - k - (lr - k) // 2)
        rows.append(dict(prompt=prompt, chosen=chosen + [2], rejected=rejected + [2]))
    batch = {}
    pad = cfg["model_pad_token_id"]
    for side in ("chosen", "rejected"):
        seqs = [r["prompt"] + r[side] for r in rows]
        T = max(len(x) for x in seqs)
        ids = torch.full((n, T), pad, dtype=torch.long)
        am = torch.zeros(n, T, dtype=torch.long)
        lab = torch.full((n, T), -100, dtype=torch.long)
        for i, x in enumerate(seqs):
            ids[i, :len(x)] = torch.tensor(x)
            am[i, :len(x)] = 1
            lab[i, len(rows[i]["prompt"]):len(x)] = torch.tensor(x[len(rows[i]["prompt"]):])
        batch[f"{side}_input_ids"], batch[f"{side}_attention_mask"], batch[f"{side}_labels"] = ids, am, lab
    return batch, px


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


def main():
    cfg = dict(CFG)
    policy, ref = build(cfg, 0), build(cfg, 0)
    with torch.no_grad():
        g = torch.Generator().manual_seed(7)
        for (n, p), q in zip(policy.named_parameters(), ref.parameters()):
            if not n.startswith(("vit.", "vision_proj.")):      # --freeze_vision_tower True freezes the tower AND the projector (:252-255)
                p.copy_((q + cfg["perturb"] * q.abs().mean() * torch.randn(q.shape, generator=g)).to(torch.bfloat16).float())
    for m in (policy, ref):
        m.freeze_vision_tower()
    batch, px = make_batch(cfg, 3)
    cb = concat(batch, cfg["model_pad_token_id"])
    px2 = torch.cat([px, px], 0)
    out = policy(input_ids=cb["input_ids"], attention_mask=cb["attention_mask"], labels=cb["labels"], pixel_values=px2, use_cache=False, return_dict=True)
    with torch.no_grad():
        rout = ref(input_ids=cb["input_ids"], attention_mask=cb["attention_mask"], labels=cb["labels"], pixel_values=px2, use_cache=False, return_dict=True)
        img_feat = ref.encode_img(px)
    tr = G.VLDPOTrainer.__new__(G.VLDPOTrainer)
    tr.accelerator = types.SimpleNamespace(device=torch.device("cpu"))
    res, nc = {}, cfg["pairs"]
    for lt in ("sigmoid", "ddpo"):
        tr.loss_type, tr.beta, tr.label_smoothing, tr.reference_free = lt, cfg["beta"], 0.0, False
        tr.label_pad_token_id, tr.is_encoder_decoder = -100, False
        kw = dict(average_log_prob=False, label_pad_token_id=-100, is_encoder_decoder=False)
        if lt == "ddpo":
            kw["mask_shared_tokens"] = True
        lp = G.VLDPOTrainer.get_batch_logps(out.logits, out.labels, **kw)
        rlp = G.VLDPOTrainer.get_batch_logps(rout.logits, rout.labels, **kw)
        losses, cr, rr = tr.dpo_loss(lp[:nc], lp[nc:], rlp[:nc], rlp[nc:])
        res[lt] = dict(logps=lp.detach(), ref_logps=rlp.detach(), losses=losses.detach(), loss=losses.mean())
    policy.zero_grad()
    res["sigmoid"]["loss"].backward()
    grads = {k: p.grad.detach().clone() for k, p in policy.named_parameters() if k in GRAD_KEYS}
    assert len(grads) == len(GRAD_KEYS), sorted(set(GRAD_KEYS) - set(grads))
    z = {"config_json": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), "rows_json": np.frombuffer(b"[]", dtype=np.uint8)}
    bits = lambda t: t.detach().to(torch.bfloat16).view(torch.int16).numpy()   # noqa: E731
    rsd = ref.state_dict()
    for k, v in policy.state_dict().items():
        if "position_ids" in k or "inv_freq" in k:
            continue
        name = k.replace("vit.vision_tower.", "vit.vision_tower.vision_model.") if k.startswith("vit.vision_tower.") and ".vision_model." not in k else k
        z["w16." + name] = bits(v)                     # checkpoint names of transformers 4.41.0 (inner `vision_model.` level)
        if not torch.equal(v, rsd[k]):
            z["ref_w16." + name] = bits(rsd[k])
    for k, v in batch.items():
        z["batch." + k] = v.numpy()
    z["batch.pixel_values"] = px.numpy()
    z["image_features"] = img_feat.numpy()
    z["image_position_map"] = out.image_position_map.numpy()
    z["merged_labels"] = out.labels.numpy()
    z["logits"] = out.logits.detach().numpy().astype(np.float32)
    for lt, d in res.items():
        for k, v in d.items():
            z[f"{lt}.{k}"] = v.detach().numpy()
    for k, v in grads.items():
        if v.numel() > 40000:
            z["grad_probe." + k] = v.reshape(-1)[::17].numpy()
            z["grad_norm." + k] = np.array(float(v.norm()))
        else:
            z["grad." + k] = v.numpy()
    path = os.path.join(OUT, "internlmxc2_small.npz")
    np.savez_compressed(path, **z)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB; loss", {k: float(v['loss'].detach()) for k, v in res.items()})
    print("logps", res["sigmoid"]["logps"], res["sigmoid"]["ref_logps"])


if __name__ == "__main__" and (len(sys.argv) < 2 or sys.argv[1] == "model"):
    main()


def gen_tokenize():
    """the reference's InternLMXC2Processor.process_batch_conv / format_multimodal_prompt with the stand-in tokenizer of
    tests/qwen_standin.py -> tests/golden/internlm_tokenize.json"""
    sys.path.insert(0, ROOT)
    from tests.qwen_standin import StandInInternLMTokenizer
    proc = RI.InternLMXC2Processor.__new__(RI.InternLMXC2Processor)
    proc._InternLMXC2Processor__tokenizer = StandInInternLMTokenizer()
    rows = [dict(prompt="What is in the picture?", answer="A dog on a sofa.", img_path="imgs/a.jpg"),
            dict(prompt="<image>Describe it. 中文", answer="", img_path="b.png"),
            dict(prompt="hi <image> there", answer="yes " * 7, img_path="c.jpeg")]
    fmt = [RI.InternLMXC2Processor.format_multimodal_prompt(r["prompt"], r["img_path"]) for r in rows]
    conv = [[{"from": "user", "value": f}, {"from": "assistant", "value": r["answer"]}] for f, r in zip(fmt, rows)]
    out = dict(rows=rows, format_multimodal_prompt=fmt, process_batch_conv=proc.process_batch_conv(conv),
               process_batch_conv_end=proc.process_batch_conv(conv, add_end_for_empty_value=True),
               is_valid=[RI.InternLMXC2Processor.is_multimodal_prompt_valid(x) for x in fmt],
               removed=[RI.InternLMXC2Processor.remove_image_placeholder(x) for x in fmt],
               template=vars(proc.chat_template))
    path = os.path.join(OUT, "internlm_tokenize.json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=False)
    print("wrote", path, f"{os.path.getsize(path) / 1e3:.1f} kB")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tokenize":
    gen_tokenize()
