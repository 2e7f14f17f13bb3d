"""CPU oracle for the LLaVA-1.5 DPO training step.  TEST INFRASTRUCTURE - NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
file, and only as the checker.  The product path (vl-rlhf_amd/) never imports it and fails
loudly when the HIP library is missing.

This is a plain-PyTorch fp32 RESTATEMENT (no HuggingFace / trl / reference import) of the
algorithm on the reference's DPO hot path.  Each function cites the reference lines it
follows (paths relative to /root/reference).  Where the arithmetic lives in a third-party
dependency that is not vendored in the reference (transformers==4.41.0, trl==0.8.1,
torch AdamW) the published algorithm is restated and the call site is cited.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which oracle/make_golden.py generated in the build container by
running the reference's own functions (get_batch_logps, dpo_loss, collator, LLaVA merge,
get_diff_ids) composed with the installed HF CLIP / projector / LLaMA modules.

LoRA: `lora=` restates peft's lora.Linear.forward (result = base(x) + lora_B(lora_A(dropout(x))) * scaling; peft is a
pinned dependency of the reference, requirements.txt, NOT installed in the build container, so no golden vectors could be
generated from it).  It is pinned indirectly: tests/test_oracle_golden.py checks that the LoRA forward equals the
golden-pinned base forward on the merged weights W + scaling * B A.  `dropout_mask` restates the counter-based mask of
the HIP path (vl-rlhf_amd/csrc/elementwise.hip) so that both sides can be run with the SAME mask.

`emulate_bf16=True` rounds every tensor the HIP path stores as bf16 (weights, GEMM outputs,
norm/rope/attention/activation outputs, residual stream) to bf16 and back, so the HIP path
can be compared against it with a tight tolerance; fp32 mode is the reference-exact one.
"""
import difflib
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# What the MI355X path rounds to bf16 with the fp32 residual stream (ABI v4, vlr_llama_cfg.resid_f32): weights, the operands of every
# MFMA (norm outputs, q / k after RoPE, v, softmax probabilities, attention output, silu(gate) * up, the final hidden state) and the
# vision tower's operands and residual stream - NOT the decoder's residual stream, NOT the projector output (the image rows of the
# merged embeddings), NOT q / k before RoPE or gate / up (they exist only as fp32 accumulators).  emulate_bf16=HIP_ROUNDING is the
# oracle's model of that path; emulate_bf16=True (everything the r01 / r02 path stored) remains for the bf16-stream mode.
HIP_ROUNDING = frozenset(("w", "vit_op", "vit_resid", "xn", "rope", "v", "p", "attn", "act", "hidden", "lora_x", "lora_u"))


def _rt(x, on, tag=None):
    """bf16 round trip.  `on`: False (fp32, reference-exact), True (every tensor the HIP path stores as bf16), or a set of
    rounding-point tags (error-budget runs, oracle/depth_parity.py): only tensors whose tag is in the set are rounded.
    Tags of the decoder: x0 xn qkv v rope p attn resid gu act hidden; "w" weights; "vit" the whole vision tower +
    projector.  "p" (softmax probabilities, the A operand of the P.V MFMA) is rounded only when named explicitly."""
    if isinstance(on, (set, frozenset)):
        on = tag in on
    return x.to(torch.bfloat16).to(torch.float32) if on else x


# ----------------------------------------------------------------------------------------------------------
# Collator + concatenation (integer work)
# ----------------------------------------------------------------------------------------------------------
def collate(features: List[dict], pad_token_id: int = 0, label_pad_token_id: int = IGNORE_INDEX) -> dict:
    """src/vlrlhf/base/collator.py:26-68 (decoder-only branch): chosen_/rejected_ right-padded, prompt_
    left-padded; ids -> pad_token_id, labels -> label_pad_token_id, masks -> 0; *_logps -> float tensor;
    everything else passed through as a list."""
    out = {}
    for k in features[0].keys():
        if k.endswith("_input_ids") or k.endswith("_attention_mask") or k.endswith("_labels"):
            if k.endswith("_input_ids"):
                pad = pad_token_id
            elif k.endswith("_labels"):
                pad = label_pad_token_id
            else:
                pad = 0
            n = max(len(f[k]) for f in features)
            t = torch.full((len(features), n), pad, dtype=torch.long)
            for i, f in enumerate(features):
                v = torch.tensor(f[k], dtype=torch.long)
                if "prompt" in k:
                    t[i, n - len(v):] = v
                else:
                    t[i, : len(v)] = v
            out[k] = t
        elif k.endswith("_logps"):
            out[k] = torch.tensor([f[k] for f in features])
        else:
            out[k] = [f[k] for f in features]
    return out


def concatenated_inputs(batch: dict, label_pad_token_id: int = IGNORE_INDEX, padding_value: int = 0) -> dict:
    """src/vlrlhf/base/trainer.py:124-146 + trl==0.8.1 DPOTrainer.concatenated_inputs (not vendored; call site
    trainer.py:132-134): pad chosen/rejected tensors on the right to the common max length and stack chosen over
    rejected; every image tensor / list is DUPLICATED (trainer.py:138-142)."""
    n = max(batch["chosen_input_ids"].shape[1], batch["rejected_input_ids"].shape[1])
    out = {}
    for field, pad in (("input_ids", padding_value), ("attention_mask", 0), ("labels", label_pad_token_id)):
        parts = []
        for side in ("chosen", "rejected"):
            t = batch[f"{side}_{field}"]
            if t.shape[1] < n:
                t = torch.cat([t, torch.full((t.shape[0], n - t.shape[1]), pad, dtype=t.dtype)], dim=1)
            parts.append(t)
        out[f"concatenated_{field}"] = torch.cat(parts, dim=0)
    if "img_input_dict" in batch:
        d = {}
        for k, v in batch["img_input_dict"].items():
            if isinstance(v, torch.Tensor):
                d[k] = torch.cat([v, v], dim=0)
            elif isinstance(v, list):
                d[k] = v + v
            else:
                raise ValueError(f"Unsupported type {type(v)} for concatenation.")
        out["concatenated_img_input_dict"] = d
    return out


# ----------------------------------------------------------------------------------------------------------
# Vision tower + projector (third-party arithmetic: transformers CLIPVisionModel, LlavaMultiModalProjector;
# call sites src/vlrlhf/models/Llava/__init__.py:178-191)
# ----------------------------------------------------------------------------------------------------------
def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def clip_vit_features(pixel_values, W: Dict[str, torch.Tensor], cfg: dict, emulate_bf16=False,
                      prefix="vision_tower.vision_model."):
    """hidden_states[-2] of the CLIP ViT with the CLS token dropped (Llava/__init__.py:178-183).  Pre-LN encoder:
    x += out_proj(attn(LN1(x))); x += fc2(quick_gelu(fc1(LN2(x)))).  Only layers 0..L-2 are evaluated."""
    # tag "vit" = the whole tower + projector; finer (error-budget runs): "vit_op" MFMA operands only, "vit_resid" the tower's residual
    # stream, "vit_out" the projector output (= the image rows of the merged embeddings)
    whole = (emulate_bf16 is True) or (isinstance(emulate_bf16, (set, frozenset)) and "vit" in emulate_bf16)
    r = lambda t: _rt(t, True if whole else emulate_bf16, "vit_op")  # noqa: E731
    rr = lambda t: _rt(t, True if whole else emulate_bf16, "vit_resid")  # noqa: E731
    B = pixel_values.shape[0]
    D, P, nh = cfg["vit_hidden"], cfg["patch_size"], cfg["vit_heads"]
    g = cfg["image_size"] // P
    w_pe = W[prefix + "embeddings.patch_embedding.weight"].reshape(D, -1)          # [D, 3*P*P]
    patches = pixel_values.reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P)
    x = r(patches) @ r(w_pe).t()
    cls = W[prefix + "embeddings.class_embedding"].reshape(1, 1, D).expand(B, 1, D)
    x = torch.cat([cls, x], dim=1) + W[prefix + "embeddings.position_embedding.weight"][None]
    eps = cfg.get("vit_ln_eps", 1e-5)
    x = rr(F.layer_norm(x, (D,), W[prefix + "pre_layrnorm.weight"], W[prefix + "pre_layrnorm.bias"], eps))
    hd = D // nh
    n_eval = cfg["vit_layers"] + 1 + cfg.get("vit_feature_layer", -2)   # vision_feature_layer = -2 (LLaVA); -1 = every layer (InternLM-XC2)
    for i in range(n_eval):
        p = f"{prefix}encoder.layers.{i}."
        h = r(F.layer_norm(x, (D,), W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], eps))
        q = r(h @ r(W[p + "self_attn.q_proj.weight"]).t() + W[p + "self_attn.q_proj.bias"])
        k = r(h @ r(W[p + "self_attn.k_proj.weight"]).t() + W[p + "self_attn.k_proj.bias"])
        v = r(h @ r(W[p + "self_attn.v_proj.weight"]).t() + W[p + "self_attn.v_proj.bias"])
        T = x.shape[1]
        q, k, v = (t.reshape(B, T, nh, hd).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1) @ v
        a = r(a.transpose(1, 2).reshape(B, T, D))
        x = rr(x + (a @ r(W[p + "self_attn.out_proj.weight"]).t() + W[p + "self_attn.out_proj.bias"]))
        h = r(F.layer_norm(x, (D,), W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], eps))
        h = r(quick_gelu(h @ r(W[p + "mlp.fc1.weight"]).t() + W[p + "mlp.fc1.bias"]))
        x = rr(x + (h @ r(W[p + "mlp.fc2.weight"]).t() + W[p + "mlp.fc2.bias"]))
    return x[:, 1:]


def projector(feat, W, emulate_bf16=False, prefix="multi_modal_projector."):
    """Linear -> GELU(erf) -> Linear (transformers LlavaMultiModalProjector; call site Llava/__init__.py:191)."""
    whole = (emulate_bf16 is True) or (isinstance(emulate_bf16, (set, frozenset)) and "vit" in emulate_bf16)
    r = lambda t: _rt(t, True if whole else emulate_bf16, "vit_op")  # noqa: E731
    h = r(F.gelu(r(feat) @ r(W[prefix + "linear_1.weight"]).t() + W[prefix + "linear_1.bias"]))
    return _rt(h @ r(W[prefix + "linear_2.weight"]).t() + W[prefix + "linear_2.bias"], True if whole else emulate_bf16, "vit_out")


# ----------------------------------------------------------------------------------------------------------
# Image/text merge
# ----------------------------------------------------------------------------------------------------------
def merge_input_ids_with_image_features(image_features, inputs_embeds, input_ids, attention_mask, labels,
                                        image_token_index: int, pad_token_id: int,
                                        ignore_index: int = IGNORE_INDEX):
    """src/vlrlhf/models/Llava/__init__.py:36-109, restated position-wise.

    Every `<image>` id expands to P = image_features.shape[1] slots.  Rows are right-aligned ("left padding")
    unless some row ends with the MODEL's pad id (:39) - the collator pads with 0/unk, the model pad id is 32001,
    so in training the branch is always "left" and the offset is zero when every row has one image.
    Returns (embeds, attention_mask, labels, position_ids, image_position_map) in the reference's order (:109).

    Difference kept on purpose: the reference finds the image slots as "rows of the output that are still all
    zero" (:87-88), so a text token whose embedding row is exactly zero would be mistaken for an image slot and
    trip the count check (:90-94).  Here slots are found by position; the same ValueError is raised when the
    number of image slots and image features disagree."""
    n_img, P, H = image_features.shape
    B, T = input_ids.shape
    left_padding = not bool((input_ids[:, -1] == pad_token_id).any())
    is_img = input_ids == image_token_index
    S = int(is_img.sum(-1).max()) * (P - 1) + T
    new_pos = torch.cumsum(is_img.long() * (P - 1) + 1, dim=-1) - 1
    nb_image_pad = S - 1 - new_pos[:, -1]
    if left_padding:
        new_pos = new_pos + nb_image_pad[:, None]
    out = torch.zeros(B, S, H, dtype=inputs_embeds.dtype)
    out_mask = torch.zeros(B, S, dtype=attention_mask.dtype)
    out_labels = torch.full((B, S), ignore_index, dtype=input_ids.dtype)
    written = torch.zeros(B, S, dtype=torch.bool)
    bi, ti = torch.where(~is_img)
    dst = new_pos[bi, ti]
    out[bi, dst] = inputs_embeds[bi, ti]
    out_mask[bi, dst] = attention_mask[bi, ti]
    if labels is not None:
        out_labels[bi, dst] = labels[bi, ti]
    written[bi, dst] = True
    free = ~written
    img_map = free & ((free.long().cumsum(-1) - 1) >= nb_image_pad[:, None])
    if int(img_map.sum()) != n_img * P:
        raise ValueError(
            f"The input provided to the model are wrong. The number of image tokens is {int(is_img.sum())} while"
            f" the number of image given to the model is {n_img}. This prevents correct indexing and breaks batch"
            " generation.")
    out[img_map] = image_features.reshape(-1, H).to(out.dtype)
    out_mask = out_mask | img_map.to(out_mask.dtype)
    position_ids = (out_mask.cumsum(-1) - 1).masked_fill(out_mask == 0, 1)
    bi, ti = torch.where(input_ids == pad_token_id)
    out[bi, new_pos[bi, ti]] = 0
    return out, out_mask, (out_labels if labels is not None else None), position_ids, img_map


# ----------------------------------------------------------------------------------------------------------
# LLaMA decoder (third-party arithmetic: transformers LlamaForCausalLM; call site Llava/__init__.py:232-243)
# ----------------------------------------------------------------------------------------------------------
def rms_norm(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def rope_tables(position_ids, head_dim, theta=10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = position_ids[..., None].float() * inv          # [B,S,hd/2]
    emb = torch.cat([ang, ang], dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(x, cos, sin):
    """x [B,nh,S,hd]; rotate-half convention."""
    hd = x.shape[-1]
    x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
    rot = torch.cat([-x2, x1], dim=-1)
    return x * cos[:, None] + rot * sin[:, None]


def causal_padding_bias(attention_mask):
    """additive [B,1,S,S]: key j visible to query i iff j <= i and attention_mask[b,j] != 0."""
    B, S = attention_mask.shape
    causal = torch.ones(S, S, dtype=torch.bool).tril()
    vis = causal[None] & (attention_mask[:, None, :] != 0)
    return torch.zeros(B, 1, S, S).masked_fill(~vis[:, None], torch.finfo(torch.float32).min)


LORA_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
_M64 = (1 << 64) - 1


def _mix64(z):
    """splitmix64 finaliser on Python ints / numpy uint64 arrays (wrapping arithmetic)."""
    import numpy as np
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def dropout_mask(seed: int, n: int, p: float):
    """keep-mask (uint8 [n], n % 8 == 0) of vlr_dropout(seed): group g of 8 consecutive elements draws
    r_j = mix64(key ^ (2g+j)), key = mix64(seed); element e keeps iff the 16-bit field e%4 of r_{e//4} >= round(p*65536)."""
    import numpy as np
    assert n % 8 == 0
    key = _mix64(np.array([seed & _M64], dtype=np.uint64))[0]
    g = np.arange(n // 8, dtype=np.uint64)
    thr = np.uint64(int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)))
    out = np.empty((n // 8, 8), dtype=np.uint8)
    for j in range(2):
        rj = _mix64(key ^ (np.uint64(2) * g + np.uint64(j)))
        for e in range(4):
            out[:, 4 * j + e] = ((rj >> np.uint64(16 * e)) & np.uint64(0xFFFF)) >= thr
    return torch.from_numpy(out.reshape(-1))


def lora_names(layer: int, target: str, prefix="base_model.model.language_model.model.layers."):
    mod = "self_attn" if target in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp"
    base = f"{prefix}{layer}.{mod}.{target}"
    return base + ".lora_A.weight", base + ".lora_B.weight"


def lora_delta(h, lora, layer, target, r):
    """peft lora.Linear.forward adapter term: lora_B(lora_A(dropout(x))) * scaling (tuners/lora/layer.py); `lora` =
    dict(W={peft name: tensor}, scale=lora_alpha/r, dropout=p, seed=None|int).  With a seed the dropout mask of target
    t in layer l is dropout_mask(seed + 8*l + t) over the flattened [B*S, in] input (the HIP path's convention)."""
    na, nb = lora_names(layer, target)
    A, B = r(lora["W"][na], "w"), r(lora["W"][nb], "w")
    p = float(lora.get("dropout", 0.0) or 0.0)
    if p > 0 and lora.get("seed") is not None:
        m = dropout_mask(lora["seed"] + 8 * layer + LORA_TARGETS.index(target), h.numel(), p).view(h.shape).to(h.dtype)
        h = h * m                      # (exact: the HIP path zeroes the dropped elements of the staged operand; 1 / (1 - p) rides in the scale)
        return r(lora["scale"] / (1.0 - p) * (h @ A.t()), "lora_u") @ B.t()
    return r(lora["scale"] * (h @ A.t()), "lora_u") @ B.t()


def lora_merged_weights(W, lora, cfg, prefix="language_model.model.layers."):
    """peft merge: W + scaling * B A for every adapted linear (dropout-free equivalent of the adapter path)."""
    out = dict(W)
    for l in range(cfg["layers"]):
        for t in LORA_TARGETS:
            na, nb = lora_names(l, t)
            mod = "self_attn" if t in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp"
            k = f"{prefix}{l}.{mod}.{t}.weight"
            out[k] = W[k] + lora["scale"] * (lora["W"][nb] @ lora["W"][na])
    return out


def random_lora(cfg, r, alpha, seed=0, b_std=0.0, dropout=0.0):
    """peft init: A ~ U(-1/sqrt(in), 1/sqrt(in)) (kaiming_uniform a=sqrt(5)), B = 0 (b_std > 0: N(0, b_std) for tests)."""
    g = torch.Generator().manual_seed(seed)
    H, I = cfg["hidden"], cfg["inter"]
    nh = cfg["heads"]
    hd = cfg.get("head_dim", H // nh)
    Nq, Nkv = nh * hd, cfg.get("kv_heads", nh) * hd
    dims = dict(q_proj=(Nq, H), k_proj=(Nkv, H), v_proj=(Nkv, H), o_proj=(H, Nq), gate_proj=(I, H), up_proj=(I, H), down_proj=(H, I))
    Wl = {}
    for l in range(cfg["layers"]):
        for t in LORA_TARGETS:
            out, inn = dims[t]
            na, nb = lora_names(l, t)
            Wl[na] = (torch.rand(r, inn, generator=g) * 2 - 1) / math.sqrt(inn)
            Wl[nb] = torch.randn(out, r, generator=g) * b_std if b_std > 0 else torch.zeros(out, r)
    return dict(W=Wl, scale=float(alpha) / r, dropout=dropout, seed=None, r=r)


def llama_hidden(embeds, attention_mask, position_ids, W, cfg, emulate_bf16=False,
                 prefix="language_model.model.", collect=None, lora=None):
    """All decoder layers + final RMSNorm -> hidden [B,S,H] (what lm_head consumes).  `kv_heads` < heads = grouped-query
    attention (Mistral / InternLM2): k, v have kv_heads heads, each shared by heads/kv_heads query heads."""
    r = lambda t, tag=None: _rt(t, emulate_bf16, tag)  # noqa: E731
    B, S, H = embeds.shape
    nh = cfg["heads"]
    nkv = cfg.get("kv_heads", nh)
    hd = cfg.get("head_dim", H // nh)
    eps = cfg.get("rms_eps", 1e-5)
    cos, sin = rope_tables(position_ids, hd, cfg.get("rope_theta", 10000.0))
    bias = causal_padding_bias(attention_mask)
    x = r(embeds, "x0")
    for i in range(cfg["layers"]):
        p = f"{prefix}layers.{i}."
        h = r(rms_norm(x, W[p + "input_layernorm.weight"], eps), "xn")
        q = r(h @ r(W[p + "self_attn.q_proj.weight"], "w").t(), "qkv")
        k = r(h @ r(W[p + "self_attn.k_proj.weight"], "w").t(), "qkv")
        v = r(h @ r(W[p + "self_attn.v_proj.weight"], "w").t(), "qkv")
        if lora is not None:
            q, k, v = (r(y + lora_delta(h, lora, i, t, r), "qkv") for y, t in ((q, "q_proj"), (k, "k_proj"), (v, "v_proj")))
        v = r(v, "v")          # tag "v": the value projection alone (an MFMA operand as stored; q / k are operands only after RoPE)
        q = q.reshape(B, S, nh, hd).transpose(1, 2)
        k, v = (t.reshape(B, S, nkv, hd).transpose(1, 2) for t in (k, v))
        q, k = r(apply_rope(q, cos, sin), "rope"), r(apply_rope(k, cos, sin), "rope")
        if nkv != nh:
            k, v = (t.repeat_interleave(nh // nkv, dim=1) for t in (k, v))
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd) + bias, dim=-1)
        if isinstance(emulate_bf16, (set, frozenset)):
            att = r(att, "p")
        a = r((att @ v).transpose(1, 2).reshape(B, S, nh * hd), "attn")
        x = r(x + a @ r(W[p + "self_attn.o_proj.weight"], "w").t(), "resid")
        if lora is not None:
            x = r(x + lora_delta(a, lora, i, "o_proj", r), "resid")
        h = r(rms_norm(x, W[p + "post_attention_layernorm.weight"], eps), "xn")
        gate = r(h @ r(W[p + "mlp.gate_proj.weight"], "w").t(), "gu")
        up = r(h @ r(W[p + "mlp.up_proj.weight"], "w").t(), "gu")
        if lora is not None:
            gate, up = r(gate + lora_delta(h, lora, i, "gate_proj", r), "gu"), r(up + lora_delta(h, lora, i, "up_proj", r), "gu")
        act = r(F.silu(gate) * up, "act")
        x = r(x + act @ r(W[p + "mlp.down_proj.weight"], "w").t(), "resid")
        if lora is not None:
            x = r(x + lora_delta(act, lora, i, "down_proj", r), "resid")
        if collect is not None:
            collect.append(x)
    return r(rms_norm(x, W[prefix + "norm.weight"], eps), "hidden")


def lm_logits(hidden, W, emulate_bf16=False, key="language_model.lm_head.weight"):
    return hidden @ _rt(W[key], emulate_bf16, "w").t()


# ----------------------------------------------------------------------------------------------------------
# Log-probabilities and the DPO loss
# ----------------------------------------------------------------------------------------------------------
def get_diff_ids(a_seq, b_seq, min_match_size=3):
    """src/vlrlhf/utils/diff_lib.py:116-125,73-83,135-163,173-180: indices of a_seq / b_seq that lie in a
    'replace' span - a gap between two kept matching blocks (SequenceMatcher blocks shorter than
    min_match_size are dropped, the terminating empty block is kept) that is non-empty on BOTH sides."""
    blocks = difflib.SequenceMatcher(None, a_seq, b_seq).get_matching_blocks()
    kept = [m for m in blocks[:-1] if m[2] >= min_match_size] + [blocks[-1]]
    a_ids, b_ids = [], []
    a_prev, b_prev = 0, 0
    for m in kept:
        a_gap, b_gap = (a_prev, m[0]), (b_prev, m[1])
        if a_gap[0] != a_gap[1] and b_gap[0] != b_gap[1]:
            a_ids += range(*a_gap)
            b_ids += range(*b_gap)
        a_prev, b_prev = m[0] + m[2], m[1] + m[2]
    # trailing gap after the last kept block (diff_lib.py:73-83 closes the span list at len(seq)); the last kept
    # block is the zero-length terminator at (len(a), len(b)), so this gap is empty - kept for fidelity.
    a_gap, b_gap = (a_prev, len(a_seq)), (b_prev, len(b_seq))
    if a_gap[0] != a_gap[1] and b_gap[0] != b_gap[1]:
        a_ids += range(*a_gap)
        b_ids += range(*b_gap)
    return sorted(set(a_ids)), sorted(set(b_ids))


def ddpo_shared_mask(labels, label_pad_token_id=IGNORE_INDEX, min_match_size=3):
    """src/vlrlhf/base/trainer.py:161-184: mask (on the SHIFTED labels, pad -> 0) of tokens that differ between
    the chosen half and the rejected half of the batch."""
    sh = labels[:, 1:].clone()
    sh[sh == label_pad_token_id] = 0
    n = sh.shape[0] // 2
    assert n * 2 == sh.shape[0]
    m = torch.zeros_like(sh, dtype=torch.bool)
    for i in range(n):
        c, rj = get_diff_ids(sh[i].tolist(), sh[n + i].tolist(), min_match_size)
        m[i, c] = True
        m[n + i, rj] = True
    return m


def get_batch_logps(logits, labels, average_log_prob=False, label_pad_token_id=IGNORE_INDEX,
                    mask_shared_tokens=False):
    """src/vlrlhf/base/trainer.py:148-188 (decoder-only)."""
    if logits.shape[:-1] != labels.shape:
        raise ValueError("Logits (batch and sequence length dim) and labels must have the same shape.")
    tgt = labels[:, 1:].clone()
    lg = logits[:, :-1, :]
    mask = tgt != label_pad_token_id
    tgt[~mask] = 0
    tok = torch.gather(lg.float().log_softmax(-1), 2, tgt[..., None]).squeeze(2)
    if mask_shared_tokens:
        mask = mask & ddpo_shared_mask(labels, label_pad_token_id)
    s = (tok * mask).sum(-1)
    return s / mask.sum(-1) if average_log_prob else s


def dpo_loss(policy_chosen_logps, policy_rejected_logps, reference_chosen_logps, reference_rejected_logps,
             beta=0.1, label_smoothing=0.0, loss_type="sigmoid", reference_free=False):
    """src/vlrlhf/base/trainer.py:244-301."""
    pi = policy_chosen_logps - policy_rejected_logps
    ref = torch.zeros(1, dtype=pi.dtype) if reference_free else reference_chosen_logps - reference_rejected_logps
    x = pi - ref
    if loss_type in ("sigmoid", "ddpo"):
        losses = -F.logsigmoid(beta * x) * (1 - label_smoothing) - F.logsigmoid(-beta * x) * label_smoothing
    elif loss_type == "hinge":
        losses = torch.relu(1 - beta * x)
    elif loss_type == "ipo":
        losses = (x - 1 / (2 * beta)) ** 2
    elif loss_type == "kto_pair":
        chosen_kl = (policy_chosen_logps - reference_chosen_logps).mean().clamp(min=0)
        rejected_kl = (policy_rejected_logps - reference_rejected_logps).mean().clamp(min=0)
        cl = policy_chosen_logps - reference_chosen_logps
        rl = policy_rejected_logps - reference_rejected_logps
        losses = torch.cat((1 - torch.sigmoid(beta * (cl - rejected_kl)), 1 - torch.sigmoid(beta * (chosen_kl - rl))), 0)
    else:
        raise ValueError(f"Unknown loss type: {loss_type}. Should be one of ['sigmoid', 'hinge', 'ipo', 'kto_pair']")
    chosen_rewards = beta * (policy_chosen_logps - reference_chosen_logps).detach()
    rejected_rewards = beta * (policy_rejected_logps - reference_rejected_logps).detach()
    return losses, chosen_rewards, rejected_rewards


# ----------------------------------------------------------------------------------------------------------
# Whole model / whole step
# ----------------------------------------------------------------------------------------------------------
def llava_forward(W, cfg, input_ids, attention_mask, labels, pixel_values, emulate_bf16=False,
                  dedupe_images=True, return_hidden=False, lora=None, collect=None):
    """src/vlrlhf/models/Llava/__init__.py:111-271 on the training path: embed -> ViT(hidden_states[-2], no CLS)
    -> projector -> merge -> decoder -> logits.  Returns (logits fp32, merged labels, aux).
    `dedupe_images`: the concatenated batch carries every image twice (trainer.py:138-142); the ViT is frozen and
    deterministic so it is evaluated once per distinct image and the features are repeated - identical result."""
    emb = _rt(W["language_model.model.embed_tokens.weight"][input_ids], emulate_bf16, "w")
    n = pixel_values.shape[0]
    if dedupe_images and n % 2 == 0 and torch.equal(pixel_values[: n // 2], pixel_values[n // 2:]):
        feat = clip_vit_features(pixel_values[: n // 2], W, cfg, emulate_bf16)
        img = projector(feat, W, emulate_bf16)
        img = torch.cat([img, img], dim=0)
    else:
        feat = clip_vit_features(pixel_values, W, cfg, emulate_bf16)
        img = projector(feat, W, emulate_bf16)
    merged, mask, mlabels, pos, img_map = merge_input_ids_with_image_features(
        img, emb, input_ids, attention_mask, labels, cfg["image_token"], cfg.get("model_pad_token_id", cfg["image_token"] + 1))
    hidden = llama_hidden(merged, mask, pos, W, cfg, emulate_bf16, lora=lora, collect=collect)
    aux = dict(vit_feat=feat, image_features=img, merged=merged, mask=mask, pos=pos, img_map=img_map, hidden=hidden)
    if return_hidden:
        return hidden, mlabels, aux
    return lm_logits(hidden, W, emulate_bf16), mlabels, aux


# ----------------------------------------------------------------------------------------------------------
# LLaVA-Next (anyres tiles + Mistral GQA): the deltas of src/vlrlhf/models/LlavaNext/__init__.py on the DPO path
# ----------------------------------------------------------------------------------------------------------
def select_best_resolution(original_size, possible_resolutions):
    """transformers image_processing_utils.select_best_resolution (used by image_size_to_num_patches /
    get_anyres_image_grid_shape; call sites LlavaNext/__init__.py:216-222 and HF pack_image_features): the (height, width) of
    the pinpoint that keeps the most of the image and then wastes the least."""
    oh, ow = original_size
    best, max_eff, min_waste = None, 0, float("inf")
    for h, w in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            best, max_eff, min_waste = (h, w), eff, waste
    return best


def anyres_num_patches(image_size, pinpoints, tile):
    """HF image_size_to_num_patches: tiles of the best resolution + the base image"""
    h, w = select_best_resolution(tuple(int(x) for x in image_size), pinpoints)
    return len(range(0, h, tile)) * len(range(0, w, tile)) + 1


def pack_image_features(image_features, image_sizes, newline, cfg):
    """HF LlavaNext pack_image_features ("spatial_unpad"; call site LlavaNext/__init__.py:255-259): per image
    [base | tiles laid out on their grid, un-padded to the original aspect ratio, one `image_newline` column per row]."""
    g = cfg["image_size"] // cfg["patch_size"]
    out, lens = [], []
    for feat, size in zip(image_features, image_sizes):
        if feat.shape[0] > 1:
            base, tiles = feat[0], feat[1:]
            bh, bw = select_best_resolution(tuple(int(x) for x in size), cfg["image_grid_pinpoints"])
            nph, npw = bh // cfg["image_size"], bw // cfg["image_size"]
            t = tiles.view(nph, npw, g, g, -1).permute(4, 0, 2, 1, 3).reshape(tiles.shape[-1], nph * g, npw * g)
            oh, ow = (int(x) for x in size)
            ch, cw = t.shape[1:]
            if ow / oh > cw / ch:
                nh_ = int(round(oh * (cw / ow), 7))
                pad = (ch - nh_) // 2
                t = t[:, pad: ch - pad, :]
            else:
                nw_ = int(round(ow * (ch / oh), 7))
                pad = (cw - nw_) // 2
                t = t[:, :, pad: cw - pad]
            t = torch.cat((t, newline[:, None, None].expand(-1, t.shape[1], 1).to(t.dtype)), dim=-1)
            f = torch.cat((base, t.flatten(1, 2).transpose(0, 1)), dim=0)
        else:
            f = torch.cat((feat[0], newline[None].to(feat.dtype)), dim=0)
        out.append(f)
        lens.append(f.shape[0])
    return out, torch.tensor(lens, dtype=torch.long)


def llavanext_merge(image_features, feature_lens, inputs_embeds, input_ids, attention_mask, labels, image_token_index,
                    padding_side="left", ignore_index=IGNORE_INDEX):
    """src/vlrlhf/models/LlavaNext/__init__.py:38-171 restated position-wise.  Differences from the LLaVA-1.5 merge: every
    `<image>` expands to ITS OWN feature_lens[i] slots; padding is recognised from the attention mask (:57-71: zeros on the
    left -> rows end-aligned, zeros on the right -> start-aligned, neither -> `padding_side`); padded text tokens are not
    copied at all; the image slots of a row are exactly the free slots inside its [valid length] window (:141-150).
    Returns (embeds, mask, position_ids, labels, image_to_overwrite) in the reference's order (:171)."""
    B, T = input_ids.shape
    H = inputs_embeds.shape[-1]
    if int(feature_lens.sum()) != image_features.shape[0]:
        raise ValueError(f"feature_lens={feature_lens} / {int(feature_lens.sum())} != image_features.shape={tuple(image_features.shape)}")
    lpad, rpad = bool((attention_mask[:, 0] == 0).any()), bool((attention_mask[:, -1] == 0).any())
    left = True
    if B > 1:
        if lpad and not rpad:
            left = True
        elif rpad and not lpad:
            left = False
        elif not lpad and not rpad:
            left = padding_side == "left"
        else:
            raise ValueError(f"both side of attention_mask has zero, invalid. {attention_mask}")
    is_img = input_ids == image_token_index
    n_img_row = is_img.sum(-1)
    if int(is_img.sum()) != feature_lens.shape[0]:
        raise ValueError(f"Number of image tokens in input_ids ({int(is_img.sum())}) different from num_images ({feature_lens.shape[0]}).")
    per_row = torch.split(feature_lens, n_img_row.tolist())
    seq_len = (attention_mask == 1).long().sum(-1) - n_img_row + torch.tensor([int(x.sum()) for x in per_row])
    S = int(seq_len.max())
    step = torch.ones(B, T, dtype=torch.long)
    step[is_img] = feature_lens
    new_pos = torch.cumsum(step, -1) - 1
    if left:
        new_pos = new_pos + (S - 1 - new_pos[:, -1:])
    out = torch.zeros(B, S, H, dtype=inputs_embeds.dtype)
    out_mask = torch.zeros(B, S, dtype=attention_mask.dtype)
    out_labels = torch.full((B, S), ignore_index, dtype=torch.long)
    bi, ti = torch.where((~is_img) & (attention_mask == 1))
    dst = new_pos[bi, ti]
    out[bi, dst] = inputs_embeds[bi, ti]
    out_mask[bi, dst] = attention_mask[bi, ti]
    if labels is not None:
        out_labels[bi, dst] = labels[bi, ti]
    free = torch.ones(B, S, dtype=torch.bool)
    free[bi, dst] = False
    idx = torch.arange(S)[None].expand(B, S)
    free &= ((S - idx) <= seq_len[:, None]) if left else (idx < seq_len[:, None])
    if int(free.sum()) != image_features.shape[0]:
        raise ValueError(f"image_to_overwrite.sum()={int(free.sum())} != num_image_features={image_features.shape[0]} The input provided to "
                         "the model are wrong. This prevents correct indexing and breaks batch generation.")
    out[free] = image_features.to(out.dtype)
    out_mask = out_mask | free.to(out_mask.dtype)
    position_ids = (out_mask.cumsum(-1) - 1).masked_fill(out_mask == 0, 1)
    return out, out_mask, position_ids, (out_labels if labels is not None else None), free


def llavanext_forward(W, cfg, input_ids, attention_mask, labels, pixel_values, image_sizes, emulate_bf16=False,
                      dedupe_images=True, collect=None, lora=None):
    """LlavaNextForRL.forward on the training path (LlavaNext/__init__.py:205-265, 306-316): embeddings with `<image>` ids
    mapped to 0 (:209-211), ViT on the first num_patches tiles of every image, projector, pack, merge, Mistral decoder
    (grouped-query attention: cfg['kv_heads']).  pixel_values [n_img, max_patches, 3, s, s], image_sizes [n_img, 2]."""
    ids0 = input_ids.clone()
    ids0[input_ids == cfg["image_token"]] = 0
    emb = _rt(W["language_model.model.embed_tokens.weight"][ids0], emulate_bf16, "w")
    n = pixel_values.shape[0]
    dup = dedupe_images and n % 2 == 0 and torch.equal(pixel_values[: n // 2], pixel_values[n // 2:]) and torch.equal(image_sizes[: n // 2], image_sizes[n // 2:])
    nu = n // 2 if dup else n
    npatch = [anyres_num_patches(sz, cfg["image_grid_pinpoints"], cfg["image_size"]) for sz in image_sizes[:nu]]
    pv = torch.cat([x[:k] for x, k in zip(pixel_values[:nu], npatch)], dim=0)
    feat = clip_vit_features(pv, W, cfg, emulate_bf16)
    img = projector(feat, W, emulate_bf16)
    packed, lens = pack_image_features(torch.split(img, npatch, dim=0), image_sizes[:nu], _rt(W["image_newline"], emulate_bf16, "w"), cfg)
    packed = torch.cat(packed, dim=0)
    if dup:
        packed, lens = torch.cat([packed, packed], 0), torch.cat([lens, lens], 0)
    merged, mask, pos, mlabels, img_map = llavanext_merge(packed, lens, emb, input_ids, attention_mask, labels, cfg["image_token"],
                                                          cfg.get("padding_side", "left"))
    hidden = llama_hidden(merged, mask, pos, W, cfg, emulate_bf16, collect=collect, lora=lora)
    aux = dict(vit_feat=feat, projected=img, packed=packed, feature_lens=lens, merged=merged, mask=mask, pos=pos, img_map=img_map,
               hidden=hidden, num_patches=npatch)
    return lm_logits(hidden, W, emulate_bf16), mlabels, aux


def concatenated_forward(W, cfg, batch, loss_type="sigmoid", emulate_bf16=False, lora=None, collect=None):
    """src/vlrlhf/base/trainer.py:190-242 -> (chosen_logps, rejected_logps, chosen_logits, rejected_logits)."""
    cb = concatenated_inputs(batch)
    n = batch["chosen_labels"].shape[0]
    if cfg.get("image_grid_pinpoints"):      # LLaVA-Next
        logits, labels, _ = llavanext_forward(W, cfg, cb["concatenated_input_ids"], cb["concatenated_attention_mask"],
                                              cb["concatenated_labels"], cb["concatenated_img_input_dict"]["pixel_values"],
                                              cb["concatenated_img_input_dict"]["image_sizes"], emulate_bf16, collect=collect, lora=lora)
    else:
        logits, labels, _ = llava_forward(W, cfg, cb["concatenated_input_ids"], cb["concatenated_attention_mask"],
                                          cb["concatenated_labels"], cb["concatenated_img_input_dict"]["pixel_values"],
                                          emulate_bf16, lora=lora, collect=collect)
    lp = get_batch_logps(logits, labels, mask_shared_tokens=(loss_type == "ddpo"))
    return lp[:n], lp[n:], logits[:n], logits[n:]


def compute_loss(W_policy, W_ref, cfg, batch, beta=0.1, loss_type="sigmoid", label_smoothing=0.0,
                 reference_free=False, emulate_bf16=False, lora=None):
    """trl==0.8.1 DPOTrainer.get_batch_loss_metrics (not vendored; reached from src/vlrlhf/base/trainer.py:303-305):
    policy pass with grad, reference pass without (or batch['reference_*_logps'] when present), dpo_loss,
    loss = losses.mean(), eight metrics.  With `lora` the policy is W_policy + adapters and the reference pass is
    W_ref (= the same base weights) with the adapters disabled (trl null_ref_context)."""
    pc, pr, pcl, prl = concatenated_forward(W_policy, cfg, batch, loss_type, emulate_bf16, lora=lora)
    with torch.no_grad():
        if "reference_chosen_logps" in batch and "reference_rejected_logps" in batch:
            rc, rr = batch["reference_chosen_logps"], batch["reference_rejected_logps"]
        else:
            rc, rr, _, _ = concatenated_forward(W_ref, cfg, batch, loss_type, emulate_bf16)
    losses, cr, rrw = dpo_loss(pc, pr, rc, rr, beta, label_smoothing, loss_type, reference_free)
    metrics = {
        "rewards/chosen": cr.mean(), "rewards/rejected": rrw.mean(),
        "rewards/accuracies": (cr > rrw).float().mean(), "rewards/margins": (cr - rrw).mean(),
        "logps/rejected": pr.detach().mean(), "logps/chosen": pc.detach().mean(),
        "logits/rejected": prl.detach().mean(), "logits/chosen": pcl.detach().mean(),
    }
    return losses.mean(), metrics


def trainable_names(W, freeze_vision_tower=True):
    return [k for k in W if not (freeze_vision_tower and k.startswith("vision_tower."))]


def is_no_decay(name: str) -> bool:
    """transformers Trainer.get_decay_parameter_names: LayerNorm/RMSNorm weights and biases get no weight decay."""
    return ("norm" in name) or name.endswith(".bias")


def clip_grad_norm_(grads: Dict[str, torch.Tensor], max_norm: float):
    """torch.nn.utils.clip_grad_norm_ (L2): total = sqrt(sum g^2); g *= min(1, max_norm / (total + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def adamw_step(W, grads, state, lr, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.0, step: Optional[int] = None):
    """torch.optim.AdamW single step (decoupled decay): p *= 1 - lr*wd; m,v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)."""
    state["step"] = (state.get("step", 0) + 1) if step is None else step
    t = state["step"]
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    for k, g in grads.items():
        m = state.setdefault("m." + k, torch.zeros_like(W[k]))
        v = state.setdefault("v." + k, torch.zeros_like(W[k]))
        wd = 0.0 if is_no_decay(k) else weight_decay
        W[k].mul_(1 - lr * wd)
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        W[k].addcdiv_(m, denom, value=-lr / bc1)


def dpo_train_step(W_policy, W_ref, cfg, batch, optim: dict, state: dict, beta=0.1, loss_type="sigmoid",
                   emulate_bf16=False):
    """One full optimizer step on the CPU: forward x2, loss, backward (torch autograd on the restated forward),
    clip, AdamW.  Returns (loss, metrics, grads, total_norm)."""
    names = trainable_names(W_policy)
    leaves = {k: W_policy[k].detach().clone().requires_grad_(True) for k in names}
    Wp = dict(W_policy)
    Wp.update(leaves)
    loss, metrics = compute_loss(Wp, W_ref, cfg, batch, beta, loss_type, emulate_bf16=emulate_bf16)
    loss.backward()
    grads = {k: leaves[k].grad for k in names if leaves[k].grad is not None}
    total = clip_grad_norm_(grads, optim["max_grad_norm"])
    with torch.no_grad():
        adamw_step(W_policy, grads, state, optim["lr"], optim["beta1"], optim["beta2"], optim["eps"],
                   optim["weight_decay"])
    return loss.detach(), metrics, grads, total


# ----------------------------------------------------------------------------------------------------------
# Synthetic workload of SURVEY.md section 8(d) (shared by tests and bench so both sides see identical inputs)
# ----------------------------------------------------------------------------------------------------------
def synthetic_pixels(n, image_size, seed):
    import numpy as np
    g = np.random.Generator(np.random.PCG64(seed))
    px = g.integers(0, 256, size=(n, 3, image_size, image_size), dtype=np.uint8).astype(np.float32) / 255.0
    mean = np.array(CLIP_MEAN, dtype=np.float32)[None, :, None, None]
    std = np.array(CLIP_STD, dtype=np.float32)[None, :, None, None]
    return torch.from_numpy((px - mean) / std)


def synthetic_batch(pairs, text_len, image_token, vocab_hi, image_size, seed, ragged=False, prompt_frac=0.5):
    """BOS at 0, one <image> at index 4, ids U{3..vocab_hi-1}; prompt shared by chosen and rejected; labels = ids
    with the prompt -> -100.  ragged: response lengths U{text_len/8 .. text_len/2}, right-padded 0 / -100 / 0."""
    import numpy as np
    g = np.random.Generator(np.random.PCG64(seed))
    lp = int(text_len * prompt_frac)
    rows = []
    for _ in range(pairs):
        prompt = g.integers(3, vocab_hi, size=lp).tolist()
        prompt[0] = 1
        prompt[4] = image_token
        lens = [text_len - lp, text_len - lp]
        if ragged:
            lens = [int(g.integers(max(2, (text_len - lp) // 8), text_len - lp + 1)) for _ in range(2)]
        resp = [g.integers(3, vocab_hi, size=n).tolist() for n in lens]
        rows.append(dict(
            prompt_input_ids=prompt, prompt_attention_mask=[1] * lp,
            chosen_input_ids=prompt + resp[0], chosen_attention_mask=[1] * (lp + lens[0]),
            chosen_labels=[-100] * lp + resp[0],
            rejected_input_ids=prompt + resp[1], rejected_attention_mask=[1] * (lp + lens[1]),
            rejected_labels=[-100] * lp + resp[1], img_path="synthetic"))
    batch = collate(rows)
    batch["img_input_dict"] = dict(pixel_values=synthetic_pixels(pairs, image_size, seed + 7))
    return batch


def random_weights(cfg, seed=0, std=0.02, dtype=torch.float32):
    """Random LLaVA-shaped weights with transformers==4.41.0 checkpoint names."""
    g = torch.Generator().manual_seed(seed)
    W = {}

    def rnd(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    D, P, H, I, V = cfg["vit_hidden"], cfg["patch_size"], cfg["hidden"], cfg["inter"], cfg["vocab"]
    ntok = (cfg["image_size"] // P) ** 2 + 1
    vp = "vision_tower.vision_model."
    W[vp + "embeddings.class_embedding"] = rnd(D)
    W[vp + "embeddings.patch_embedding.weight"] = rnd(D, 3, P, P)
    W[vp + "embeddings.position_embedding.weight"] = rnd(ntok, D)
    for nm in ("pre_layrnorm", "post_layernorm"):
        W[vp + nm + ".weight"] = 1 + rnd(D, s=0.05)
        W[vp + nm + ".bias"] = rnd(D)
    for i in range(cfg["vit_layers"]):
        p = f"{vp}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[p + f"self_attn.{nm}.weight"] = rnd(D, D, s=cfg.get("vit_std", std))
            W[p + f"self_attn.{nm}.bias"] = rnd(D)
        for nm in ("layer_norm1", "layer_norm2"):
            W[p + nm + ".weight"] = 1 + rnd(D, s=0.05)
            W[p + nm + ".bias"] = rnd(D)
        W[p + "mlp.fc1.weight"] = rnd(cfg["vit_mlp"], D, s=cfg.get("vit_std", std))
        W[p + "mlp.fc1.bias"] = rnd(cfg["vit_mlp"])
        W[p + "mlp.fc2.weight"] = rnd(D, cfg["vit_mlp"], s=cfg.get("vit_std", std))
        W[p + "mlp.fc2.bias"] = rnd(D)
    W["multi_modal_projector.linear_1.weight"] = rnd(H, D)
    W["multi_modal_projector.linear_1.bias"] = rnd(H)
    W["multi_modal_projector.linear_2.weight"] = rnd(H, H)
    W["multi_modal_projector.linear_2.bias"] = rnd(H)
    lp = "language_model.model."
    W[lp + "embed_tokens.weight"] = rnd(V, H)
    for i in range(cfg["layers"]):
        p = f"{lp}layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            W[p + f"self_attn.{nm}.weight"] = rnd(H, H)
        W[p + "mlp.gate_proj.weight"] = rnd(I, H)
        W[p + "mlp.up_proj.weight"] = rnd(I, H)
        W[p + "mlp.down_proj.weight"] = rnd(H, I)
        W[p + "input_layernorm.weight"] = 1 + rnd(H, s=0.05)
        W[p + "post_attention_layernorm.weight"] = 1 + rnd(H, s=0.05)
    W[lp + "norm.weight"] = 1 + rnd(H, s=0.05)
    W["language_model.lm_head.weight"] = rnd(V, H)
    return W


# ----------------------------------------------------------------------------------------------------------
# Machine-independent synthetic weights for full-size parity (tests/test_hip_depth.py, oracle/depth_parity.py).
# Integer hash -> Irwin-Hall(4) "normal" -> one fp32 multiply -> bf16: every step is exact or a single correctly
# rounded operation, so a CPU here and the GPU there produce bit-identical tensors from (seed, name) alone - 7B
# weights never have to be stored or shipped.  Restated on the product side in vlrlhf/utils/synthetic.py;
# tests/test_oracle_golden.py checks the two agree bit for bit.
# ----------------------------------------------------------------------------------------------------------
def _hash32_np(x, key):
    import numpy as np
    x ^= np.uint32(key)
    x *= np.uint32(0x45D9F3B)           # uint32 arithmetic wraps by definition
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x45D9F3B)
    x ^= x >> np.uint32(16)
    return x


def name_key(seed: int, name: str):
    import hashlib
    d = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(d[:4], "little"), int.from_bytes(d[4:8], "little")


def hashed_normal(n: int, seed: int, name: str, device="cpu"):
    """~N(0,1) fp32 [n] (sum of four 16-bit uniforms, +-3.46 sigma), a pure function of (seed, name, index):
    h1 = hash32(i ^ k1), h2 = hash32((i + 0x9E3779B9 mod 2^32) ^ k2), u = sum of their four 16-bit halves."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    k1, k2 = name_key(seed, name)
    out = np.empty(n, np.float32)
    step = 1 << 20

    def chunk(a):
        b = min(n, a + step)
        i = np.arange(a, b, dtype=np.uint32)
        h2 = _hash32_np(i + np.uint32(0x9E3779B9), k2)
        h1 = _hash32_np(i, k1)
        u = h1 & np.uint32(0xFFFF)
        u += h1 >> np.uint32(16)
        u += h2 & np.uint32(0xFFFF)
        u += h2 >> np.uint32(16)
        out[a:b] = (u.astype(np.int32) - 131070).astype(np.float32) * np.float32(1.0 / 37837.227)   # sqrt(4 * 65536^2 / 12)

    with ThreadPoolExecutor(max(1, torch.get_num_threads())) as ex:
        list(ex.map(chunk, range(0, n, step)))
    return torch.from_numpy(out)


def hashed_tensor(shape, seed, name, std=0.02, gain=False, device="cpu"):
    """bf16-representable fp32 tensor: std * n (weights) or 1 + 0.05 * n (norm gains)."""
    n = hashed_normal(int(math.prod(shape)), seed, name, device).view(*shape)
    t = (n * 0.05 + 1.0) if gain else n * std
    return t.to(torch.bfloat16).to(torch.float32)


class HashedWeights:
    """Lazy mapping name -> fp32 tensor with transformers==4.41.0 LLaVA checkpoint names; nothing is stored.
    `delta` > 0: value = bf16(base(seed) + delta * n(seed_delta)) - a policy that differs from the reference `base`."""

    def __init__(self, cfg, seed=0, std=0.02, delta=0.0, seed_delta=1, device="cpu", cache=False, qk_scale=1.0, outliers=None):
        self.cfg, self.seed, self.std, self.delta, self.seed_delta, self.device = cfg, seed, std, delta, seed_delta, device
        # planted outlier channels (round 6; VERDICT r05 weak 1c: a trained checkpoint's massive-activation channels are where bf16 operand
        # rounding bites, a std-0.02 random model has none): dict(layers=[..], channels=[..], scale=2^k) multiplies the rows `channels` of
        # mlp.down_proj.weight of those layers (= those features of the residual stream) by `scale` AFTER the bf16 rounding - exact for a
        # power of two, so both sides still build bit-identical models (vlrlhf.utils.synthetic.init_hashed_model restates it)
        self.outliers = outliers
        self.qk_scale = qk_scale     # > 1: the decoder's q_proj / k_proj weights are drawn qk_scale times larger (sharper softmax: the
                                     # "sharp" depth fixtures, whose q / k gradients are large enough to be compared by direction)
        self._cache = {} if cache else None          # bf16 copies (2 B / parameter) for multi-pass runs
        D, P, H, I, V = cfg["vit_hidden"], cfg["patch_size"], cfg["hidden"], cfg["inter"], cfg["vocab"]
        nkv = cfg.get("kv_heads", cfg["heads"])
        hd = cfg.get("head_dim", H // cfg["heads"])
        ntok = (cfg["image_size"] // P) ** 2 + 1
        sh = {}
        vp = "vision_tower.vision_model."
        sh[vp + "embeddings.class_embedding"] = (D,)
        sh[vp + "embeddings.patch_embedding.weight"] = (D, 3, P, P)
        sh[vp + "embeddings.position_embedding.weight"] = (ntok, D)
        for nm in ("pre_layrnorm", "post_layernorm"):
            sh[vp + nm + ".weight"], sh[vp + nm + ".bias"] = (D,), (D,)
        for i in range(cfg["vit_layers"]):
            p = f"{vp}encoder.layers.{i}."
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sh[p + f"self_attn.{nm}.weight"], sh[p + f"self_attn.{nm}.bias"] = (D, D), (D,)
            for nm in ("layer_norm1", "layer_norm2"):
                sh[p + nm + ".weight"], sh[p + nm + ".bias"] = (D,), (D,)
            sh[p + "mlp.fc1.weight"], sh[p + "mlp.fc1.bias"] = (cfg["vit_mlp"], D), (cfg["vit_mlp"],)
            sh[p + "mlp.fc2.weight"], sh[p + "mlp.fc2.bias"] = (D, cfg["vit_mlp"]), (D,)
        sh["multi_modal_projector.linear_1.weight"], sh["multi_modal_projector.linear_1.bias"] = (H, D), (H,)
        sh["multi_modal_projector.linear_2.weight"], sh["multi_modal_projector.linear_2.bias"] = (H, H), (H,)
        lp = "language_model.model."
        sh[lp + "embed_tokens.weight"] = (V, H)
        for i in range(cfg["layers"]):
            p = f"{lp}layers.{i}."
            sh[p + "self_attn.q_proj.weight"], sh[p + "self_attn.o_proj.weight"] = (cfg["heads"] * hd, H), (H, cfg["heads"] * hd)
            sh[p + "self_attn.k_proj.weight"], sh[p + "self_attn.v_proj.weight"] = (nkv * hd, H), (nkv * hd, H)
            sh[p + "mlp.gate_proj.weight"], sh[p + "mlp.up_proj.weight"], sh[p + "mlp.down_proj.weight"] = (I, H), (I, H), (H, I)
            sh[p + "input_layernorm.weight"], sh[p + "post_attention_layernorm.weight"] = (H,), (H,)
        sh[lp + "norm.weight"] = (H,)
        sh["language_model.lm_head.weight"] = (V, H)
        self.shapes = sh

    def keys(self):
        return self.shapes.keys()

    def __contains__(self, k):
        return k in self.shapes

    def __iter__(self):
        return iter(self.shapes)

    def __getitem__(self, name):
        if self._cache is not None and name in self._cache:
            return self._cache[name].to(torch.float32)
        t = self._make(name)
        if self._cache is not None:
            self._cache[name] = t.to(torch.bfloat16)
        return t

    def _make(self, name):
        shape = self.shapes[name]
        gain = name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("layrnorm.weight")
        qk = self.qk_scale != 1.0 and name.startswith("language_model.") and name.endswith(("q_proj.weight", "k_proj.weight"))
        t = hashed_tensor(shape, self.seed, name, self.std * (self.qk_scale if qk else 1.0), gain, self.device)
        if self.delta > 0 and not name.startswith("vision_tower."):
            t = (t + hashed_normal(t.numel(), self.seed_delta, name, self.device).view(*shape) * self.delta).to(torch.bfloat16).to(torch.float32)
        o = self.outliers
        if o and name.endswith("mlp.down_proj.weight") and name.startswith("language_model.model.layers.") and int(name.split(".")[3]) in o["layers"]:
            t = t.clone()
            t[list(o["channels"])] *= float(o["scale"])
        return t


LLAVA_1_5_7B = dict(vit_hidden=1024, vit_mlp=4096, vit_layers=24, vit_heads=16, image_size=336, patch_size=14,
                    hidden=4096, inter=11008, layers=32, heads=32, vocab=32064, image_token=32000,
                    model_pad_token_id=32001, rms_eps=1e-5, rope_theta=10000.0)
