"""CPU restatement (fp32, torch) of the InternLM-XComposer2 deltas of the DPO hot path - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Restates what the reference reaches
through InternLMXC2ForRL.forward (/root/reference/src/vlrlhf/models/InternLMXC2/__init__.py:107-236):
  merge          the LLaVA-style expansion of <ImageHere> (:33-105) = llava_dpo_oracle.merge_input_ids_with_image_features; the image token is
                 embedded as the pad token first (:129-131); `im_mask` = the image positions
  vision         CLIP ViT, LAST hidden state without CLS (build_mlp.py:55-108: select_layer -1, "patch") -> mlp2x_gelu projector (:14-27)
  decoder        InternLM2 (modeling_internlm2.py): fused grouped-query `wqkv` whose rows are laid out per K/V head as
                 [q_0 .. q_{g-1} | k | v] (:318-330), rotate-half RoPE by the index in the merged sequence (:188-203 ignore position_ids),
                 `w2(silu(w1(x)) * w3(x))` (:221-224)
  PLoRA          every decoder linear is `W x + Plora_B(Plora_A(dropout(x)))` ON THE IMAGE ROWS ONLY (build_mlp.py:158-203; r 256,
                 alpha 256 -> scaling 1, dropout 0.05 in training mode).  These are base-model weights: trained in a full fine-tune, active
                 in the reference pass.
  LoRA           peft adapters on top of the five PLoRA linears (default_lora_target :244-245; scripts/dpo_internlmxc2vl7b.sh: r 64, alpha 64)
Pinned by tests/golden/internlmxc2_small.npz (oracle/make_golden_internlm.py: the reference's own classes, eval mode).
Dropout convention of the product (the reference draws from torch's RNG): the image rows of the whole batch, in (sequence, position)
order, form a compact [R, in] matrix; target t of layer l masks it with dropout_mask(seed + 8 l + t), t = 0 wqkv, 3 wo, 4 w1, 5 w3, 6 w2."""
import math

import torch
import torch.nn.functional as F

from . import llava_dpo_oracle as O

TARGETS = {"attention.wqkv": 0, "attention.wo": 3, "feed_forward.w1": 4, "feed_forward.w3": 5, "feed_forward.w2": 6}


def qkv_row_order(heads, kv_heads, hd):
    """row permutation taking InternLM2's wqkv output layout ([kv head][q_0..q_{g-1} | k | v][hd]) to q | k | v blocks: out[i] = in[perm[i]]"""
    g = heads // kv_heads
    idx = torch.arange((heads + 2 * kv_heads) * hd).view(kv_heads, g + 2, hd)
    return torch.cat([idx[:, :g].reshape(-1), idx[:, g].reshape(-1), idx[:, g + 1].reshape(-1)])


def plora_delta(h, im_mask, W, name, r, plora):
    """Plora_B(Plora_A(dropout(x[im_mask]))) * scaling scattered back to the image rows (zeros elsewhere)"""
    A, B = r(W[name + ".Plora_A.weight"]), r(W[name + ".Plora_B.weight"])
    out = torch.zeros(*h.shape[:-1], B.shape[0], dtype=h.dtype)
    if not bool(im_mask.any()):
        return out
    part = h[im_mask]
    p = float(plora.get("p", 0.0) or 0.0) if plora else 0.0
    scale = plora.get("scale", 1.0) if plora else 1.0
    if p > 0 and plora.get("seed") is not None:
        seed = plora["seed"] + 8 * plora["layer"] + TARGETS[name.split("layers.")[1].split(".", 1)[1]]
        if plora.get("index", "full") == "full":
            # the C layer passes (vlr_decoder_layer_*_lora_ex): mask indexed over the FULL [B*S][in] input, applied exactly while the operand
            # is staged; 1 / (1 - p) rides in the scale
            m = O.dropout_mask(seed, h.numel(), p).view(h.shape).to(h.dtype)
            out[im_mask] = r(scale / (1.0 - p) * ((h * m)[im_mask] @ A.t())) @ B.t()
            return out
        # the Python-composed layer (peft LoRA stacked on PLoRA): vlr_dropout over the COMPACT [R][in] matrix of the image rows
        m = O.dropout_mask(seed, part.numel(), p).view(part.shape).to(part.dtype)
        part = r(part * m * (1.0 / (1.0 - p)))
    out[im_mask] = r(scale * (part @ A.t())) @ B.t()
    return out


def lora_names(layer, target, prefix="base_model.model.model.layers."):
    base = f"{prefix}{layer}.{target}"
    return base + ".lora_A.weight", base + ".lora_B.weight"


def lora_delta(h, lora, layer, target, r):
    na, nb = lora_names(layer, target)
    A, B = r(lora["W"][na]), r(lora["W"][nb])
    p = float(lora.get("dropout", 0.0) or 0.0)
    if p > 0 and lora.get("seed") is not None:
        m = O.dropout_mask(lora["seed"] + 8 * layer + TARGETS[target], h.numel(), p).view(h.shape).to(h.dtype)
        h = r(h * m * (1.0 / (1.0 - p)))
    return r(lora["scale"] * (h @ A.t())) @ B.t()


def random_lora(cfg, r, alpha, seed=0, b_std=0.0, dropout=0.0):
    g = torch.Generator().manual_seed(seed)
    H, I = cfg["hidden"], cfg["inter"]
    hd = H // cfg["heads"]
    N = (cfg["heads"] + 2 * cfg.get("kv_heads", cfg["heads"])) * hd
    dims = {"attention.wqkv": (H, N), "attention.wo": (H, H), "feed_forward.w1": (H, I), "feed_forward.w3": (H, I), "feed_forward.w2": (I, H)}
    W = {}
    for l in range(cfg["layers"]):
        for t, (din, dout) in dims.items():
            na, nb = lora_names(l, t)
            W[na] = (torch.rand(r, din, generator=g) * 2 - 1) / math.sqrt(din)
            W[nb] = torch.randn(dout, r, generator=g) * b_std
    return dict(W=W, scale=alpha / r, dropout=dropout, seed=None, r=r)


def internlm_hidden(x, attention_mask, position_ids, im_mask, W, cfg, emulate_bf16=False, lora=None, plora=None, prefix="model."):
    """InternLM2Model layers + norm on merged embeddings (modeling_internlm2.py:497-560, 760-860)"""
    r = lambda t, tag=None: O._rt(t, emulate_bf16, tag)   # noqa: E731
    rw = lambda t: O._rt(t, emulate_bf16, "w")             # noqa: E731
    B, S, H = x.shape
    nh, nkv = cfg["heads"], cfg.get("kv_heads", cfg["heads"])
    hd = H // nh
    eps = cfg.get("rms_eps", 1e-5)
    # the vendored apply_rotary_pos_emb (modeling_internlm2.py:188-203) takes cos / sin rows 0 .. S-1 and never indexes them with
    # position_ids: the rotary position is the index in the MERGED sequence (as Qwen), whatever the padding
    cos, sin = O.rope_tables(torch.arange(S)[None].expand(B, S), hd, cfg.get("rope_theta", 1000000.0))
    bias = O.causal_padding_bias(attention_mask)
    perm = qkv_row_order(nh, nkv, hd)
    pl = dict(plora or {}, scale=cfg.get("plora_alpha", 256) / cfg.get("plora_r", 256))

    def lin(h, l, target, out_perm=None):
        name = f"{prefix}layers.{l}.{target}"
        y = h @ rw(W[name + ".weight"]).t() + plora_delta(h, im_mask, W, name, rw, dict(pl, layer=l))
        if lora is not None:
            y = y + lora_delta(h, lora, l, target, rw)
        return y if out_perm is None else y[..., out_perm]

    x = r(x, "x0")
    for l in range(cfg["layers"]):
        p = f"{prefix}layers.{l}."
        h = r(O.rms_norm(x, W[p + "attention_norm.weight"], eps), "xn")
        qkv = r(lin(h, l, "attention.wqkv", perm), "qkv")                 # -> q | k | v blocks
        q = qkv[..., : nh * hd].reshape(B, S, nh, hd).transpose(1, 2)
        k = qkv[..., nh * hd: (nh + nkv) * hd].reshape(B, S, nkv, hd).transpose(1, 2)
        v = qkv[..., (nh + nkv) * hd:].reshape(B, S, nkv, hd).transpose(1, 2)
        q, k, v = r(O.apply_rope(q, cos, sin), "rope"), r(O.apply_rope(k, cos, sin), "rope"), r(v, "v")
        if nkv != nh:
            k, v = (t.repeat_interleave(nh // nkv, dim=1) for t in (k, v))
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + bias, dim=-1)
        ctx = r((r(att, "p") @ v).transpose(1, 2).reshape(B, S, H), "attn")
        x = r(x + lin(ctx, l, "attention.wo"), "resid")
        h = r(O.rms_norm(x, W[p + "ffn_norm.weight"], eps), "xn")
        act = r(F.silu(r(lin(h, l, "feed_forward.w1"), "gu")) * r(lin(h, l, "feed_forward.w3"), "gu"), "act")
        x = r(x + lin(act, l, "feed_forward.w2"), "resid")
    return r(O.rms_norm(x, W[prefix + "norm.weight"], eps), "hidden")


def internlm_forward(W, cfg, input_ids, attention_mask, labels, pixel_values, emulate_bf16=False, lora=None, plora=None,
                     dedupe_images=True, return_hidden=False):
    """InternLMXC2ForRL.forward on the training path -> (logits fp32, merged labels, aux)"""
    pad = cfg["model_pad_token_id"]
    fake = torch.where(input_ids == cfg["image_token"], torch.full_like(input_ids, pad), input_ids)          # :129-131
    emb = O._rt(W["model.tok_embeddings.weight"][fake], emulate_bf16, "w")
    vcfg = dict(cfg, vit_feature_layer=cfg.get("vit_feature_layer", -1))
    n = pixel_values.shape[0]
    vp = "vit.vision_tower.vision_model."
    Wp = {"p.linear_1.weight": W["vision_proj.0.weight"], "p.linear_1.bias": W["vision_proj.0.bias"],
          "p.linear_2.weight": W["vision_proj.2.weight"], "p.linear_2.bias": W["vision_proj.2.bias"]}
    if dedupe_images and n % 2 == 0 and torch.equal(pixel_values[: n // 2], pixel_values[n // 2:]):
        feat = O.clip_vit_features(pixel_values[: n // 2], W, vcfg, emulate_bf16, prefix=vp)
        img = O.projector(feat, Wp, emulate_bf16, prefix="p.")
        img = torch.cat([img, img], 0)
    else:
        feat = O.clip_vit_features(pixel_values, W, vcfg, emulate_bf16, prefix=vp)
        img = O.projector(feat, Wp, emulate_bf16, prefix="p.")
    merged, mask, mlabels, pos, img_map = O.merge_input_ids_with_image_features(img, emb, input_ids, attention_mask, labels, cfg["image_token"], pad)
    hidden = internlm_hidden(merged, mask, pos, img_map, W, cfg, emulate_bf16, lora=lora, plora=plora)
    aux = dict(image_features=img, merged=merged, mask=mask, pos=pos, img_map=img_map, hidden=hidden)
    if return_hidden:
        return hidden, mlabels, aux
    return O.lm_logits(hidden, W, emulate_bf16, key="output.weight"), mlabels, aux


def concatenated_forward(W, cfg, batch, loss_type="sigmoid", emulate_bf16=False, lora=None, plora=None):
    cb = O.concatenated_inputs(batch, padding_value=cfg["model_pad_token_id"])
    logits, labels, _ = internlm_forward(W, cfg, cb["concatenated_input_ids"], cb["concatenated_attention_mask"], cb["concatenated_labels"],
                                         cb["concatenated_img_input_dict"]["pixel_values"], emulate_bf16, lora=lora, plora=plora)
    n = batch["chosen_labels"].shape[0]
    lp = O.get_batch_logps(logits, labels, mask_shared_tokens=(loss_type == "ddpo"))
    return lp[:n], lp[n:], logits[:n], logits[n:]


def compute_loss(W_policy, W_ref, cfg, batch, beta=0.1, loss_type="sigmoid", emulate_bf16=False, lora=None, plora=None, ref_plora=None):
    pc, pr, _, _ = concatenated_forward(W_policy, cfg, batch, loss_type, emulate_bf16, lora=lora, plora=plora)
    with torch.no_grad():
        rc, rr, _, _ = concatenated_forward(W_ref, cfg, batch, loss_type, emulate_bf16, plora=ref_plora)
    losses, cr, rrw = O.dpo_loss(pc, pr, rc, rr, beta, 0.0, loss_type, False)
    return losses.mean(), dict(pc=pc, pr=pr, rc=rc, rr=rr, margins=(cr - rrw).mean())
