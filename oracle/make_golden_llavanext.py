#!/usr/bin/env python3
"""Golden vectors for the LLaVA-Next (anyres + Mistral GQA) DPO path.  TEST INFRASTRUCTURE - build container only.

    python oracle/make_golden_llavanext.py      ->  tests/golden/llavanext_small.npz

Composite oracle ("LLaVA-Next (C4)"): the REFERENCE's own
`LlavaNextForRL._merge_input_ids_with_image_features` (/root/reference/src/vlrlhf/models/LlavaNext/__init__.py:38-171),
`get_batch_logps`, `dpo_loss` and collator, imported under the stubs of oracle/make_golden.py, composed exactly as the
reference's forward (:205-265) composes them with the installed HuggingFace parts: `image_size_to_num_patches`,
CLIPVisionModel (`hidden_states[-2]`, CLS dropped), the LLaVA projector, `pack_image_features` (anyres grid reshape,
unpad, `image_newline` column) and MistralForCausalLM (grouped-query attention, eager).  Kernel-compatible toy widths
(decoder head_dim 128 with 2 query heads sharing 1 K/V head, ViT head_dim 64) so the HIP path is compared with this file
directly.  Version-drift caveat: the reference pins transformers 4.41.0, the container has 5.x - `unpad_image` gained a
`round(., 7)` since; the two differ only on exact float ties, which the sizes used here avoid.
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference)

from vlrlhf.models.LlavaNext import LlavaNextForRL  # noqa: E402  (the reference's wrapper)
from transformers import CLIPVisionConfig, CLIPVisionModel, MistralConfig, MistralForCausalLM  # noqa: E402
from transformers.models.llava_next import modeling_llava_next as HFN  # noqa: E402

CFG = dict(
    vit_hidden=128, vit_mlp=256, vit_layers=3, vit_heads=2, image_size=28, patch_size=14,
    hidden=256, inter=256, layers=2, heads=2, kv_heads=1, vocab=192, image_token=180, rope_theta=1000000.0,
    image_grid_pinpoints=[[28, 56], [56, 28], [56, 56], [84, 28], [28, 84]],
    pairs=2, prompt_len=(8, 12), resp_len=(6, 20), beta=0.1, w_scale=3.0, perturb=0.05,
    image_sizes=[[40, 75], [63, 30]],      # (height, width) of the two original images -> 1x2 and 2x1 tile grids, both unpadded
    padding_side="left",
)


def build(cfg, seed):
    torch.manual_seed(seed)
    vcfg = CLIPVisionConfig(hidden_size=cfg["vit_hidden"], intermediate_size=cfg["vit_mlp"], num_hidden_layers=cfg["vit_layers"],
                            num_attention_heads=cfg["vit_heads"], image_size=cfg["image_size"], patch_size=cfg["patch_size"],
                            hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=cfg["vit_hidden"])
    vcfg._attn_implementation = "eager"
    tcfg = MistralConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], intermediate_size=cfg["inter"],
                         num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], num_key_value_heads=cfg["kv_heads"],
                         rms_norm_eps=1e-5, rope_theta=cfg["rope_theta"], max_position_embeddings=4096, pad_token_id=None,
                         tie_word_embeddings=False, sliding_window=None, head_dim=cfg["hidden"] // cfg["heads"])
    tcfg._attn_implementation = "eager"
    lcfg = types.SimpleNamespace(vision_config=vcfg, text_config=tcfg, image_token_index=cfg["image_token"],
                                 projector_hidden_act="gelu", vision_feature_layer=-2, vision_feature_select_strategy="default",
                                 image_grid_pinpoints=cfg["image_grid_pinpoints"], ignore_index=-100, multimodal_projector_bias=True)
    vit = CLIPVisionModel(vcfg).float().eval()
    proj = HFN.LlavaNextMultiModalProjector(lcfg).float().eval()
    llm = MistralForCausalLM(tcfg).float().eval()
    newline = torch.nn.Parameter(torch.randn(cfg["hidden"]) * 0.3)
    with torch.no_grad():
        for p in list(llm.parameters()) + list(proj.parameters()) + list(vit.parameters()):
            if p.dim() >= 2:
                p.mul_(cfg["w_scale"])
        for n, p in list(vit.named_parameters()) + list(llm.named_parameters()):
            if "norm" in n and p.dim() == 1 and "weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            if "bias" in n:
                p.add_(0.02 * torch.randn_like(p))
    round_bf16(vit, proj, llm, newline)
    return vit, proj, llm, newline, lcfg


def round_bf16(*mods):
    """every weight is made bf16-representable BEFORE the golden values are computed, so the fixture stores 2 bytes per weight
    and the HIP path (bf16 storage) starts from exactly the same numbers"""
    with torch.no_grad():
        for m in mods:
            for p in ([m] if isinstance(m, torch.nn.Parameter) else m.parameters()):
                p.copy_(p.to(torch.bfloat16).to(torch.float32))


def bits(t):
    return t.detach().to(torch.bfloat16).view(torch.int16).cpu().numpy().copy()


def state_dict(vit, proj, llm, newline):
    sd = G.state_dict_441(vit, proj, llm)
    sd["image_newline"] = newline.detach().clone()
    return sd


def make_batch(cfg, seed):
    """rows + collated batch (reference collator) + anyres pixel_values [B, max_patches, 3, s, s] and image_sizes [B, 2]"""
    rows, batch = G.make_batch(dict(cfg), seed)
    g = np.random.default_rng(seed + 5)
    s = cfg["image_size"]
    sizes = torch.tensor(cfg["image_sizes"], dtype=torch.long)
    npatch = [HFN.image_size_to_num_patches(sz, cfg["image_grid_pinpoints"], s) for sz in sizes]
    pv = torch.zeros(len(npatch), max(npatch), 3, s, s)
    for i, n in enumerate(npatch):
        pv[i, :n] = torch.from_numpy(g.standard_normal((n, 3, s, s)).astype(np.float32))
    batch["img_input_dict"] = dict(pixel_values=pv, image_sizes=sizes)
    return rows, batch, npatch


def forward(vit, proj, llm, newline, lcfg, cfg, cb, keep=False):
    """the reference's LlavaNextForRL.forward on the training path (LlavaNext/__init__.py:205-265, 306-316)"""
    ids = cb["concatenated_input_ids"]
    for_emb = ids.clone()
    for_emb[ids == lcfg.image_token_index] = 0
    embeds = llm.get_input_embeddings()(for_emb)
    pv5 = cb["concatenated_img_input_dict"]["pixel_values"]
    image_sizes = cb["concatenated_img_input_dict"]["image_sizes"]
    npatch = [HFN.image_size_to_num_patches(image_size=sz, grid_pinpoints=lcfg.image_grid_pinpoints, patch_size=lcfg.vision_config.image_size)
              for sz in image_sizes]
    pv = torch.cat([x[:n] for x, n in zip(pv5, npatch)], dim=0)
    feat = vit(pv, output_hidden_states=True).hidden_states[lcfg.vision_feature_layer][:, 1:]
    img = proj(feat)
    img = torch.split(img, npatch, dim=0)
    fake_model = types.SimpleNamespace(config=lcfg)
    packed, feature_lens = HFN.LlavaNextModel.pack_image_features(fake_model, img, image_sizes, "default", image_newline=newline)
    packed = torch.cat(packed, dim=0)
    fake_self = types.SimpleNamespace(config=lcfg, padding_side=cfg["padding_side"])
    merged, mask, pos, labels, img_map = LlavaNextForRL._merge_input_ids_with_image_features(
        fake_self, packed, feature_lens, embeds, ids, cb["concatenated_attention_mask"], None, labels=cb["concatenated_labels"])
    out = llm(inputs_embeds=merged, attention_mask=mask, position_ids=pos, use_cache=False)
    inter = dict(vit_feat=feat, proj=torch.cat(img, 0), packed=packed, feature_lens=feature_lens, merged=merged, mask=mask, pos=pos,
                 labels=labels, img_map=img_map, npatch=npatch) if keep else None
    return out.logits.float(), labels, inter


def main():
    cfg = CFG
    vit, proj, llm, newline, lcfg = build(cfg, 4242)
    rows, batch, npatch = make_batch(cfg, 4243)
    cb = G.concatenated_inputs(batch)
    B = cfg["pairs"]
    out = {"config_json": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)}
    ref_sd = state_dict(vit, proj, llm, newline)
    with torch.no_grad():
        ref_logits, ref_labels, _ = forward(vit, proj, llm, newline, lcfg, cfg, cb)
        ref_logps = G.VLDPOTrainer.get_batch_logps(ref_logits, ref_labels, average_log_prob=False)
        ref_logps_ddpo = G.VLDPOTrainer.get_batch_logps(ref_logits, ref_labels, mask_shared_tokens=True)
    G.perturb(llm, proj, cfg["perturb"])
    with torch.no_grad():
        newline.add_(0.05 * torch.sin(torch.arange(newline.numel(), dtype=torch.float32)))
    round_bf16(proj, llm, newline)
    pol_sd = state_dict(vit, proj, llm, newline)
    for p in vit.parameters():
        p.requires_grad_(False)
    logits, labels, inter = forward(vit, proj, llm, newline, lcfg, cfg, cb, keep=True)
    logps = G.VLDPOTrainer.get_batch_logps(logits, labels, average_log_prob=False)
    logps_ddpo = G.VLDPOTrainer.get_batch_logps(logits, labels, mask_shared_tokens=True)
    for lt in ("sigmoid", "ddpo", "ipo"):
        pl, rl = (logps_ddpo, ref_logps_ddpo) if lt == "ddpo" else (logps, ref_logps)
        losses, cr, rr_ = G.dpo_loss_ref(lt, cfg["beta"], pl[:B], pl[B:], rl[:B], rl[B:])
        out[f"loss_{lt}"], out[f"chosen_rewards_{lt}"], out[f"rejected_rewards_{lt}"] = map(G.to_np, (losses, cr, rr_))
    # DDPO is the configuration BASELINE.json configs[3] names: its training loss and gradients
    losses, _, _ = G.dpo_loss_ref("ddpo", cfg["beta"], logps_ddpo[:B], logps_ddpo[B:], ref_logps_ddpo[:B], ref_logps_ddpo[B:])
    loss = losses.mean()
    loss.backward()
    named = [("language_model." + n, p) for n, p in llm.named_parameters()] + \
            [("multi_modal_projector." + n, p) for n, p in proj.named_parameters()] + [("image_newline", newline)]
    gsq = sum(float((p.grad.double() ** 2).sum()) for _, p in named)
    out["grad_norm"] = np.array(gsq ** 0.5)
    keep = ("image_newline", "multi_modal_projector", "embed_tokens", "lm_head", "layers.0.self_attn", "layers.1.mlp.down_proj", "norm")
    for n, p in named:          # a subset keeps the fixture small: every kind of tensor, both layers
        if any(k in n for k in keep):
            out["grad." + n] = G.to_np(p.grad)
    for k, v in pol_sd.items():             # "w16." = bf16 bit patterns (tests/golden_util.load_case turns them back into fp32)
        out["w16." + k] = bits(v)
    for k, v in ref_sd.items():
        if not k.startswith("vision_tower."):
            out["ref_w16." + k] = bits(v)
    for k in ("chosen_input_ids", "chosen_attention_mask", "chosen_labels", "rejected_input_ids", "rejected_attention_mask",
              "rejected_labels", "prompt_input_ids", "prompt_attention_mask"):
        out["batch." + k] = G.to_np(batch[k])
    out["batch.pixel_values"] = G.to_np(batch["img_input_dict"]["pixel_values"])
    out["batch.image_sizes"] = G.to_np(batch["img_input_dict"]["image_sizes"])
    out["rows_json"] = np.frombuffer(json.dumps(rows).encode(), dtype=np.uint8)
    for k in ("concatenated_input_ids", "concatenated_attention_mask", "concatenated_labels"):
        out["cat." + k] = G.to_np(cb[k])
    out["num_patches"] = np.array(inter["npatch"])
    out["feature_lens"] = G.to_np(inter["feature_lens"])
    F = int(inter["feature_lens"][:B].sum())
    out["vit_feat"] = G.to_np(inter["vit_feat"][: sum(npatch)])
    out["projected"] = G.to_np(inter["proj"][: sum(npatch)])
    out["packed_features"] = G.to_np(inter["packed"][:F])
    out["merged_embeds"], out["merged_mask"], out["merged_pos"] = G.to_np(inter["merged"]), G.to_np(inter["mask"]), G.to_np(inter["pos"])
    out["merged_labels"], out["image_position_map"] = G.to_np(inter["labels"]), G.to_np(inter["img_map"])
    out["logits"] = G.to_np(logits)
    out["policy_logps"], out["policy_logps_ddpo"] = G.to_np(logps), G.to_np(logps_ddpo)
    out["ref_logps"], out["ref_logps_ddpo"] = G.to_np(ref_logps), G.to_np(ref_logps_ddpo)
    out["loss_mean_ddpo"] = np.array(float(loss))
    # merge-only known answers: right padding, left padding, a text-only row mixed in is NOT supported by the reference's count check
    ka = {}
    fake_self = types.SimpleNamespace(config=lcfg, padding_side="left")
    g = torch.Generator().manual_seed(9)
    for tag, ids, am in (
            ("right", [[1, 180, 5, 6, 7, 0, 0], [1, 2, 180, 6, 7, 8, 9]], [[1, 1, 1, 1, 1, 0, 0], [1] * 7]),
            ("left", [[0, 0, 1, 180, 5, 6, 7], [1, 2, 180, 6, 7, 8, 9]], [[0, 0, 1, 1, 1, 1, 1], [1] * 7]),
            ("nopad_two_images", [[1, 180, 5, 180, 7, 3, 4]], [[1] * 7])):
        ids, am = torch.tensor(ids), torch.tensor(am)
        n_img = int((ids == 180).sum())
        fl = torch.tensor([3, 5, 2][:n_img])
        feats = torch.randn(int(fl.sum()), 8, generator=g)
        emb = torch.randn(ids.shape[0], ids.shape[1], 8, generator=g)
        lab = torch.where(am == 1, ids, torch.full_like(ids, -100))
        fe, fm, pos, fl_, imap = LlavaNextForRL._merge_input_ids_with_image_features(fake_self, feats, fl, emb, ids, am, None, labels=lab)
        for k, v in (("ids", ids), ("am", am), ("fl", fl), ("feats", feats), ("emb", emb), ("lab", lab), ("out_emb", fe), ("out_mask", fm),
                     ("out_pos", pos), ("out_labels", fl_), ("out_map", imap)):
            ka[f"mg.{tag}.{k}"] = G.to_np(v)
    out.update(ka)
    path = os.path.join(G.OUT_DIR, "llavanext_small.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] llavanext_small: loss(ddpo)={float(loss):.6f} logps={G.to_np(logps)} feature_lens={inter['feature_lens'].tolist()} "
          f"patches={npatch} S={inter['merged'].shape[1]} gradnorm={gsq ** 0.5:.4f} -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    torch.set_num_threads(4)
    main()
