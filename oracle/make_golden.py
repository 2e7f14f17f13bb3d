#!/usr/bin/env python3
"""Golden-vector generator for the LLaVA DPO hot path.  TEST INFRASTRUCTURE ONLY.

Runs ONLY in the build container (needs /root/reference and the installed
HuggingFace `transformers`); the vectors it writes under tests/golden/ are data
(inputs + expected outputs) and travel to the GPU box, the reference does not.

What is "the reference" here: the reference repo cannot
be imported as-is (missing trl/peft/deepspeed/...), so this script

  * stubs the absent third-party modules (SURVEY.md Appendix C recipe),
  * imports the reference's OWN hot-path functions
      - VLDPOTrainer.get_batch_logps      src/vlrlhf/base/trainer.py:148-188
      - VLDPOTrainer.dpo_loss             src/vlrlhf/base/trainer.py:244-301
      - VLDPODataCollatorWithPadding      src/vlrlhf/base/collator.py:26-68
      - LlavaForRL._merge_input_ids_with_image_features
                                          src/vlrlhf/models/Llava/__init__.py:36-109
      - get_diff_ids                      src/vlrlhf/utils/diff_lib.py:173-180
  * composes them with the installed HF CLIPVisionModel / LlavaMultiModalProjector /
    LlamaForCausalLM (eager attention, fp32) exactly as LlavaForRL.forward does
    (src/vlrlhf/models/Llava/__init__.py:174-243),
  * and records inputs, weights and outputs for seeded tiny configurations.

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz)
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch
import transformers  # noqa: F401
from transformers import PreTrainedModel, AutoModelForCausalLM, Trainer, TrainingArguments, PreTrainedTokenizerBase  # noqa: F401
from transformers.trainer_callback import TrainerCallback  # noqa: F401
from transformers.trainer_utils import EvalPrediction, EvalLoopOutput  # noqa: F401
from transformers.tokenization_utils_base import BatchEncoding  # noqa: F401
import accelerate.utils  # noqa: F401
import datasets  # noqa: F401

REF_SRC = "/root/reference/src"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Base:
        def __init__(self, *a, **k):
            pass

    class _Logger:
        def __getattr__(self, n):
            return lambda *a, **k: None

    mod("loguru", logger=_Logger())
    mod("wandb", Image=_Base)
    mod("trl", DPOTrainer=_Base, PPOTrainer=_Base, PPOConfig=_Base, SFTTrainer=_Base, RewardTrainer=_Base,
        RewardConfig=_Base, AutoModelForCausalLMWithValueHead=_Base)
    mod("trl.trainer")
    mod("trl.trainer.reward_config", RewardConfig=_Base)
    mod("peft", PeftConfig=_Base, LoraConfig=_Base, PeftModel=_Base,
        prepare_model_for_kbit_training=lambda *a, **k: None, get_peft_model=lambda *a, **k: None)
    mod("deepspeed", zero=types.SimpleNamespace(GatheredParameters=None))
    mod("deepspeed.runtime")
    mod("deepspeed.runtime.zero")
    mod("deepspeed.runtime.zero.partition_parameters", ZeroParamStatus=types.SimpleNamespace(NOT_AVAILABLE=0))
    ds = mod("transformers.deepspeed", is_deepspeed_zero3_enabled=lambda: False)
    try:
        transformers.deepspeed = ds
    except Exception:
        pass
    transformers.__dict__["deepspeed"] = ds


_install_stubs()
sys.path.insert(0, REF_SRC)
from vlrlhf.base.trainer import VLDPOTrainer  # noqa: E402
from vlrlhf.base.collator import VLDPODataCollatorWithPadding  # noqa: E402
from vlrlhf.models.Llava import LlavaForRL  # noqa: E402
from vlrlhf.utils.diff_lib import get_diff_ids  # noqa: E402

from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM, LlavaConfig  # noqa: E402
from transformers.models.llava.modeling_llava import LlavaMultiModalProjector  # noqa: E402

IMAGE_TOKEN = 32000  # remapped per config below (tiny vocabularies)
LOSS_TYPES = ["sigmoid", "hinge", "ipo", "kto_pair", "ddpo"]


# ----------------------------------------------------------------------------------------------
def build_models(cfg, seed):
    torch.manual_seed(seed)
    vcfg = CLIPVisionConfig(
        hidden_size=cfg["vit_hidden"], intermediate_size=cfg["vit_mlp"], num_hidden_layers=cfg["vit_layers"],
        num_attention_heads=cfg["vit_heads"], image_size=cfg["image_size"], patch_size=cfg["patch_size"],
        hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=cfg["vit_hidden"],
    )
    vcfg._attn_implementation = "eager"
    tcfg = LlamaConfig(
        vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], intermediate_size=cfg["inter"],
        num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], num_key_value_heads=cfg["heads"],
        rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=4096, pad_token_id=None,
        tie_word_embeddings=False, attention_bias=False, mlp_bias=False,
    )
    tcfg._attn_implementation = "eager"
    lcfg = LlavaConfig(vision_config=vcfg, text_config=tcfg, image_token_index=cfg["image_token"],
                       projector_hidden_act="gelu", vision_feature_layer=-2,
                       vision_feature_select_strategy="default")
    vit = CLIPVisionModel(vcfg).float().eval()
    proj = LlavaMultiModalProjector(lcfg).float().eval()
    llm = LlamaForCausalLM(tcfg).float().eval()
    # HF init gives N(0, 0.02): make the model less degenerate so softmaxes/logits carry signal
    with torch.no_grad():
        for p in list(llm.parameters()) + list(proj.parameters()) + list(vit.parameters()):
            if p.dim() >= 2:
                p.mul_(cfg.get("w_scale", 3.0))
        for n, p in list(vit.named_parameters()) + list(llm.named_parameters()):
            if "norm" in n and p.dim() == 1 and "weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            if "bias" in n:
                p.add_(0.02 * torch.randn_like(p))
    return vit, proj, llm, lcfg


def state_dict_441(vit, proj, llm):
    """Name the weights as transformers==4.41.0's LlavaForConditionalGeneration checkpoint does
    (the version the reference pins: pyproject.toml:14)."""
    sd = {}
    for k, v in vit.state_dict().items():
        if k.endswith("position_ids"):
            continue
        # transformers 5.x dropped the inner `vision_model.` level; 4.41.0 checkpoints carry it
        sd["vision_tower." + (k if k.startswith("vision_model.") else "vision_model." + k)] = v
    for k, v in proj.state_dict().items():
        sd["multi_modal_projector." + k] = v
    for k, v in llm.state_dict().items():
        sd["language_model." + k] = v
    return {k: v.detach().clone() for k, v in sd.items()}


def make_batch(cfg, seed):
    """Ragged synthetic DPO rows -> reference collator -> batch (SURVEY 8d 'ragged variant')."""
    g = np.random.default_rng(seed)
    rows = []
    for b in range(cfg["pairs"]):
        lp = int(g.integers(cfg["prompt_len"][0], cfg["prompt_len"][1] + 1))
        lc = int(g.integers(cfg["resp_len"][0], cfg["resp_len"][1] + 1))
        lr = int(g.integers(cfg["resp_len"][0], cfg["resp_len"][1] + 1))
        body = g.integers(3, cfg["image_token"], size=lp - 2).tolist()
        img_at = int(g.integers(1, 4))
        prompt = [1] + body[:img_at] + [cfg["image_token"]] + body[img_at:]
        chosen_resp = g.integers(3, cfg["image_token"], size=lc).tolist() + [2]
        # the rejected answer shares a prefix / suffix with the chosen one so DDPO masks are non-trivial
        rej_resp = list(chosen_resp[: max(3, lc // 3)]) + g.integers(3, cfg["image_token"], size=lr).tolist()
        rej_resp += list(chosen_resp[-max(4, lc // 4):])
        row = dict(
            prompt_input_ids=prompt, prompt_attention_mask=[1] * len(prompt),
            chosen_input_ids=prompt + chosen_resp, chosen_attention_mask=[1] * (len(prompt) + len(chosen_resp)),
            chosen_labels=[-100] * len(prompt) + chosen_resp,
            rejected_input_ids=prompt + rej_resp, rejected_attention_mask=[1] * (len(prompt) + len(rej_resp)),
            rejected_labels=[-100] * len(prompt) + rej_resp,
            img_path=f"synthetic_{b}.jpg",
        )
        rows.append(row)
    coll = VLDPODataCollatorWithPadding(pad_token_id=0, label_pad_token_id=-100, is_encoder_decoder=False)
    batch = coll(rows)
    px = g.integers(0, 256, size=(cfg["pairs"], 3, cfg["image_size"], cfg["image_size"]), dtype=np.uint8)
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)[None, :, None, None]
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)[None, :, None, None]
    pv = ((px.astype(np.float32) / 255.0) - mean) / std
    batch["img_input_dict"] = dict(pixel_values=torch.from_numpy(pv))
    return rows, batch


def concatenated_inputs(batch, label_pad_token_id=-100, padding_value=0):
    """trl 0.8.1 DPOTrainer.concatenated_inputs (static, decoder-only branch) + the reference's image
    duplication (src/vlrlhf/base/trainer.py:135-145).  trl is absent here; its published algorithm: pad
    chosen_* / rejected_* to the common max length (labels -> label_pad, ids -> padding_value, mask -> 0) and
    concatenate on dim 0."""
    from vlrlhf.utils.common import pad_to_length  # the reference's own helper
    out = {}
    max_length = max(batch["chosen_input_ids"].shape[1], batch["rejected_input_ids"].shape[1])
    for side in ("chosen", "rejected"):
        for k in batch:
            if k.startswith(side) and isinstance(batch[k], torch.Tensor):
                if "labels" in k:
                    pad_value = label_pad_token_id
                elif k.endswith("_input_ids"):
                    pad_value = padding_value
                elif k.endswith("_attention_mask"):
                    pad_value = 0
                ck = k.replace(side, "concatenated")
                padded = pad_to_length(batch[k], max_length, pad_value=pad_value)
                out[ck] = padded if side == "chosen" else torch.cat((out[ck], padded), dim=0)
    out["concatenated_img_input_dict"] = {k: torch.cat([v, v], dim=0) for k, v in batch["img_input_dict"].items()}
    return out


def composite_forward(vit, proj, llm, lcfg, cb, want_intermediates=False):
    """LlavaForRL.forward restated with reference merge + installed HF parts
    (src/vlrlhf/models/Llava/__init__.py:174-243)."""
    ids = cb["concatenated_input_ids"]
    embeds = llm.get_input_embeddings()(ids)
    pv = cb["concatenated_img_input_dict"]["pixel_values"]
    vout = vit(pv, output_hidden_states=True)
    feat = vout.hidden_states[lcfg.vision_feature_layer][:, 1:]
    image_features = proj(feat)
    fake_self = types.SimpleNamespace(
        pad_token_id=lcfg.image_token_index + 1,  # model pad id (32001 in the real checkpoint) never in input_ids
        config=types.SimpleNamespace(image_token_index=lcfg.image_token_index, ignore_index=-100),
    )
    merged, mask, labels, pos, img_map = LlavaForRL._merge_input_ids_with_image_features(
        fake_self, image_features, embeds, ids, cb["concatenated_attention_mask"], cb["concatenated_labels"])
    out = llm(inputs_embeds=merged, attention_mask=mask, position_ids=pos, use_cache=False,
              output_hidden_states=want_intermediates)
    logits = out.logits.float()
    inter = dict(vit_feat=feat, image_features=image_features, merged=merged, mask=mask, labels=labels, pos=pos,
                 img_map=img_map)
    if want_intermediates:
        inter["hidden_last"] = out.hidden_states[-1]   # after final norm? (HF: last entry is post-norm)
        inter["hidden_l0"] = out.hidden_states[1]
    return logits, labels, inter


def dpo_loss_ref(loss_type, beta, pc, pr, rc, rr, label_smoothing=0.0, reference_free=False):
    fake = types.SimpleNamespace(beta=beta, label_smoothing=label_smoothing, loss_type=loss_type,
                                 reference_free=reference_free,
                                 accelerator=types.SimpleNamespace(device=torch.device("cpu")))
    return VLDPOTrainer.dpo_loss(fake, pc, pr, rc, rr)


def perturb(llm, proj, scale):
    """policy = ref + deterministic delta (no RNG: reproducible from the stored ref weights)."""
    with torch.no_grad():
        for i, (n, p) in enumerate(list(llm.named_parameters()) + list(proj.named_parameters())):
            idx = torch.arange(p.numel(), dtype=torch.float64)
            delta = (scale * torch.sin(idx * 0.37 + i)).to(torch.float32).reshape(p.shape)
            p.add_(delta * p.abs().mean())


def to_np(x):
    return x.detach().cpu().numpy().copy()


def gen_case(name, cfg, seed):
    vit, proj, llm, lcfg = build_models(cfg, seed)
    rows, batch = make_batch(cfg, seed + 1)
    cb = concatenated_inputs(batch)
    B = cfg["pairs"]
    out = {"config_json": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)}

    # ---- reference (frozen) model: weights as built
    ref_sd = state_dict_441(vit, proj, llm)
    with torch.no_grad():
        ref_logits, ref_labels, _ = composite_forward(vit, proj, llm, lcfg, cb)
        ref_logps = VLDPOTrainer.get_batch_logps(ref_logits, ref_labels, average_log_prob=False)
        ref_logps_ddpo = VLDPOTrainer.get_batch_logps(ref_logits, ref_labels, mask_shared_tokens=True)

    # ---- policy: perturbed copy, trainable LLM + projector, frozen ViT
    perturb(llm, proj, cfg.get("perturb", 0.05))
    pol_sd = state_dict_441(vit, proj, llm)
    for p in vit.parameters():
        p.requires_grad_(False)
    logits, labels, inter = composite_forward(vit, proj, llm, lcfg, cb, want_intermediates=True)
    logps = VLDPOTrainer.get_batch_logps(logits, labels, average_log_prob=False)
    logps_avg = VLDPOTrainer.get_batch_logps(logits, labels, average_log_prob=True)
    logps_ddpo = VLDPOTrainer.get_batch_logps(logits, labels, mask_shared_tokens=True)

    beta = cfg["beta"]
    for lt in LOSS_TYPES:
        pl, rl = (logps_ddpo, ref_logps_ddpo) if lt == "ddpo" else (logps, ref_logps)
        losses, cr, rr_ = dpo_loss_ref(lt, beta, pl[:B], pl[B:], rl[:B], rl[B:])
        out[f"loss_{lt}"] = to_np(losses)
        out[f"chosen_rewards_{lt}"] = to_np(cr)
        out[f"rejected_rewards_{lt}"] = to_np(rr_)
    ls, _, _ = dpo_loss_ref("sigmoid", beta, logps[:B], logps[B:], ref_logps[:B], ref_logps[B:], label_smoothing=0.1)
    out["loss_sigmoid_ls0p1"] = to_np(ls)
    lf, _, _ = dpo_loss_ref("sigmoid", beta, logps[:B], logps[B:], ref_logps[:B], ref_logps[B:], reference_free=True)
    out["loss_sigmoid_reffree"] = to_np(lf)

    # ---- backward of the training loss (sigmoid, mean over pairs; trl compute_loss -> losses.mean())
    losses, _, _ = dpo_loss_ref("sigmoid", beta, logps[:B], logps[B:], ref_logps[:B], ref_logps[B:])
    loss = losses.mean()
    loss.backward()
    named = [("language_model." + n, p) for n, p in llm.named_parameters()] + \
            [("multi_modal_projector." + n, p) for n, p in proj.named_parameters()]
    gsq = sum(float((p.grad.double() ** 2).sum()) for _, p in named)
    out["grad_norm"] = np.array(gsq ** 0.5)
    for n, p in named:
        out["grad." + n] = to_np(p.grad)

    # ---- one optimizer step, HF Trainer recipe: clip_grad_norm_(1.0) then torch AdamW with decay on
    #      non-norm non-bias params (scripts/dpo_llava.sh:35-41 betas/eps; transformers Trainer.get_decay_parameter_names)
    decay, no_decay = [], []
    for n, p in named:
        (no_decay if ("norm" in n or n.endswith(".bias")) else decay).append(p)
    hp = cfg["optim"]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": hp["weight_decay"]},
                             {"params": no_decay, "weight_decay": 0.0}],
                            lr=hp["lr"], betas=(hp["beta1"], hp["beta2"]), eps=hp["eps"])
    total = torch.nn.utils.clip_grad_norm_([p for _, p in named], hp["max_grad_norm"])
    out["clip_total_norm"] = to_np(total)
    opt.step()
    asq = 0.0
    for i, (n, p) in enumerate(named):
        asq += float((p.detach().double() ** 2).sum())
        if cfg.get("store_all_after_step", False) or i % 5 == 0:
            out["after_step." + n] = to_np(p)
    out["after_step_sqnorm"] = np.array(asq)

    # ---- DDPO index sets from the reference diff library (shifted, 0-filled labels; trainer.py:161-182)
    sh = labels[:, 1:].clone()
    sh[sh == -100] = 0
    for b in range(B):
        c_mod, r_mod = get_diff_ids(sh[b].tolist(), sh[B + b].tolist(), min_match_size=3)
        out[f"ddpo_chosen_ids_{b}"] = np.array(c_mod, dtype=np.int64)
        out[f"ddpo_rejected_ids_{b}"] = np.array(r_mod, dtype=np.int64)

    # ---- record
    for k, v in pol_sd.items():
        out["w." + k] = to_np(v)
    for k, v in ref_sd.items():
        if not k.startswith("vision_tower."):
            out["ref_w." + k] = to_np(v)
    for k in ("chosen_input_ids", "chosen_attention_mask", "chosen_labels", "rejected_input_ids",
              "rejected_attention_mask", "rejected_labels", "prompt_input_ids", "prompt_attention_mask"):
        out["batch." + k] = to_np(batch[k])
    out["batch.pixel_values"] = to_np(batch["img_input_dict"]["pixel_values"])
    out["rows_json"] = np.frombuffer(json.dumps(rows).encode(), dtype=np.uint8)
    for k in ("concatenated_input_ids", "concatenated_attention_mask", "concatenated_labels"):
        out["cat." + k] = to_np(cb[k])
    out["vit_feat"] = to_np(inter["vit_feat"][:B])
    out["image_features"] = to_np(inter["image_features"][:B])
    out["merged_embeds"] = to_np(inter["merged"])
    out["merged_mask"] = to_np(inter["mask"])
    out["merged_labels"] = to_np(inter["labels"])
    out["merged_pos"] = to_np(inter["pos"])
    out["image_position_map"] = to_np(inter["img_map"])
    out["hidden_l0"] = to_np(inter["hidden_l0"])
    out["hidden_last"] = to_np(inter["hidden_last"])
    out["logits"] = to_np(logits)
    out["ref_logits_mean"] = np.array([float(ref_logits[:B].mean()), float(ref_logits[B:].mean())])
    out["policy_logps"] = to_np(logps)
    out["policy_logps_avg"] = to_np(logps_avg)
    out["policy_logps_ddpo"] = to_np(logps_ddpo)
    out["ref_logps"] = to_np(ref_logps)
    out["ref_logps_ddpo"] = to_np(ref_logps_ddpo)
    out["loss_mean_sigmoid"] = np.array(float(loss))
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {name}: loss={float(loss):.6f} logps={to_np(logps)} ref={to_np(ref_logps)} "
          f"gradnorm={gsq ** 0.5:.4f} -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def gen_known_answers():
    """Small known-answer vectors of the pure-integer / scalar pieces."""
    out = {}
    # collator on two ragged rows
    rows = [
        dict(prompt_input_ids=[1, 2, 3], prompt_attention_mask=[1, 1, 1],
             chosen_input_ids=[1, 2, 3, 4, 5], chosen_attention_mask=[1] * 5, chosen_labels=[-100, -100, -100, 4, 5],
             rejected_input_ids=[1, 2, 3, 6], rejected_attention_mask=[1] * 4, rejected_labels=[-100, -100, -100, 6],
             img_path="a.jpg", reference_chosen_logps=-1.5, reference_rejected_logps=-2.5),
        dict(prompt_input_ids=[1, 2], prompt_attention_mask=[1, 1],
             chosen_input_ids=[1, 2, 4], chosen_attention_mask=[1] * 3, chosen_labels=[-100, -100, 4],
             rejected_input_ids=[1, 2, 7, 8, 9, 10], rejected_attention_mask=[1] * 6,
             rejected_labels=[-100, -100, 7, 8, 9, 10],
             img_path="b.jpg", reference_chosen_logps=-0.5, reference_rejected_logps=-3.0),
    ]
    coll = VLDPODataCollatorWithPadding(pad_token_id=0, label_pad_token_id=-100, is_encoder_decoder=False)
    b = coll(rows)
    out["collator_rows_json"] = np.frombuffer(json.dumps(rows).encode(), dtype=np.uint8)
    for k, v in b.items():
        if isinstance(v, torch.Tensor):
            out["collator." + k] = to_np(v)
    # dpo_loss sweep for every loss type
    g = torch.Generator().manual_seed(7)
    pc, pr, rc, rr = [torch.randn(6, generator=g) * 8 - 40 for _ in range(4)]
    out["kl.pc"], out["kl.pr"], out["kl.rc"], out["kl.rr"] = map(to_np, (pc, pr, rc, rr))
    for lt in LOSS_TYPES:
        for ls in (0.0, 0.2):
            for rf in (False, True):
                for beta in (0.1, 0.5):
                    l, c, r = dpo_loss_ref(lt, beta, pc, pr, rc, rr, label_smoothing=ls, reference_free=rf)
                    key = f"kl.{lt}.ls{ls}.rf{int(rf)}.b{beta}"
                    out[key + ".losses"], out[key + ".cr"], out[key + ".rr"] = to_np(l), to_np(c), to_np(r)
    # policy == ref => ln 2
    l, _, _ = dpo_loss_ref("sigmoid", 0.1, pc, pr, pc, pr)
    out["kl.ln2"] = to_np(l)
    # get_batch_logps on random logits incl. ddpo (Appendix A.2 known answer)
    logits = torch.randn(4, 13, 60, generator=g)
    chosen = [-100, -100, 10, 11, 12, 13, 30, 31, 20, 21, 22, -100, -100]
    rejected = [-100, -100, 10, 11, 12, 13, 40, 41, 42, 20, 21, 22, 50]
    labels = torch.tensor([chosen, chosen[:5] + [7, 8, 9, 1, 2, 3, 4, 5], rejected, rejected[:6] + [9] * 7])
    out["lp.logits"], out["lp.labels"] = to_np(logits), to_np(labels)
    out["lp.sum"] = to_np(VLDPOTrainer.get_batch_logps(logits, labels))
    out["lp.avg"] = to_np(VLDPOTrainer.get_batch_logps(logits, labels, average_log_prob=True))
    out["lp.ddpo"] = to_np(VLDPOTrainer.get_batch_logps(logits, labels, mask_shared_tokens=True))
    sh = labels[:, 1:].clone()
    sh[sh == -100] = 0
    c_mod, r_mod = get_diff_ids(sh[0].tolist(), sh[2].tolist(), min_match_size=3)
    out["lp.ddpo_c0"], out["lp.ddpo_r0"] = np.array(c_mod), np.array(r_mod)
    # merge known answer (Appendix C): [1,<img>,5,6,7,0,0] with 4 image features
    fake_self = types.SimpleNamespace(pad_token_id=99, config=types.SimpleNamespace(image_token_index=50, ignore_index=-100))
    ids = torch.tensor([[1, 50, 5, 6, 7, 0, 0], [1, 2, 50, 6, 7, 8, 9]])
    emb = torch.randn(2, 7, 8, generator=g)
    feats = torch.randn(2, 4, 8, generator=g)
    am = torch.tensor([[1, 1, 1, 1, 1, 0, 0], [1] * 7])
    lab = torch.tensor([[-100, -100, -100, 6, 7, -100, -100], [-100, -100, -100, -100, 7, 8, 9]])
    fe, fm, fl, pos, imap = LlavaForRL._merge_input_ids_with_image_features(fake_self, feats, emb, ids, am, lab)
    out["mg.ids"], out["mg.emb"], out["mg.feats"], out["mg.am"], out["mg.lab"] = map(to_np, (ids, emb, feats, am, lab))
    out["mg.out_emb"], out["mg.out_mask"], out["mg.out_labels"], out["mg.out_pos"], out["mg.out_map"] = map(
        to_np, (fe, fm, fl, pos, imap))
    path = os.path.join(OUT_DIR, "known_answers.npz")
    os.makedirs(OUT_DIR, exist_ok=True)
    np.savez_compressed(path, **out)
    print(f"[golden] known answers -> {path} ({os.path.getsize(path) / 1e3:.1f} kB)")


def gen_processor_answers():
    """The reference's OWN LlavaProcessor (src/vlrlhf/models/Llava/__init__.py:315-432) + VLProcessor base
    (base/processor.py:11-164) run on the committed tiny real tokenizer (tests/golden/tiny_llava_processor, a
    LlamaTokenizerFast with BOS and cross-boundary merges): the prompt string VLDPOTrainer.tokenize_row hands to trl
    (base/trainer.py:105-118) and the token / label lists of process_batch_conv.  -> tests/golden/processor_answers.json"""
    from vlrlhf.models.Llava import LlavaProcessor
    proc = LlavaProcessor(os.path.join(OUT_DIR, "tiny_llava_processor"))
    proc.train()
    rows = [dict(prompt="What is shown in this picture?", chosen="A small brown dog is running.", rejected="Two people sitting at a table.", img_path="a.jpg"),
            dict(prompt="<image>Is there a cat in the photo?", chosen="No, there is no cat.", rejected="Yes.", img_path=["b.jpg"]),
            dict(prompt="How many apples are on the table", chosen="three apples", rejected="There are three apples and one orange on the table.", img_path="c.jpg")]
    out = dict(pad_token_id=proc.tokenizer.pad_token_id, unk_token_id=proc.tokenizer.unk_token_id, rows=[])
    for r in rows:
        prompt = proc.format_multimodal_prompt(r["prompt"], r["img_path"])
        conv = proc.make_single_turn_conv(prompt, "")
        pr = proc.process_batch_conv([conv], system_message=None, add_end_for_empty_value=False)
        full = proc.process_batch_conv([proc.make_single_turn_conv(prompt, r["chosen"])])
        out["rows"].append(dict(row=r, formatted_prompt=prompt, conv=conv, prompt_raw_str=pr["raw_str"][0],
                                prompt_full=pr["full"], chosen_full=full["full"], chosen_raw_str=full["raw_str"][0],
                                valid=proc.is_multimodal_prompt_valid(prompt), stripped=proc.remove_image_placeholder(prompt)))
    # VLProcessor.__call__ (base/processor.py:95-164; the base method, so no image file is opened): raw texts are wrapped in a single-turn
    # conversation and run through process_batch_conv - a text without the placeholder gets it prepended - then padded on either side
    from vlrlhf.base.processor import VLProcessor as RefVLProcessor
    texts = ["What is shown in this picture?", "<image>\nIs there a cat in the photo?", "How many apples are on the table"]
    paths = ["a.jpg", ["b.jpg"], "c.jpg"]
    convs = [proc.make_single_turn_conv(proc.format_multimodal_prompt(r["prompt"], r["img_path"]), r["chosen"]) for r in rows]
    out["call"] = []
    for kw in (dict(texts=list(texts), images_path=paths), dict(texts=list(texts), images_path=paths, padding_side="right"),
               dict(texts=list(texts)), dict(convs=convs), dict(convs=convs, padding_side="right")):
        enc = RefVLProcessor.__call__(proc, **json.loads(json.dumps(kw)))       # (the reference formats `texts` in place: hand it a copy)
        out["call"].append(dict(kwargs=kw, **{k: enc[k].tolist() for k in ("input_ids", "attention_mask", "labels")}))
    path = os.path.join(OUT_DIR, "processor_answers.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"[golden] processor answers -> {path}")


def gen_dataset_answers():
    """The reference's own pair mining for VLFeedback (src/vlrlhf/utils/data.py:11-82), captured without the hub: `load_dataset`
    is replaced by a stand-in whose .map() hands us the reference's nested `make_batch_pairs`, which is then run on synthetic
    completion lists for score_margin -1 (largest gap only) and 1.0.  -> tests/golden/vlfeedback_pairs.json"""
    import vlrlhf.utils.data as RD
    captured = {}

    class FakeDS:
        column_names = ["prompt", "img_path", "completions", "id"]

        def map(self, fn, **kw):
            captured["fn"] = fn
            return self

    RD.load_dataset = lambda *a, **k: FakeDS()
    g = np.random.default_rng(3)
    aspects = ("Helpfulness", "Ethical Considerations", "Visual Faithfulness")
    samples = dict(prompt=[], img_path=[], completions=[])
    for i in range(9):
        n = int(g.integers(2, 5))
        annos = []
        for c in range(n):
            a = {asp: {"Rating": str(int(g.integers(1, 6)))} for asp in aspects}
            if i == 4 and c == 1:
                a["Helpfulness"]["Rating"] = "N/A"                    # the reference skips pairs whose rating does not parse
            annos.append(a)
        if i == 6:
            annos = [annos[0]] * n                                      # all ties: the sample yields nothing
        samples["prompt"].append(f"question {i}?")
        samples["img_path"].append(f"img_{i}.jpg")
        samples["completions"].append(dict(annotations=annos, response=[f"answer {i}.{c}" for c in range(n)]))
    out = dict(samples=samples, results={})
    for margin in (-1, 1.0):
        RD.make_vlfeedback_paired_dataset(types.SimpleNamespace(score_margin=margin))
        res = captured["fn"](samples)
        out["results"][str(margin)] = {k: list(v) for k, v in res.items()}
    path = os.path.join(OUT_DIR, "vlfeedback_pairs.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"[golden] vlfeedback pairs -> {path}: {[len(v['prompt']) for v in out['results'].values()]} pairs")


OPT = dict(lr=1e-3, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.05, max_grad_norm=1.0)

CASES = {
    # odd sizes on purpose: nothing here is a multiple of a tile
    "llava_tiny": dict(
        vit_hidden=32, vit_mlp=80, vit_layers=3, vit_heads=4, image_size=28, patch_size=14,
        hidden=48, inter=112, layers=2, heads=4, vocab=123, image_token=120,
        pairs=2, prompt_len=(6, 9), resp_len=(5, 11), beta=0.1, optim=OPT, w_scale=4.0, perturb=0.08,
        store_all_after_step=True),
    # kernel-compatible widths (LLM head_dim 128, ViT head_dim 64) at toy depth: the HIP path is checked
    # against THIS fixture directly, not only against the CPU restatement
    "llava_hipsmall": dict(
        vit_hidden=64, vit_mlp=128, vit_layers=3, vit_heads=1, image_size=56, patch_size=14,
        hidden=128, inter=256, layers=2, heads=1, vocab=192, image_token=180,
        pairs=2, prompt_len=(8, 12), resp_len=(6, 20), beta=0.1, optim=OPT, w_scale=3.0, perturb=0.05),
}

if __name__ == "__main__":
    torch.set_num_threads(4)
    if sys.argv[1:] == ["processor"]:        # adds the processor fixture without regenerating the tensor fixtures
        gen_processor_answers()
        sys.exit(0)
    if sys.argv[1:] == ["datasets"]:
        gen_dataset_answers()
        sys.exit(0)
    gen_known_answers()
    gen_processor_answers()
    for i, (name, cfg) in enumerate(CASES.items()):
        gen_case(name, cfg, seed=1000 + 17 * i)
