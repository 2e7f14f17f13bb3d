"""LLaVA-Next (LLaVA-1.6, anyres tiles; Mistral-7B or Vicuna-7B decoder) wrapper for the MI355X DPO path - mirror of
/root/reference/src/vlrlhf/models/LlavaNext/__init__.py (LlavaNextForRLOutputWithPast :25-33, LlavaNextForRL :36-371,
LlavaNextProcessor :393-514, LlavaNextDPODataCollatorWithPadding :517-523, core_mapper at the end of the file).

What differs from LLaVA-1.5 on the hot path (SURVEY.md A15, Appendix A.8, Appendix B "C4"):
  * every image is a base tile + an anyres grid of 336x336 tiles (`image_sizes` picks the grid); all tiles go through the
    frozen CLIP ViT and the projector, then `pack_image_features` lays the tiles out on their grid, crops the padding away and
    appends the trainable `image_newline` embedding to every grid row - here a row GATHER from the projector output whose
    index list is built on the host (models/LlavaNext/anyres.py), and a row SCATTER + fixed-order column sum in the backward;
  * each `<image>` id expands to ITS image's feature length, padding is recognised from the attention mask and the merge
    returns `(embeds, mask, position_ids, labels, map)` - a different tuple order from LLaVA-1.5 (reference :171);
  * Mistral: grouped-query attention (32 query heads share 8 K/V heads: vlr_attn_*_gqa), I = 14336, rope_theta 1e6.
The arithmetic is the same engine (vlrlhf.engine.LlavaHipEngine with `kv_heads` / `image_grid_pinpoints` in its config)."""
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch

from ...base.collator import VLDPODataCollatorWithPadding
from ...base.processor import VLChatTemplate
from ...base.trainer import VLDPOTrainer
from ...utils.common import flatten_list
from ..Llava import CLIP_MEAN, CLIP_STD, LlavaForRL, LlavaProcessor, LlavaRLOutputWithPast
from ..utils import ModelCoreMapper
from . import anyres

LLAVA_NEXT_MISTRAL_7B = dict(
    vit_hidden=1024, vit_mlp=4096, vit_layers=24, vit_heads=16, image_size=336, patch_size=14,
    hidden=4096, inter=14336, layers=32, heads=32, kv_heads=8, vocab=32064, image_token=32000, model_pad_token_id=32001,
    rms_eps=1e-5, rope_theta=1000000.0, padding_side="left",
    image_grid_pinpoints=[[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]])


@dataclass
class LlavaNextForRLOutputWithPast(LlavaRLOutputWithPast):
    pass


class LlavaNextForRL(LlavaForRL):
    """`model(input_ids=, attention_mask=, labels=, use_cache=False, pixel_values=[n, tiles, 3, s, s], image_sizes=[n, 2])` ->
    output with lazy `.logits`, expanded `.labels`, `.image_position_map` (reference forward :173-345)."""

    padding_side = "left"          # transformers LlavaNextForConditionalGeneration default; read by the merge when no row is padded

    def __init__(self, cfg: dict, *a, **k):
        cfg = dict(cfg)
        if not cfg.get("image_grid_pinpoints"):
            raise ValueError("LlavaNextForRL needs image_grid_pinpoints in its config")
        cfg.setdefault("padding_side", self.padding_side)
        super().__init__(cfg, *a, **k)

    def create_reference_model(self):
        ref = LlavaNextForRL(dict(self.engine.cfg), engine=self.engine, weights=self.weights.clone(), trainable=False)
        ref.eval()
        return ref

    def prefetch_vision(self, img_input_dict):
        """run the frozen tower on the CALLER's stream before the reference pass is forked onto its side stream: both passes then
        read the cached features (computing them inside the side-stream pass would let the policy pass read them unordered)"""
        pv, sizes = img_input_dict.get("pixel_values"), img_input_dict.get("image_sizes")
        if pv is not None and sizes is not None:
            self.engine.anyres_vision_features(pv, sizes, int(getattr(pv, "_vlr_dup", 1)))

    @staticmethod
    def _merge_input_ids_with_image_features(image_features, feature_lens, inputs_embeds, input_ids, attention_mask, position_ids=None,
                                             labels=None, image_token_index=None, ignore_index=-100, padding_side="left"):
        """the reference's merge (:38-171) for callers that want tensors: returns (embeds, mask, position_ids, labels, map).
        The engine itself only needs the index part (anyres.merge_index) and gathers on the device."""
        mi = anyres.merge_index(input_ids.cpu().numpy(), attention_mask.cpu().numpy(), labels.cpu().numpy() if labels is not None else None,
                                feature_lens.cpu().numpy(), int(image_token_index), padding_side, dup=1, ignore_index=ignore_index)
        dev = inputs_embeds.device
        src = torch.from_numpy(mi["src"]).to(dev).long()
        B, S = src.shape
        out = torch.zeros(B, S, inputs_embeds.shape[-1], dtype=inputs_embeds.dtype, device=dev)
        tb, ts = torch.where(src >= 0)
        out[tb, ts] = inputs_embeds[tb, src[tb, ts]]
        ib, is_ = torch.where((src < 0) & (src != anyres.SRC_ZERO))
        out[ib, is_] = image_features[(-src[ib, is_] - 1)].to(out.dtype)
        lab = torch.from_numpy(mi["labels"]).to(dev) if labels is not None else None
        return (out, torch.from_numpy(mi["mask"]).to(dev).to(attention_mask.dtype), torch.from_numpy(mi["pos"]).to(dev).long(), lab,
                torch.from_numpy(mi["img_map"]).to(dev))


class LlavaNextProcessor(LlavaProcessor):
    def __init__(self, model_name_or_path=None, tokenizer=None, image_processor=None, llm_name: str = "mistral", **kwargs) -> None:
        if model_name_or_path is not None:
            import transformers
            self.processor = transformers.LlavaNextProcessor.from_pretrained(model_name_or_path, **kwargs)
            self._tok, self._ip = self.processor.tokenizer, self.processor.image_processor
            with open(os.path.join(model_name_or_path, "config.json")) as f:
                self._llm_name = str(json.load(f).get("text_config", {}).get("_name_or_path", llm_name))
        else:
            self.processor = None
            self._tok, self._ip, self._llm_name = tokenizer, image_processor, llm_name

    @property
    def chat_template(self):
        """reference :404-427: keyed on config.text_config._name_or_path"""
        if "mistral" in self._llm_name:
            return VLChatTemplate(system_begin=None, system_end=None, user_begin="[INST] ", user_end=" [/INST]", assistant_begin="",
                                  assistant_end="", image_placeholder="<image>\n")
        if "vicuna" in self._llm_name:
            return VLChatTemplate(system_begin=None, system_end=None, user_begin="USER: ", user_end="", assistant_begin="ASSISTANT: ",
                                  assistant_end="", image_placeholder="<image>\n")
        raise ValueError(f"unknown LLaVA-Next language model '{self._llm_name}' (reference supports mistral / vicuna)")

    def process_batch_conv(self, sources, system_message=None, add_end_for_empty_value=False):
        """reference :436-480: as LLaVA-1.5, with the Vicuna system sentence prepended for vicuna checkpoints"""
        if "vicuna" not in self._llm_name:
            return super().process_batch_conv(sources, system_message, add_end_for_empty_value)
        prefix = ("A chat between a curious human and an artificial intelligence assistant. The assistant gives helpful, detailed, and "
                  "polite answers to the human's questions. ")
        if not isinstance(sources, list) or not isinstance(sources[0], list):
            raise ValueError("sources must be a batch of conversations, eg. List[List[Dict]]")
        t = self.chat_template
        begin = {"user": t.user_begin, "assistant": t.assistant_begin}
        end = {"user": t.user_end, "assistant": t.assistant_end}
        raw_texts, b_ids, b_masks, b_labels = [], [], [], []
        for source in sources:
            raw, labels, prev = prefix, [], 0
            ids, masks = [], []
            for i, s in enumerate(source):
                raw += begin[s["from"]] + s["value"] + (end[s["from"]] if s["value"] != "" or add_end_for_empty_value else "")
                text_tokens = self.tokenizer(s["value"], padding=False, add_special_tokens=(i == 0))
                cur = self.tokenizer(raw)
                ids, masks = cur["input_ids"], cur["attention_mask"]
                ext = len(ids) - prev
                prev = len(ids)
                labels.extend([-100] * ext)
                if s["from"] == "assistant" and len(text_tokens["input_ids"]) != 0:
                    n = min(ext, len(text_tokens["input_ids"]), len(labels))
                    labels[-n:] = text_tokens["input_ids"][-n:]
            labels = [l if m == 1 else -100 for l, m in zip(labels, masks)]
            b_ids.append(ids)
            b_masks.append(masks)
            b_labels.append(labels)
            raw_texts.append(raw)
        return {"prompt": None, "answer": None, "full": dict(input_ids=b_ids, attention_mask=b_masks, labels=b_labels), "raw_str": raw_texts}

    def train(self):
        pass                      # reference :503-504: no pad-token change

    def __call__(self, texts=None, convs=None, images_path=None, padding=True, padding_side="left", check_format=True):
        inputs = super(LlavaProcessor, self).__call__(texts, convs, images_path, padding, padding_side, check_format)
        if images_path is not None:
            inputs.update(load_anyres_images(flatten_list(images_path), self.image_processor))
        return inputs


def load_anyres_images(items, image_processor=None) -> Dict[str, torch.Tensor]:
    """paths / PIL images -> {'pixel_values': [n, max_tiles, 3, s, s], 'image_sizes': [n, 2]} through the HF
    LlavaNextImageProcessor (reference :517-523).  Synthetic items `dict(pixel_values=[tiles,3,s,s], image_size=(h, w))` pass through."""
    if all(isinstance(i, dict) for i in items):
        mt = max(i["pixel_values"].shape[0] for i in items)
        pv = torch.zeros(len(items), mt, *items[0]["pixel_values"].shape[1:])
        for k, i in enumerate(items):
            pv[k, : i["pixel_values"].shape[0]] = i["pixel_values"].float()
        return dict(pixel_values=pv, image_sizes=torch.tensor([list(i["image_size"]) for i in items], dtype=torch.long))
    from PIL import Image
    imgs = [Image.open(i).convert("RGB") if isinstance(i, str) else i for i in items]
    out = image_processor(images=imgs, return_tensors="pt")
    return dict(pixel_values=out["pixel_values"], image_sizes=out["image_sizes"])


@dataclass
class LlavaNextDPODataCollatorWithPadding(VLDPODataCollatorWithPadding):
    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        padded = super().__call__(features)
        ip = self.processor.image_processor if self.processor is not None else None
        padded["img_input_dict"] = load_anyres_images(padded["img_path"], ip)
        return padded


class LlavaNextDPOTrainer(VLDPOTrainer):
    ...


core_mapper = ModelCoreMapper(
    model=LlavaNextForRL,
    processor=LlavaNextProcessor,
    dpo_collator=LlavaNextDPODataCollatorWithPadding,
    dpo_trainer=LlavaNextDPOTrainer,
)
