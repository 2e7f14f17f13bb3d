"""InternLM-XComposer2-VL wrapper for the MI355X DPO path - mirror of /root/reference/src/vlrlhf/models/InternLMXC2/__init__.py
(InternLMXC2ForRL :32-283, InternLMXC2Processor :297-421, InternLMXC2DPODataCollatorWithPadding :424-431, core_mapper :483-496) for
BASELINE.json configs[4].  The model runs on vlrlhf.engine_internlm (PLoRA on the image rows, fused grouped-query wqkv, index-based
rotary); the merge is the LLaVA one (<ImageHere> expands to the 1225 projected patches of the 490 px image)."""
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Literal, Optional, Union

import torch

from ...base.collator import VLDPODataCollatorWithPadding
from ...base.processor import VLChatTemplate, VLProcessor
from ...base.trainer import VLDPOTrainer
from ...engine_internlm import InternLMHipEngine
from ...utils.common import flatten_list
from ..Llava import LazyLogits, LlavaForRL, LlavaRLOutputWithPast, _HiddenFn
from ..QwenVL import load_qwen_pixel_values
from ..utils import ModelCoreMapper

INTERNLM_XC2_VL_7B = dict(family="internlm_xc2", hidden=4096, inter=14336, layers=32, heads=32, kv_heads=8, vocab=92544, rms_eps=1e-5,
                          rope_theta=1000000.0, vit_hidden=1024, vit_mlp=4096, vit_layers=24, vit_heads=16, image_size=490, patch_size=14,
                          vit_feature_layer=-1, image_token=92544, model_pad_token_id=2, plora_r=256, plora_alpha=256, plora_dropout=0.05)

# the meta instruction the reference prepends to every conversation (InternLMXC2/__init__.py:349) - a constant of the released model
META_INSTRUCTION = (
    "<s>[UNUSED_TOKEN_146]system\nYou are an AI assistant whose name is InternLM-XComposer (浦语·灵笔).\n"
    "-InternLM-XComposer (浦语·灵笔) is a multi-modality conversational language model that is developed by Shanghai AI Laboratory "
    "(上海人工智能实验室). It is designed to be helpful, honest, and harmless.\n"
    "-InternLM-XComposer (浦语·灵笔) can understand and communicate fluently in the language chosen by the user such as English and 中文.\n"
    "-InternLM-XComposer (浦语·灵笔) is capable of comprehending and articulating responses effectively based on the provided image."
    "[UNUSED_TOKEN_145]\n")


def _cfg_from_hf(hf: dict) -> dict:
    """InternLMXcomposer2Config (configuration_internlm_xcomposer2.py:90-137) + the fields the reference's checkpoint adds"""
    v = hf.get("vision", {})
    return dict(family="internlm_xc2", hidden=hf["hidden_size"], inter=hf["intermediate_size"], layers=hf["num_hidden_layers"],
                heads=hf["num_attention_heads"], kv_heads=hf.get("num_key_value_heads") or hf["num_attention_heads"], vocab=hf["vocab_size"],
                rms_eps=hf.get("rms_norm_eps", 1e-5), rope_theta=hf.get("rope_theta", 1000000.0),
                vit_hidden=v.get("hidden_size", 1024), vit_mlp=v.get("intermediate_size", 4096), vit_layers=v.get("num_hidden_layers", 24),
                vit_heads=v.get("num_attention_heads", 16), image_size=hf.get("img_size", 490), patch_size=v.get("patch_size", 14),
                vit_feature_layer=-1, image_token=hf.get("image_token_index", hf["vocab_size"]), model_pad_token_id=hf.get("pad_token_id", 2),
                ignore_index=hf.get("ignore_index", -100), plora_r=256, plora_alpha=256, plora_dropout=0.05)


def _hf_from_cfg(c: dict) -> dict:
    return dict(architectures=["InternLMXComposer2ForCausalLM"], model_type="internlmxcomposer2", vocab_size=c["vocab"], hidden_size=c["hidden"],
                intermediate_size=c["inter"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"], num_key_value_heads=c.get("kv_heads", c["heads"]),
                rms_norm_eps=c.get("rms_eps", 1e-5), rope_theta=c.get("rope_theta", 1000000.0), bias=False, pad_token_id=c["model_pad_token_id"],
                img_size=c["image_size"], image_token_index=c["image_token"], ignore_index=-100, max_length=4096, torch_dtype="bfloat16",
                vision=dict(hidden_size=c["vit_hidden"], intermediate_size=c["vit_mlp"], num_hidden_layers=c["vit_layers"],
                            num_attention_heads=c["vit_heads"], patch_size=c["patch_size"]))


class InternLMXC2ForRL(LlavaForRL):
    engine_cls = InternLMHipEngine

    def __init__(self, cfg: dict, engine=None, weights=None, trainable: bool = True):
        super().__init__(dict(cfg, family="internlm_xc2"), engine=engine, weights=weights, trainable=trainable)
        self.pad_token_id = cfg.get("model_pad_token_id", 2)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, torch_dtype=None, use_flash_attention_2=None, **kwargs):
        from safetensors.torch import load_file
        path = pretrained_model_name_or_path
        with open(os.path.join(path, "config.json")) as f:
            hf = json.load(f)
        m = cls(_cfg_from_hf(hf))
        m.hf_config = hf
        sd = {}
        idx = os.path.join(path, "model.safetensors.index.json")
        files = sorted(set(json.load(open(idx))["weight_map"].values())) if os.path.exists(idx) else \
            [f for f in os.listdir(path) if f.endswith(".safetensors")]
        for fn in files:
            sd.update(load_file(os.path.join(path, fn)))
        m.engine.load_state_dict(sd)
        return m

    def save_pretrained(self, output_dir, max_shard_bytes: int = 5 << 30, state_dict=None):
        if getattr(self, "hf_config", None) is None:
            self.hf_config = _hf_from_cfg(self.engine.cfg)
        return super().save_pretrained(output_dir, max_shard_bytes, state_dict)

    @property
    def default_lora_target(self):
        return ["attention.wqkv", "attention.wo", "feed_forward.w1", "feed_forward.w2", "feed_forward.w3"]

    def freeze_vision_tower(self):
        """reference :252-255: the tower AND vision_proj are frozen (the engine keeps both outside the optimizer's range)"""
        self._vision_frozen = True

    def prepare_default_generation_kwargs(self, generation_config):
        generation_config.do_sample = False
        generation_config.eos_token_id = 2
        return dict(generation_config=generation_config)

    def forward(self, input_ids=None, pixel_values=None, im_mask=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        """reference :107-236 on the training path -> lazy `logits`, the EXPANDED `labels`, `image_position_map` (= im_mask)"""
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("generation / KV-cache inputs are outside the MI355X DPO training path")
        assert pixel_values is not None
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        dup = int(getattr(pixel_values, "_vlr_dup", 1))
        grad = torch.is_grad_enabled() and self._trainable and self.training
        if grad:
            hidden = _HiddenFn.apply(self._anchor, self, input_ids, attention_mask, labels, pixel_values, dup, None)
            c = self._last_ctx
        else:
            c = self.engine.forward_hidden(self.weights, input_ids, attention_mask, labels, pixel_values, image_dup=dup, save=False,
                                           tag="policy_ng" if self.weights is self.engine.policy else "ref")
            hidden = c["hidden"]
        out_labels = c["labels"] if labels is not None else torch.full_like(c["mask"], -100, dtype=torch.long)
        if c.get("meta") is not None:
            out_labels._vlr_meta = c["meta"]
        return LlavaRLOutputWithPast(loss=None, logits=LazyLogits(self.engine, c, hidden), labels=out_labels, image_position_map=c["img_map"])


class InternLMXC2Processor(VLProcessor):
    def __init__(self, model_name_or_path=None, tokenizer=None, image_size: Optional[int] = None, **kwargs) -> None:
        if tokenizer is None:
            import transformers
            tokenizer = transformers.AutoTokenizer.from_pretrained(model_name_or_path, use_fast=False, trust_remote_code=True)
            tokenizer.add_tokens("<ImageHere>", special_tokens=True)
        self._tok = tokenizer
        self.image_size = image_size
        if image_size is None:
            self.image_size = 490
            cfg = os.path.join(model_name_or_path, "config.json") if model_name_or_path else None
            if cfg and os.path.exists(cfg):
                self.image_size = json.load(open(cfg)).get("img_size", 490)

    @property
    def tokenizer(self):
        return self._tok

    @property
    def chat_template(self):
        return VLChatTemplate(system_begin="<s>[UNUSED_TOKEN_146]system\n", system_end="[UNUSED_TOKEN_145]\n",
                              user_begin="[UNUSED_TOKEN_146]user\n", user_end="[UNUSED_TOKEN_145]\n",
                              assistant_begin="[UNUSED_TOKEN_146]assistant\n", assistant_end="[UNUSED_TOKEN_145]\n",
                              image_placeholder="<ImageHere>")

    def image_processor(self, images):
        """reference :318-333: bicubic resize to img_size, [0,1], CLIP normalisation"""
        return load_qwen_pixel_values(list(images), self.image_size)

    def save_pretrained(self, output_dir):
        return None

    def process_batch_conv(self, sources, system_message=None, add_end_for_empty_value=False):
        """reference :338-383: the running text (meta instruction + turns) is re-tokenised after every turn; the ids a turn added get
        label -100 except the assistant's words, which are copied from the tokenisation of the bare answer (aligned at the end; when the
        tokenizer merged across the turn boundary only the last `added` ids are taken)."""
        if not isinstance(sources, list) or not isinstance(sources[0], list):
            raise ValueError("sources must be a batch of conversations, eg. List[List[Dict]]")
        t = self.chat_template
        begin = {"user": t.user_begin, "assistant": t.assistant_begin}
        end = {"user": t.user_end, "assistant": t.assistant_end}
        out = dict(input_ids=[], attention_mask=[], labels=[])
        raw_texts = []
        for source in sources:
            raw, labels, prev = META_INSTRUCTION, [], 0
            ids, masks = [], []
            for i, s in enumerate(source):
                piece = begin[s["from"]] + s["value"] + (end[s["from"]] if s["value"] != "" or add_end_for_empty_value else "")
                raw += piece
                words = self.tokenizer(s["value"], padding=False, add_special_tokens=(i == 0))["input_ids"]
                alone = self.tokenizer(piece, padding=False, add_special_tokens=(i == 0))["input_ids"]
                cur = self.tokenizer(raw)
                ids, masks = cur["input_ids"], cur["attention_mask"]
                added = len(ids) - prev
                prev = len(ids)
                labels.extend([-100] * added)
                if s["from"] == "assistant" and len(words) != 0:
                    if added < len(alone):
                        labels[-added:] = words[-added:]
                    else:
                        labels[-len(words):] = words
            labels = [l if m == 1 else -100 for l, m in zip(labels, masks)]
            assert len(ids) == len(masks) == len(labels)
            out["input_ids"].append(ids)
            out["attention_mask"].append(masks)
            out["labels"].append(labels)
            raw_texts.append(raw)
        return {"prompt": None, "answer": None, "full": out, "raw_str": raw_texts}

    @staticmethod
    def format_multimodal_prompt(prompt: str, img_paths: Optional[Union[List[str], str]] = None):
        if img_paths is None:
            return prompt
        if isinstance(img_paths, str):
            img_paths = [img_paths]
        if len(img_paths) == 1 and "<image>" not in prompt:
            return "<ImageHere>" + prompt
        assert prompt.count("<image>") == len(img_paths), \
            f"The number of given image ({len(img_paths)}) does not match the number of image placeholders in the prompt: {prompt}"
        return prompt.replace("<image>", "<ImageHere>")

    @staticmethod
    def remove_image_placeholder(prompt: str):
        return prompt.replace("<ImageHere>", "")

    @staticmethod
    def is_multimodal_prompt_valid(prompt: str):
        return "<ImageHere>" in prompt

    def train(self):
        self.tokenizer.padding_side = "right"

    def infer(self):
        self.tokenizer.padding_side = "left"

    def __call__(self, texts=None, convs=None, images_path=None, padding: bool = True,
                 padding_side: Literal["right", "left"] = "left", check_format: bool = True):
        inputs = super().__call__(texts, convs, images_path, padding, padding_side, check_format)
        if images_path is not None:
            inputs["pixel_values"] = self.image_processor(flatten_list(images_path))
        return inputs


@dataclass
class InternLMXC2DPODataCollatorWithPadding(VLDPODataCollatorWithPadding):
    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        padded = super().__call__(features)
        padded["img_input_dict"] = dict(pixel_values=self.processor.image_processor(padded["img_path"]))
        return padded


class InternLMXC2DPOTrainer(VLDPOTrainer):
    ...


core_mapper = ModelCoreMapper(
    model=InternLMXC2ForRL,
    processor=InternLMXC2Processor,
    dpo_collator=InternLMXC2DPODataCollatorWithPadding,
    dpo_trainer=InternLMXC2DPOTrainer,
)
