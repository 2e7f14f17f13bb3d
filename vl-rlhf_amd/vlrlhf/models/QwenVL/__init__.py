"""Qwen-VL wrapper for the MI355X DPO path - mirror of /root/reference/src/vlrlhf/models/QwenVL/__init__.py (QwenVLForRL :24-46,
QwenVLProcessor :64-227, QwenVLDPODataCollatorWithPadding :230, QwenVLDPOTrainer.tokenize_row :256-347, core_mapper :358-371) for
BASELINE.json configs[2] (Qwen-VL-Chat, LoRA r 64 on c_attn / attn.c_proj / w1 / w2, scripts/dpo_qwenvl.sh).

Differences that are deliberate: the reference model opens the image files INSIDE forward (the paths are byte strings in the token
ids, modeling_qwen.py:525-537); here the collator loads and normalises them ahead of the step (`img_input_dict.pixel_values`, so
the prefetching loader can overlap it) and forward decodes the paths only when no pixels were handed in.  The ViT trunk is frozen;
the resampler (`attn_pool`) trains with the language model in a full fine-tune, as in the reference, and is frozen under LoRA."""
import json
import os
import re
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

import torch

from ...base.collator import VLDPODataCollatorWithPadding
from ...base.processor import VLChatTemplate, VLProcessor
from ...base.trainer import VLDPOTrainer
from ...engine_qwen import QwenVLHipEngine
from ...utils.common import flatten_list
from ..Llava import CLIP_MEAN, CLIP_STD, LazyLogits, LlavaForRL, LlavaRLOutputWithPast, _HiddenFn
from ..utils import ModelCoreMapper

IGNORE_TOKEN_ID = -100           # transformers LabelSmoother.ignore_index

QWEN_VL_CHAT = dict(family="qwen_vl", hidden=4096, inter=11008, layers=32, heads=32, vocab=151936, rms_eps=1e-6, rope_theta=10000.0,
                    image_start_id=151857, pad_token_id=151643,
                    visual=dict(width=1664, heads=16, layers=48, mlp_ratio=4.9231, patch_size=14, image_size=448, output_dim=4096,
                                n_queries=256))


def _cfg_from_hf(hf: dict) -> dict:
    """QWenConfig (configuration_qwen.py:15-36) -> engine config"""
    v = dict(hf.get("visual", {}))
    return dict(family="qwen_vl", hidden=hf["hidden_size"], inter=hf["intermediate_size"] // 2, layers=hf["num_hidden_layers"],
                heads=hf["num_attention_heads"], vocab=hf["vocab_size"], rms_eps=hf.get("layer_norm_epsilon", 1e-6),
                rope_theta=hf.get("rotary_emb_base", 10000.0), image_start_id=v.get("image_start_id", 151857),
                pad_token_id=hf.get("pad_token_id") or 151643,
                visual=dict(width=v.get("width", 1664), heads=v.get("heads", 16), layers=v.get("layers", 48), mlp_ratio=v.get("mlp_ratio", 4.9231),
                            patch_size=v.get("patch_size", 14), image_size=v.get("image_size", 448), output_dim=v.get("output_dim", 4096),
                            n_queries=v.get("n_queries", 256)))


def _hf_from_cfg(c: dict) -> dict:
    v = c["visual"]
    return dict(architectures=["QWenLMHeadModel"], model_type="qwen", vocab_size=c["vocab"], hidden_size=c["hidden"],
                intermediate_size=2 * c["inter"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                kv_channels=c["hidden"] // c["heads"], layer_norm_epsilon=c.get("rms_eps", 1e-6), rotary_emb_base=c.get("rope_theta", 10000.0),
                no_bias=True, bf16=True, torch_dtype="bfloat16", tie_word_embeddings=False,
                visual=dict(v, image_start_id=c["image_start_id"]))


def decode_image_paths(input_ids: torch.Tensor, image_start_id: int) -> List[str]:
    """modeling_qwen.py:525-534: every (<img>, </img>) pair holds the utf-8 bytes of a path, closed by <imgpad> = image_start_id + 2"""
    paths = []
    for row in input_ids.tolist():
        bos = [j for j, t in enumerate(row) if t == image_start_id]
        eos = [j for j, t in enumerate(row) if t == image_start_id + 1]
        assert len(bos) == len(eos), "unbalanced <img> / </img> markers"
        for a, b in zip(bos, eos):
            seg = row[a + 1: b - 1]
            seg = seg[: seg.index(image_start_id + 2)]
            paths.append(bytes(seg).decode("utf-8"))
    return paths


def load_qwen_pixel_values(items, image_size: int = 448) -> torch.Tensor:
    """visual.py:356-363 + :417-427 (`image_transform` of `encode`): RGB, bicubic resize to image_size x image_size, [0,1], CLIP
    normalisation -> fp32 [n,3,s,s].  Items that already are [3,s,s] tensors (synthetic data) pass through."""
    if len(items) and all(isinstance(i, torch.Tensor) for i in items):
        return torch.stack([i.float() for i in items])
    import numpy as np
    from PIL import Image
    mean, std = torch.tensor(CLIP_MEAN).view(3, 1, 1), torch.tensor(CLIP_STD).view(3, 1, 1)
    out = []
    for it in items:
        im = (Image.open(it) if isinstance(it, str) else it).convert("RGB").resize((image_size, image_size), Image.BICUBIC)
        x = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        out.append((x - mean) / std)
    return torch.stack(out)


class QwenVLForRL(LlavaForRL):
    engine_cls = QwenVLHipEngine

    def __init__(self, cfg: dict, engine=None, weights=None, trainable: bool = True):
        super().__init__(dict(cfg, family="qwen_vl"), engine=engine, weights=weights, trainable=trainable)
        self.pad_token_id = cfg.get("pad_token_id", 151643)
        self.config["visual"] = dict(cfg["visual"], image_start_id=cfg["image_start_id"])
        self._px_cache = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, torch_dtype=None, use_flash_attention_2=None, **kwargs):
        """reads a Qwen-VL(-Chat) checkpoint directory (config.json = QWenConfig, *.safetensors)"""
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
        return ["c_attn", "attn.c_proj", "w1", "w2"]

    def freeze_vision_tower(self):
        """reference :33-37 freezes the tower and re-enables `attn_pool`; under the shipped LoRA configuration peft freezes it again.
        Same here: the engine trains `ap.*` in a full fine-tune and nothing of the tower under LoRA."""
        self._vision_frozen = True

    def prepare_default_generation_kwargs(self, generation_config):
        generation_config.stop_words_ids = [[151645], [151644]]
        generation_config.do_sample = False
        return dict(generation_config=generation_config)

    def prefetch_vision(self, img_input_dict):
        pv = img_input_dict.get("pixel_values")
        if pv is not None:
            dup = getattr(pv, "_vlr_dup", 1)
            self.engine.vit_trunk(pv[: pv.shape[0] // dup] if dup > 1 else pv)      # the frozen trunk, once, on the caller's stream

    def _pixels_from_ids(self, input_ids):
        """the reference behaviour: image files named in the ids are opened here (cached per ids tensor: the reference pass and the
        policy pass of one step see the same batch)"""
        key = (input_ids.data_ptr(), tuple(input_ids.shape), input_ids._version)
        if self._px_cache is not None and self._px_cache[0] == key:
            return self._px_cache[1]
        paths = decode_image_paths(input_ids, self.engine.cfg["image_start_id"])
        px = load_qwen_pixel_values(paths, self.engine.cfg["visual"]["image_size"]).to(self.engine.dev) if paths else None
        self._px_cache = (key, px)
        return px

    def forward(self, input_ids=None, past_key_values=None, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, pixel_values=None):
        """reference QWenLMHeadModel.forward (modeling_qwen.py:775-860) on the training path -> lazy `logits`, `labels` (unchanged:
        the image slots are part of the ids) and `image_position_map`"""
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("generation / KV-cache inputs are outside the MI355X DPO training path")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if pixel_values is None:
            pixel_values = self._pixels_from_ids(input_ids)
        dup = int(getattr(pixel_values, "_vlr_dup", 1)) if pixel_values is not None else 1
        grad = torch.is_grad_enabled() and self._trainable and self.training
        if grad:
            hidden = _HiddenFn.apply(self._anchor, self, input_ids, attention_mask, labels, pixel_values, dup, None)
            c = self._last_ctx
        else:
            c = self.engine.forward_hidden(self.weights, input_ids, attention_mask, labels, pixel_values, image_dup=dup, save=False,
                                           tag="policy_ng" if self.weights is self.engine.policy else "ref")
            hidden = c["hidden"]
        out_labels = c["labels"] if labels is not None else None
        if out_labels is not None and c.get("meta") is not None:
            out_labels._vlr_meta = c["meta"]
        return LlavaRLOutputWithPast(loss=None, logits=LazyLogits(self.engine, c, hidden), labels=out_labels,
                                     image_position_map=c["img_map"])


# ----------------------------------------------------------------------------------------------------------
class QwenVLProcessor(VLProcessor):
    _mm_pattern = re.compile(r"<img>.+</img>\n")

    def __init__(self, model_name_or_path=None, tokenizer=None, image_size: Optional[int] = None, **kwargs) -> None:
        if tokenizer is None:
            import transformers
            kwargs.setdefault("trust_remote_code", True)
            tokenizer = transformers.AutoTokenizer.from_pretrained(model_name_or_path, **kwargs)
        self._tok = tokenizer
        self.image_size = image_size
        if image_size is None:
            self.image_size = 448
            cfg = os.path.join(model_name_or_path, "config.json") if model_name_or_path else None
            if cfg and os.path.exists(cfg):
                self.image_size = json.load(open(cfg)).get("visual", {}).get("image_size", 448)

    @property
    def tokenizer(self):
        return self._tok

    @property
    def chat_template(self):
        return VLChatTemplate(system_begin="<|im_start|>system", system_end="<|im_end|>", user_begin="<|im_start|>user",
                              user_end="<|im_end|>", assistant_begin="<|im_start|>assistant", assistant_end="<|im_end|>",
                              image_placeholder="<img>")

    @property
    def image_processor(self):
        return None

    def save_pretrained(self, output_dir):
        return None

    def process_batch_conv(self, sources, system_message="You are a helpful assistant.", add_end_for_empty_value=False):
        """reference :96-193 (Qwen-VL's own ChatML preprocessing): per conversation the ids / targets of the prompt part (system +
        user turn), of the answer part (assistant turn) and of the whole, targets masked with -100 except the assistant's words"""
        if not isinstance(sources, list) or not isinstance(sources[0], list):
            raise ValueError("sources must be a batch of conversations, eg. List[List[Dict]]")
        tok = self.tokenizer
        enc = lambda s: list(tok(s).input_ids)    # noqa: E731
        role_text = {"user": "<|im_start|>user", "assistant": "<|im_start|>assistant"}
        im_start, im_end, nl = tok.im_start_id, tok.im_end_id, enc("\n")
        sys_ids = [im_start] + enc("system") + nl + enc(system_message) + [im_end] + nl
        sys_tgt = [im_start] + [IGNORE_TOKEN_ID] * (len(sys_ids) - 3) + [im_end] + nl
        parts = {k: dict(input_ids=[], labels=[]) for k in ("prompt", "answer", "full")}
        raw_texts = []
        for source in sources:
            if source[0]["from"] != "user":
                source = source[1:]
            ids, tgt = list(sys_ids), list(sys_tgt)
            p_ids, p_tgt, a_ids, a_tgt = [], [], [], []
            raw = f"<|im_start|>system\n{system_message}<|im_end|>\n"
            for turn in source:
                if turn["from"] not in role_text:
                    raise NotImplementedError
                role = role_text[turn["from"]]
                r_ids = enc(role)
                has_text = turn["value"] != "" or add_end_for_empty_value
                t_ids = r_ids + nl + (enc(turn["value"]) + [im_end] + nl if has_text else [])
                raw += f"{role}\n" + (f"{turn['value']}<|im_end|>\n" if has_text else "")
                ids += t_ids
                if turn["from"] == "user":
                    t_tgt = [im_start] + [IGNORE_TOKEN_ID] * (len(t_ids) - 3) + [im_end] + nl if has_text else [im_start, IGNORE_TOKEN_ID]
                    p_ids += ids                       # the reference extends (not replaces): a second user turn repeats the prefix
                    p_tgt += tgt + t_tgt
                else:
                    t_tgt = [im_start] + [IGNORE_TOKEN_ID] * len(r_ids) + (t_ids[len(r_ids) + 1:-2] + [im_end] + nl if has_text else [])
                    a_ids += t_ids
                    a_tgt += t_tgt
                tgt += t_tgt
            assert len(ids) == len(tgt), f"{len(ids)} != {len(tgt)}"
            assert len(p_ids) == len(p_tgt) and len(a_ids) == len(a_tgt)
            for k, (i_, t_) in dict(prompt=(p_ids, p_tgt), answer=(a_ids, a_tgt), full=(ids, tgt)).items():
                parts[k]["input_ids"].append(i_)
                parts[k]["labels"].append(t_)
            raw_texts.append(raw)
        pad = tok.pad_token_id
        for d in parts.values():
            d["attention_mask"] = [[int(t != pad) for t in row] for row in d["input_ids"]]
        return {"prompt": parts["prompt"], "answer": parts["answer"], "full": parts["full"], "raw_str": raw_texts}

    @staticmethod
    def format_multimodal_prompt(prompt: str, img_paths: Optional[Union[List[str], str]] = None):
        if img_paths is None:
            return prompt
        if isinstance(img_paths, str):
            img_paths = [img_paths]
        if len(img_paths) == 1 and "<image>" not in prompt:
            return f"Picture 1: <img>{img_paths[0]}</img>\n{prompt}"
        assert prompt.count("<image>") == len(img_paths), \
            f"The number of given image ({len(img_paths)}) does not match the number of image placeholders in the prompt: {prompt}"
        for p in img_paths:
            prompt = prompt.replace("<image>", f"<img>{p}</img>\n", 1)
        return prompt

    @staticmethod
    def remove_image_placeholder(prompt: str):
        return re.sub(QwenVLProcessor._mm_pattern, "", prompt)

    @staticmethod
    def is_multimodal_prompt_valid(prompt: str):
        return bool(QwenVLProcessor._mm_pattern.search(prompt))

    def train(self):
        self.tokenizer.pad_token_id = self.tokenizer.eod_id
        self.tokenizer.eos_token_id = self.tokenizer.eod_id
        self.tokenizer.padding_side = "right"

    def infer(self):
        self.tokenizer.padding_side = "left"
        self.tokenizer.pad_token_id = self.tokenizer.eod_id


@dataclass
class QwenVLDPODataCollatorWithPadding(VLDPODataCollatorWithPadding):
    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        padded = super().__call__(features)
        size = getattr(self.processor, "image_size", None) or 448
        padded["img_input_dict"] = dict(pixel_values=load_qwen_pixel_values(flatten_list(padded["img_path"]), size))
        return padded


class QwenVLDPOTrainer(VLDPOTrainer):
    def tokenize_row(self, feature, model=None) -> Dict:
        """reference :256-347: Qwen-VL's own ChatML preprocessing instead of trl's prompt / prompt+answer split.  The labels of the
        answer come from process_batch_conv; an EOS (= <|endoftext|>) is appended; tokens equal to EOS inside the parts get attention 0;
        truncation as in trl (prompt first, by `truncation_mode`, then the responses)."""
        prompt = self.processor.format_multimodal_prompt(feature["prompt"], feature["img_path"])
        conv = {side: self.processor.process_batch_conv([self.processor.make_single_turn_conv(prompt, feature[side])])
                for side in ("chosen", "rejected")}
        eos = self.tokenizer.eos_token_id

        def part(d):
            row = {k: list(v[0]) for k, v in d.items()}
            row["attention_mask"] = [0 if t == eos else m for t, m in zip(row["input_ids"], row["attention_mask"])]
            return row
        toks = dict(prompt=part(conv["chosen"]["prompt"]), chosen=part(conv["chosen"]["answer"]), rejected=part(conv["rejected"]["answer"]))
        for side in ("chosen", "rejected"):
            toks[side]["input_ids"].append(eos)
            toks[side]["labels"].append(eos)
            toks[side]["attention_mask"].append(1)
        longer = max(len(toks["chosen"]["input_ids"]), len(toks["rejected"]["input_ids"]))
        if len(toks["prompt"]["input_ids"]) + longer > self.max_length:
            if self.truncation_mode == "keep_start":
                toks["prompt"] = {k: v[: self.max_prompt_length] for k, v in toks["prompt"].items()}
            elif self.truncation_mode == "keep_end":
                toks["prompt"] = {k: v[-self.max_prompt_length:] for k, v in toks["prompt"].items()}
            else:
                raise ValueError(f"Unknown truncation mode: {self.truncation_mode}")
        if len(toks["prompt"]["input_ids"]) + longer > self.max_length:
            for side in ("chosen", "rejected"):
                toks[side] = {k: v[: self.max_length - self.max_prompt_length] for k, v in toks[side].items()}
        n_prompt = len(toks["prompt"]["input_ids"])
        batch = {}
        for side in ("chosen", "rejected"):
            seq = {k: toks["prompt"][k] + toks[side][k] for k in toks[side]}
            seq["labels"][:n_prompt] = [self.label_pad_token_id] * n_prompt
            for k, v in seq.items():
                batch[f"{side}_{k}"] = v
        for k, v in toks["prompt"].items():
            batch[f"prompt_prompt_{k}"] = v       # sic: the reference prefixes the prompt fields twice (:336-346); kept for drop-in parity
        return batch


core_mapper = ModelCoreMapper(
    model=QwenVLForRL,
    processor=QwenVLProcessor,
    dpo_collator=QwenVLDPODataCollatorWithPadding,
    dpo_trainer=QwenVLDPOTrainer,
)
