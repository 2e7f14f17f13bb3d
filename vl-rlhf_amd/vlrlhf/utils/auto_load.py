"""Architecture string -> model package dispatch (mirror of /root/reference/src/vlrlhf/utils/auto_load.py:41-93,
118-139, 190-308, 509-578 for the DPO path)."""
import json
import os
from importlib import import_module

MODEL_NICKNAME_MAP = {
    "LlavaForConditionalGeneration": "Llava",
    "LlavaNextForConditionalGeneration": "LlavaNext",
    "QWenLMHeadModel": "QwenVL",
    "InternLMXComposer2ForCausalLM": "InternLMXC2",
    "InstructBlipForConditionalGeneration": "InstructBlip",
    "LlavaForRL": "Llava",
}
FLASH_ATTN_MODELS = ["LlavaForConditionalGeneration", "LlavaNextForConditionalGeneration", "LlavaForRL"]
IMPLEMENTED = ["Llava"]


def _architecture(model_name_or_path):
    with open(os.path.join(model_name_or_path, "config.json")) as f:
        return json.load(f)["architectures"][0]


def auto_core_mapper(architecture: str):
    nick = MODEL_NICKNAME_MAP[architecture]
    if nick not in IMPLEMENTED:
        raise NotImplementedError(f"{nick}: only {IMPLEMENTED} have an MI355X DPO path so far (SURVEY.md 8f rank 4)")
    return import_module(f".{nick}", "vlrlhf.models").core_mapper


class MyAutoModel:
    @classmethod
    def from_pretrained(cls, model_name_or_path, *args, **kwargs):
        return auto_core_mapper(_architecture(model_name_or_path)).model.from_pretrained(model_name_or_path, *args, **kwargs)


class MyAutoProcessor:
    @classmethod
    def from_pretrained(cls, model_name_or_path, **kwargs):
        return auto_core_mapper(_architecture(model_name_or_path)).processor(model_name_or_path, **kwargs)


class MyAutoDPOCollator:
    def __new__(cls, model_name_or_path, pad_token_id=0, label_pad_token_id=-100, is_encoder_decoder=False, processor=None):
        return auto_core_mapper(_architecture(model_name_or_path)).dpo_collator(pad_token_id, label_pad_token_id,
                                                                                is_encoder_decoder, processor)


class MyAutoDPOTrainer:
    def __new__(cls, model_name_or_path, *args, **kwargs):
        return auto_core_mapper(_architecture(model_name_or_path)).dpo_trainer(*args, **kwargs)


def auto_load_rlmodel(script_args, training_args, lora_args):
    """-> (model, ref_model=None, lora_config=None); vision tower frozen (reference :554-555)."""
    if getattr(training_args, "use_lora", False):
        raise NotImplementedError("LoRA is SURVEY.md 8(f) rank 2; this round trains the full LLM + projector")
    model = MyAutoModel.from_pretrained(script_args.model_name_or_path)
    if getattr(script_args, "freeze_vision_tower", True):
        model.freeze_vision_tower()
    model.config.label_pad_token_id = script_args.label_pad_token_id
    model.config.use_cache = False
    return model, None, None
