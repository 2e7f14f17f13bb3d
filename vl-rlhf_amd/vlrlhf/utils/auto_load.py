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
    "LlavaNextForRL": "LlavaNext",
    "QWenLMHeadModel": "QwenVL",
    "QwenVLForRL": "QwenVL",
    "InternLMXC2ForRL": "InternLMXC2",
}
FLASH_ATTN_MODELS = ["LlavaForConditionalGeneration", "LlavaNextForConditionalGeneration", "LlavaForRL"]
IMPLEMENTED = ["Llava", "LlavaNext", "QwenVL", "InternLMXC2"]


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
    """-> (model, ref_model=None, lora_config); vision tower frozen (reference :554-555; freeze_vision_tower=False raises).  With use_lora the LoraConfig of
    reference :559-571 is returned as a plain dict (peft itself is not needed: the trainer hands it to
    LlavaForRL.apply_lora); q_lora (GPTQ, reference :520-548) is not on the MI355X path."""
    if getattr(training_args, "use_lora", False) and getattr(lora_args, "q_lora", False):
        raise NotImplementedError("q_lora (GPTQ 4-bit base weights) is outside the MI355X DPO path")
    if not getattr(script_args, "freeze_vision_tower", True):
        # reference :554-555 would train the tower (dpo.py:54 --freeze_vision_tower False); the MI355X path has no ViT backward:
        # refuse instead of silently training with a frozen tower
        raise NotImplementedError("--freeze_vision_tower False: the MI355X DPO path has no vision-tower backward; the tower is frozen "
                                  "in every shipped script of the reference (scripts/dpo_*.sh)")
    model = MyAutoModel.from_pretrained(script_args.model_name_or_path)
    model.freeze_vision_tower()
    model.config.label_pad_token_id = script_args.label_pad_token_id
    model.config.use_cache = False
    lora_config = None
    if getattr(training_args, "use_lora", False):
        targets = lora_args.lora_target_modules
        if targets in (None, "auto"):
            targets = model.default_lora_target
        elif isinstance(targets, str):
            targets = targets.split(",")
        lora_config = dict(r=lora_args.lora_r, lora_alpha=lora_args.lora_alpha, lora_dropout=lora_args.lora_dropout,
                           target_modules=list(targets), bias=lora_args.lora_bias, task_type="CAUSAL_LM",
                           modules_to_save=lora_args.modules_to_save, seed=int(getattr(training_args, "seed", 0)))
    return model, None, lora_config
