"""DPO entry point - mirror of /root/reference/src/vlrlhf/dpo.py (ScriptArguments :16-56, LoraArguments :59-74,
TrainingArguments :77-86, main :98-149).  One process per GPU (torchrun / accelerate launch); gradients are reduced
with RCCL over xGMI by vlrlhf.parallel.GradReducer."""
import argparse
import os
from dataclasses import dataclass, field, fields
from typing import Optional


@dataclass
class ScriptArguments:
    beta: Optional[float] = 0.1
    score_margin: Optional[float] = -1
    data_path: Optional[str] = None
    data_ratio: Optional[float] = 1.0
    image_root: Optional[str] = None
    dataset_name: Optional[str] = "vlfeedback_paired"
    model_name_or_path: Optional[str] = "llava-hf/llava-1.5-7b-hf"
    max_length: Optional[int] = 512
    max_prompt_length: Optional[int] = 128
    max_target_length: Optional[int] = 128
    label_pad_token_id: Optional[int] = -100
    ignore_bias_buffers: Optional[bool] = False
    freeze_vision_tower: bool = True
    loss_type: str = "sigmoid"
    # not in the reference: shape of the `--dataset_name synthetic` rows (benchmarks / tests; no network for the hub datasets)
    synthetic_rows: int = 64
    synthetic_image_size: int = 336


@dataclass
class LoraArguments:
    lora_r: int = 64
    lora_alpha: int = 16
    lora_dropout: float = 0.05
    lora_target_modules: Optional[str] = None
    lora_bias: str = "none"
    q_lora: bool = False
    bits: int = 4
    modules_to_save: Optional[str] = None


@dataclass
class TrainingArguments:
    """the subset of transformers.TrainingArguments the DPO scripts set (scripts/dpo_llava.sh:16-61) + the reference's
    extra fields (:78-86)."""
    output_dir: str = "output"
    per_device_train_batch_size: int = 4
    gradient_accumulation_steps: int = 1
    learning_rate: float = 1e-6
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.98
    adam_epsilon: float = 1e-6
    max_grad_norm: float = 1.0
    warmup_ratio: float = 0.03
    warmup_steps: int = 0
    lr_scheduler_type: str = "cosine"
    num_train_epochs: float = 1.0
    max_steps: int = -1
    logging_steps: int = 10
    seed: int = 42
    bf16: bool = True
    tf32: bool = True
    gradient_checkpointing: bool = False
    use_lora: bool = False
    use_flash_attention_2: bool = True
    dataset_num_proc: Optional[int] = 16
    project_name: Optional[str] = "VL-RLHF"
    group_name: Optional[str] = "llava-1.5-7b-dpo"
    resume_from_checkpoint: Optional[bool] = None
    report_to: str = "none"
    run_name: str = "dpo"
    save_strategy: str = "no"
    save_steps: int = 500
    save_total_limit: int = 1
    evaluation_strategy: str = "no"
    eval_strategy: Optional[str] = None      # newer transformers spelling (scripts/dpo_llava.sh passes --eval_strategy steps)
    eval_steps: int = 500
    per_device_eval_batch_size: int = 4
    dataloader_num_workers: int = 0
    dataloader_drop_last: bool = False
    dataloader_prefetch: int = 2             # batches collated ahead on the background thread (base/loader.py); 0 = inline
    remove_unused_columns: bool = False
    local_rank: int = 0


def _parse(*classes, argv=None):
    p = argparse.ArgumentParser()
    for c in classes:
        for f in fields(c):
            t = f.type if f.type in (int, float, str) else None
            if f.type in (bool, Optional[bool]):
                p.add_argument(f"--{f.name}", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=f.default)
            else:
                base = {Optional[int]: int, Optional[float]: float, Optional[str]: str}.get(f.type, t or str)
                p.add_argument(f"--{f.name}", type=base, default=f.default)
    ns, unknown = p.parse_known_args(argv)
    if unknown:
        # HF's parser would raise; the reference scripts pass flags of subsystems that do not exist here (wandb, deepspeed,
        # gradient checkpointing knobs ...): say which ones are being ignored instead of swallowing them
        import sys
        print(f"[vlrlhf.dpo] WARNING: ignoring unsupported arguments: {' '.join(unknown)}", file=sys.stderr, flush=True)
    out = [c(**{f.name: getattr(ns, f.name) for f in fields(c)}) for c in classes]
    for o in out:
        if isinstance(o, TrainingArguments) and o.eval_strategy:
            o.evaluation_strategy = o.eval_strategy
    return out


def main(argv=None):
    from vlrlhf.parallel import init_distributed_from_env
    from vlrlhf.utils.auto_load import MyAutoDPOCollator, MyAutoDPOTrainer, MyAutoProcessor, auto_load_rlmodel
    from vlrlhf.utils.data import DATASET_MAP
    script_args, training_args, lora_args = _parse(ScriptArguments, TrainingArguments, LoraArguments, argv=argv)
    rank, local, world = init_distributed_from_env()
    training_args.local_rank = local
    model, ref_model, lora_config = auto_load_rlmodel(script_args, training_args, lora_args)
    # reference dpo.py:99 (gradient_checkpointing_kwargs use_reentrant=False; every shipped script passes --gradient_checkpointing True):
    # the engine keeps only the layer inputs and re-runs each layer's forward right before its backward (engine.hidden_backward)
    model.engine.gradient_checkpointing = bool(training_args.gradient_checkpointing)
    processor = MyAutoProcessor.from_pretrained(script_args.model_name_or_path)
    processor.train()
    dataset = DATASET_MAP[script_args.dataset_name](script_args)
    n_eval = max(1, int(len(dataset) * 0.005))
    import random
    idx = list(range(len(dataset)))
    random.Random(42).shuffle(idx)
    eval_dataset = [dataset[i] for i in idx[:n_eval]]
    train_dataset = [dataset[i] for i in idx[n_eval:]]
    train_dataset = train_dataset[: int(len(train_dataset) * script_args.data_ratio)]
    data_collator = MyAutoDPOCollator(script_args.model_name_or_path, pad_token_id=processor.tokenizer.pad_token_id,
                                      label_pad_token_id=script_args.label_pad_token_id,
                                      is_encoder_decoder=model.config.is_encoder_decoder, processor=processor)
    dpo_trainer = MyAutoDPOTrainer(
        script_args.model_name_or_path, model=model, args=training_args, beta=script_args.beta,
        train_dataset=train_dataset, eval_dataset=eval_dataset, processor=processor, max_length=script_args.max_length,
        max_target_length=script_args.max_target_length, max_prompt_length=script_args.max_prompt_length,
        generate_during_eval=False, label_pad_token_id=script_args.label_pad_token_id, data_collator=data_collator,
        peft_config=lora_config, loss_type=script_args.loss_type, ref_model=ref_model,
        dataset_num_proc=training_args.dataset_num_proc)
    dpo_trainer.use_dpo_data_collator = True
    if world > 1:
        model.engine.make_reducer()           # after the trainer: with peft_config only the adapters are reduced
    dpo_trainer.train(resume_from_checkpoint=training_args.resume_from_checkpoint)
    dpo_trainer.save_state()
    if rank == 0:
        # reference dpo.py:140-149: adapters only under LoRA (utils/common.py:97-98), else trainer.save_model -> the whole model
        if training_args.use_lora:
            model.save_adapter(training_args.output_dir, base_model_name_or_path=script_args.model_name_or_path)
        else:
            model.save_pretrained(training_args.output_dir)
        processor.save_pretrained(training_args.output_dir)
    return dpo_trainer


if __name__ == "__main__":
    main()
