"""The entry point end to end on the MI355X: an on-disk HF-layout checkpoint (config.json + safetensors + a real
LlamaTokenizerFast + CLIP image processor) -> `vlrlhf.dpo.main()` -> from_pretrained -> tokenize_row -> collator ->
PrefetchLoader (pinned memory + copy stream) -> HIP training steps -> saved output that loads again
(reference src/vlrlhf/dpo.py:98-149).  Also: checkpoint / resume reproduces the uninterrupted run bit for bit."""
import json
import os

import pytest
import torch

from tests.golden_util import write_tiny_checkpoint

pytestmark = pytest.mark.gpu


def _args(ckpt, out, **kw):
    a = dict(model_name_or_path=ckpt, dataset_name="synthetic", synthetic_rows=26, synthetic_image_size=28, output_dir=out,
             per_device_train_batch_size=2, max_steps=4, logging_steps=2, learning_rate=1e-4, max_length=96, max_prompt_length=48,
             warmup_ratio=0.0, lr_scheduler_type="constant", dataloader_prefetch=2, seed=3)
    a.update(kw)
    return [x for k, v in a.items() for x in (f"--{k}", str(v))]


def _flat(trainer):
    eng = trainer.model.engine
    eng.wait_optimizer()
    torch.cuda.synchronize()
    return (eng.lora_flat if eng.lora is not None else eng.policy.flat).clone()


def test_dpo_main_full_finetune_saves_a_loadable_model(tmp_path):
    from vlrlhf import dpo
    from vlrlhf.models.Llava import LlavaForRL
    ckpt, out = str(tmp_path / "ckpt"), str(tmp_path / "out")
    cfg, W = write_tiny_checkpoint(ckpt)
    tr = dpo.main(_args(ckpt, out, unknown_flag_xyz="1", eval_strategy="steps", eval_steps=2))
    hist = [h for h in tr.log_history if "loss" in h]
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["loss"])) for h in hist) and hist[-1]["step"] == 4
    assert any("eval_loss" in h for h in tr.log_history), "--eval_strategy steps must run evaluation (alias of evaluation_strategy)"
    assert {"rewards/chosen", "rewards/margins", "logps/chosen", "logits/chosen", "grad_norm"} <= set(hist[-1])
    trained = _flat(tr)
    w0 = LlavaForRL.from_pretrained(ckpt)
    assert not torch.equal(trained, w0.engine.policy.flat), "training did not change the weights"
    # the saved directory is a complete checkpoint: config + LLM + projector + the frozen vision tower + processor files
    for fn in ("config.json", "model.safetensors", "tokenizer.json", "trainer_state.json"):
        assert os.path.isfile(os.path.join(out, fn)), fn
    m2 = LlavaForRL.from_pretrained(out)
    assert torch.equal(m2.engine.policy.flat, trained)
    for k, v in m2.engine.vision_sd.items():
        assert torch.equal(v.cpu(), W[k]), k
    # and it computes: same batch through the reloaded model and the trained one
    from vlrlhf.utils.auto_load import MyAutoProcessor
    assert MyAutoProcessor.from_pretrained(out).tokenizer.convert_tokens_to_ids("<image>") == cfg["image_token"]


def test_checkpoint_resume_is_bit_exact(tmp_path):
    from vlrlhf import dpo
    ckpt = str(tmp_path / "ckpt")
    write_tiny_checkpoint(ckpt)
    a = dpo.main(_args(ckpt, str(tmp_path / "run_a"), max_steps=4, gradient_accumulation_steps=2))
    full = _flat(a)
    del a
    b1 = dpo.main(_args(ckpt, str(tmp_path / "run_b"), max_steps=2, gradient_accumulation_steps=2, save_strategy="steps", save_steps=1,
                        save_total_limit=1))
    del b1
    cks = sorted(d for d in os.listdir(tmp_path / "run_b") if d.startswith("checkpoint-"))
    assert cks == ["checkpoint-2"], cks                                  # save_total_limit rotated checkpoint-1 away
    st = json.load(open(tmp_path / "run_b" / "checkpoint-2" / "trainer_state.json"))
    assert st["global_step"] == 2 and st["micro_step"] == 4 and st["opt_step"] == 2
    b2 = dpo.main(_args(ckpt, str(tmp_path / "run_b"), max_steps=4, gradient_accumulation_steps=2, resume_from_checkpoint="true"))
    assert b2.state.global_step == 4
    assert torch.equal(_flat(b2), full), "resumed run differs from the uninterrupted one"
    with pytest.raises(ValueError, match="no checkpoint"):
        dpo.main(_args(ckpt, str(tmp_path / "empty"), resume_from_checkpoint="true"))


def test_lora_entrypoint_writes_peft_adapter_files(tmp_path):
    from vlrlhf import dpo
    ckpt, out = str(tmp_path / "ckpt"), str(tmp_path / "out")
    write_tiny_checkpoint(ckpt)
    tr = dpo.main(_args(ckpt, out, use_lora="true", lora_r=8, lora_alpha=16, lora_dropout=0.0, max_steps=2, logging_steps=1))
    cfg = json.load(open(os.path.join(out, "adapter_config.json")))
    allowed = {"peft_type", "task_type", "base_model_name_or_path", "r", "lora_alpha", "lora_dropout", "target_modules", "bias",
               "fan_in_fan_out", "inference_mode", "modules_to_save", "init_lora_weights"}
    assert set(cfg) <= allowed and cfg["r"] == 8 and cfg["peft_type"] == "LORA"       # LoraConfig(**cfg) fields only
    from safetensors.torch import load_file
    sd = load_file(os.path.join(out, "adapter_model.safetensors"))
    assert len(sd) == 2 * 7 * 2 and all(".lora_A.weight" in k or ".lora_B.weight" in k for k in sd)
    before = _flat(tr)
    tr.model.engine.lora_flat.zero_()
    tr.model.load_adapter(out)
    assert torch.equal(_flat(tr), before)


def test_small_dataset_and_partial_batches(tmp_path):
    """a shard smaller than the batch size still trains (HF keeps the partial batch); zero batches raise instead of spinning."""
    from vlrlhf import dpo
    ckpt = str(tmp_path / "ckpt")
    write_tiny_checkpoint(ckpt)
    tr = dpo.main(_args(ckpt, str(tmp_path / "o1"), synthetic_rows=4, per_device_train_batch_size=8, max_steps=2, logging_steps=1))
    assert tr.state.global_step == 2
    with pytest.raises(ValueError, match="yields no batch"):
        dpo.main(_args(ckpt, str(tmp_path / "o2"), synthetic_rows=4, per_device_train_batch_size=8, max_steps=2, dataloader_drop_last="true"))
