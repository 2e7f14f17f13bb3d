"""VLDPOTrainer for MI355X - mirror of /root/reference/src/vlrlhf/base/trainer.py:33-360 plus the parts of
trl==0.8.1 `DPOTrainer` and transformers `Trainer` the reference inherits on the DPO path (not vendored there:
constructor bookkeeping, tokenize_row / build_tokenized_answer, concatenated_inputs, compute_loss /
get_batch_loss_metrics, the step loop, clip + AdamW + cosine schedule).

Same names, argument order and error behaviour; the arithmetic (model forward/backward, log-probs, loss, optimizer)
runs in libvlr_hip.so through `vlrlhf.engine` - there is no PyTorch/CPU fallback.  What is deliberately NOT
reproduced: the per-micro-step `torch.cuda.empty_cache(); gc.collect()` (reference :306-307), ZeRO/DeepSpeed, wandb.
"""
import contextlib
import math
import os
import random
from collections import defaultdict
from contextlib import nullcontext
from typing import Any, Callable, Dict, List, Literal, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import _hip
from ..utils.common import pad_to_length
from ..utils.diff_lib import ddpo_shared_mask

LOSS_TYPE_IDS = {"sigmoid": 0, "ddpo": 0, "hinge": 1, "ipo": 2, "kto_pair": 3}


# ----------------------------------------------------------------------------------------------------------
# autograd boundary: three coarse Functions, each a thin shell around C-ABI calls
# ----------------------------------------------------------------------------------------------------------
class _TensorLogpsFn(torch.autograd.Function):
    """get_batch_logps on a MATERIALISED logits tensor [Bn,S,V] (the reference's A4 op): shift, log-softmax pick,
    masked sum - HIP kernels vlr_build_rows / vlr_logp_rows / vlr_seq_sum, backward vlr_dlogits_rows."""

    @staticmethod
    def forward(ctx, logits, labels, shared_mask, average, label_pad):
        Bn, S, V = logits.shape
        dev = logits.device
        lg = logits.detach().float().contiguous().view(Bn * S, V)
        rows = torch.empty(Bn * S, dtype=torch.int32, device=dev)
        tgt = torch.empty(Bn * S, dtype=torch.int32, device=dev)
        seq_off = torch.empty(Bn + 1, dtype=torch.int32, device=dev)
        sm = shared_mask.to(device=dev, dtype=torch.uint8).contiguous() if shared_mask is not None else None
        _hip.call("vlr_build_rows", labels.to(dev).contiguous(), sm, Bn, S, label_pad, rows, tgt, seq_off)
        R = int(seq_off[-1])
        out = torch.zeros(Bn, dtype=torch.float32, device=dev)
        tok = torch.empty(max(R, 1), dtype=torch.float32, device=dev)
        lse = torch.empty(max(R, 1), dtype=torch.float32, device=dev)
        if R:
            _hip.call("vlr_logp_rows", lg, rows, tgt, R, V, V, tok, lse)
            _hip.call("vlr_seq_sum", tok, seq_off, Bn, int(average), out)
        ctx.save_for_backward(lg, rows, tgt, seq_off, lse)
        ctx.meta = (Bn, S, V, R, int(average), logits.dtype)
        return out

    @staticmethod
    def backward(ctx, dlogps):
        lg, rows, tgt, seq_off, lse = ctx.saved_tensors
        Bn, S, V, R, average, dtype = ctx.meta
        grad = torch.zeros(Bn * S, V, dtype=dtype, device=lg.device)
        if R:
            compact = lg[rows[:R].long()].contiguous()
            dl = torch.empty(R, V, dtype=torch.bfloat16, device=lg.device)
            _hip.call("vlr_dlogits_rows", compact, tgt, lse, seq_off, Bn, dlogps.float().contiguous(), average, R, V, V, dl, V)
            grad[rows[:R].long()] = dl.to(dtype)
        return grad.view(Bn, S, V), None, None, None, None


class _DpoLossFn(torch.autograd.Function):
    """VLDPOTrainer.dpo_loss forward + backward in one HIP kernel (vlr_dpo_loss)."""

    @staticmethod
    def forward(ctx, pc, pr, rc, rr, beta, label_smoothing, loss_type_id, reference_free):
        n = pc.shape[0]
        dev = pc.device
        a = [t.detach().float().contiguous() for t in (pc, pr, rc, rr)]
        nl = 2 * n if loss_type_id == 3 else n
        losses = torch.empty(nl, dtype=torch.float32, device=dev)
        cr, rw, dpc, dpr = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
        mean = torch.empty(1, dtype=torch.float32, device=dev)
        _hip.call("vlr_dpo_loss", a[0], a[1], a[2], a[3], n, float(beta), float(label_smoothing), int(loss_type_id),
                  int(reference_free), losses, cr, rw, dpc, dpr, mean, None)
        ctx.save_for_backward(*a)
        ctx.meta = (n, float(beta), float(label_smoothing), int(loss_type_id), int(reference_free))
        ctx.mark_non_differentiable(cr, rw)
        return losses, cr, rw

    @staticmethod
    def backward(ctx, g_losses, g_cr, g_rw):
        a = ctx.saved_tensors
        n, beta, ls, lt, rf = ctx.meta
        dev = a[0].device
        nl = 2 * n if lt == 3 else n
        scratch = [torch.empty(max(nl, n), dtype=torch.float32, device=dev) for _ in range(3)]
        dpc, dpr = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
        mean = torch.empty(1, dtype=torch.float32, device=dev)
        _hip.call("vlr_dpo_loss", a[0], a[1], a[2], a[3], n, beta, ls, lt, rf, scratch[0], scratch[1], scratch[2], dpc, dpr,
                  mean, g_losses.float().contiguous())
        return dpc, dpr, None, None, None, None, None, None


class VLDPOTrainer:
    def __init__(
        self,
        model=None,
        ref_model=None,
        beta: float = 0.1,
        label_smoothing: float = 0,
        loss_type: Literal["sigmoid", "hinge", "ipo", "kto_pair", "ddpo"] = "sigmoid",
        args=None,
        data_collator: Any = None,
        label_pad_token_id: int = -100,
        padding_value: int = None,
        truncation_mode: str = "keep_end",
        train_dataset=None,
        eval_dataset=None,
        processor=None,
        model_init: Optional[Callable] = None,
        callbacks: Optional[List] = None,
        optimizers: Tuple = (None, None),
        preprocess_logits_for_metrics: Optional[Callable] = None,
        max_length: Optional[int] = None,
        max_prompt_length: Optional[int] = None,
        max_target_length: Optional[int] = None,
        peft_config: Optional[Dict] = None,
        is_encoder_decoder: Optional[bool] = None,
        disable_dropout: bool = True,
        generate_during_eval: bool = False,
        compute_metrics: Optional[Callable] = None,
        precompute_ref_log_probs: bool = False,
        dataset_num_proc: Optional[int] = None,
        model_init_kwargs: Optional[Dict] = None,
        ref_model_init_kwargs: Optional[Dict] = None,
        model_adapter_name: Optional[str] = None,
        ref_adapter_name: Optional[str] = None,
        reference_free: bool = False,
    ):
        # argument order = reference base/trainer.py:34-68 (MyAutoDPOTrainer passes them positionally)
        if model is None:
            raise ValueError("VLDPOTrainer needs a model")
        if peft_config is not None:
            # trl==0.8.1 DPOTrainer.__init__: model = get_peft_model(model, peft_config); with a peft policy and no
            # ref_model the reference pass runs the policy with its adapters disabled (null_ref_context)
            if not hasattr(model, "apply_lora"):
                raise ValueError("peft_config given but the model wrapper has no apply_lora()")
            model.apply_lora(peft_config)
        self.is_peft_model = bool(getattr(model, "is_peft_model", False))
        self.processor = processor
        self.tokenizer = processor.tokenizer if processor is not None else None
        self.model = model
        self.beta = beta
        self.label_smoothing = label_smoothing
        self.loss_type = loss_type
        self.args = args
        self.label_pad_token_id = label_pad_token_id
        self.padding_value = padding_value if padding_value is not None else (
            self.tokenizer.pad_token_id if self.tokenizer is not None and self.tokenizer.pad_token_id is not None else 0)
        self.truncation_mode = truncation_mode
        self.max_length = max_length if max_length is not None else 512
        self.max_prompt_length = max_prompt_length if max_prompt_length is not None else 128
        self.max_target_length = max_target_length
        self.is_encoder_decoder = bool(is_encoder_decoder) if is_encoder_decoder is not None else bool(
            getattr(getattr(model, "config", None), "is_encoder_decoder", False))
        if self.is_encoder_decoder:
            raise NotImplementedError("encoder-decoder models are not on the MI355X DPO path")
        self.generate_during_eval = generate_during_eval
        self.precompute_ref_log_probs = precompute_ref_log_probs
        self._precomputed_train_ref_log_probs = False
        self._precomputed_eval_ref_log_probs = False
        self.reference_free = reference_free
        self.dataset_num_proc = dataset_num_proc
        self.callbacks = list(callbacks or [])
        self.use_dpo_data_collator = True
        self._stored_metrics = defaultdict(lambda: defaultdict(list))
        self.log_history: List[dict] = []
        self.accelerator = _Accelerator(model)
        # trl: ref_model None and no peft -> frozen deep copy of the policy
        if ref_model is None and not reference_free and not precompute_ref_log_probs and not self.is_peft_model:
            ref_model = model.create_reference_model() if hasattr(model, "create_reference_model") else None
        self.ref_model = ref_model
        self.data_collator = data_collator
        self.train_dataset = self._tokenize_dataset(train_dataset)
        self.eval_dataset = self._tokenize_dataset(eval_dataset)
        self.ref_on_side_stream = True
        self._ref_stream = None
        # VLR_REF_PIPELINE=1: reference forward of the NEXT batch issued between the backward and the optimizer step of the
        # current one (prefetch_reference), so the HBM-bound clip + AdamW pass runs under the frozen forward.  Bit-identical
        # results, but OFF by default: measured +6 ms per step at N=1 (638.0 vs 632.1 ms, same box) - the
        # persistent GEMM workgroups fill the register file of every CU, so AdamW blocks do not co-reside with them, they
        # displace them, and the persistent tile walk pays a tail for every displaced workgroup.
        self.ref_pipeline = os.environ.get("VLR_REF_PIPELINE", "0") == "1"
        self._ref_pending = None
        self.state = _State()

    # ------------------------------------------------------------------------------------------ tokenisation
    def _tokenize_dataset(self, ds):
        if ds is None:
            return None
        rows = list(ds)
        if rows and "chosen_input_ids" not in rows[0]:
            rows = [self.tokenize_row(dict(r)) for r in rows]
        return rows

    def build_tokenized_answer(self, prompt: str, answer: str) -> Dict:
        """trl==0.8.1 DPOTrainer.build_tokenized_answer: tokenize prompt+answer jointly and split so that
        enc(prompt) + enc(answer) == enc(prompt + answer) even when the tokenizer merges across the boundary."""
        full = self.tokenizer(prompt + answer, add_special_tokens=False)
        prompt_ids = self.tokenizer(prompt, add_special_tokens=False)["input_ids"]
        answer_ids = full["input_ids"][len(prompt_ids):]
        answer_mask = full["attention_mask"][len(prompt_ids):]
        if len(full["input_ids"]) != len(prompt_ids + answer_ids):
            raise ValueError("Prompt input ids and answer input ids should have the same length.")
        start = len(prompt_ids)
        if prompt_ids != full["input_ids"][:start]:
            start -= 1
        p_ids, p_mask = full["input_ids"][:start], full["attention_mask"][:start]
        if len(p_ids) != len(p_mask):
            raise ValueError("Prompt input ids and attention mask should have the same length.")
        return dict(prompt_input_ids=p_ids, prompt_attention_mask=p_mask,
                    input_ids=full["input_ids"][start:], attention_mask=full["attention_mask"][start:] if answer_mask is not None else None)

    def _trl_tokenize_row(self, feature) -> Dict:
        """trl==0.8.1 DPOTrainer.tokenize_row, decoder-only branch (the reference restates the same truncation / label
        logic in-repo at models/QwenVL/__init__.py:281-347)."""
        prompt, chosen, rejected = feature["prompt"], feature["chosen"], feature["rejected"]
        if not isinstance(prompt, str):
            raise ValueError(f"prompt should be an str but got {type(prompt)}")
        tok = self.tokenizer
        prompt_tokens = {f"prompt_{k}": v for k, v in tok(prompt, add_special_tokens=False).items()}
        if not isinstance(chosen, str):
            raise ValueError(f"chosen should be an str but got {type(chosen)}")
        chosen_tokens = self.build_tokenized_answer(prompt, chosen)
        if not isinstance(rejected, str):
            raise ValueError(f"rejected should be an str but got {type(rejected)}")
        rejected_tokens = self.build_tokenized_answer(prompt, rejected)
        c_len, r_len = len(chosen_tokens["prompt_input_ids"]), len(rejected_tokens["prompt_input_ids"])
        p_len = min(c_len, r_len)
        for k in ("prompt_input_ids", "prompt_attention_mask"):
            prompt_tokens[k] = prompt_tokens[k][:p_len]
        ndiff = sum(a != b for a, b in zip(chosen_tokens["prompt_input_ids"], rejected_tokens["prompt_input_ids"]))
        if ndiff > 1 or abs(c_len - r_len) > 1:
            raise ValueError("Chosen and rejected prompt_input_ids might only differ on the last token due to tokenizer "
                             "merge ops.")
        bos, eos = tok.bos_token_id, tok.eos_token_id
        for t in (prompt_tokens, chosen_tokens, rejected_tokens):
            t["prompt_input_ids"] = [bos] + list(t["prompt_input_ids"])
            t["prompt_attention_mask"] = [1] + list(t["prompt_attention_mask"])
        for t in (chosen_tokens, rejected_tokens):
            t["input_ids"] = list(t["input_ids"]) + [eos]
            t["attention_mask"] = list(t["attention_mask"]) + [1]
        longer = max(len(chosen_tokens["input_ids"]), len(rejected_tokens["input_ids"]))
        for t in (chosen_tokens, rejected_tokens, prompt_tokens):
            if len(t["prompt_input_ids"]) + longer > self.max_length:
                if self.truncation_mode == "keep_start":
                    for k in ("prompt_input_ids", "prompt_attention_mask"):
                        t[k] = t[k][: self.max_prompt_length]
                elif self.truncation_mode == "keep_end":
                    for k in ("prompt_input_ids", "prompt_attention_mask"):
                        t[k] = t[k][-self.max_prompt_length:]
                else:
                    raise ValueError(f"Unknown truncation mode: {self.truncation_mode}")
        for t in (chosen_tokens, rejected_tokens):
            if len(t["prompt_input_ids"]) + longer > self.max_length:
                for k in ("input_ids", "attention_mask"):
                    t[k] = t[k][: self.max_length - self.max_prompt_length]
        batch = {}
        for name, t in (("chosen_", chosen_tokens), ("rejected_", rejected_tokens)):
            seq_ids = t["prompt_input_ids"] + t["input_ids"]
            seq_mask = t["prompt_attention_mask"] + t["attention_mask"]
            labels = list(seq_ids)
            n = len(t["prompt_input_ids"])
            labels[:n] = [self.label_pad_token_id] * n
            batch[name + "input_ids"], batch[name + "attention_mask"], batch[name + "labels"] = seq_ids, seq_mask, labels
        batch["prompt_input_ids"] = prompt_tokens["prompt_input_ids"]
        batch["prompt_attention_mask"] = prompt_tokens["prompt_attention_mask"]
        return batch

    def tokenize_row(self, feature, model=None) -> Dict:
        """reference base/trainer.py:105-122."""
        prompt = self.processor.format_multimodal_prompt(feature["prompt"], feature["img_path"])
        conv = self.processor.make_single_turn_conv(prompt, "")
        raw = self.processor.process_batch_conv([conv], system_message=None, add_end_for_empty_value=False)["raw_str"][0]
        end = self.processor.chat_template.assistant_end
        feature = dict(feature)
        feature["chosen"] += end
        feature["rejected"] += end
        feature["prompt"] = raw
        batch = self._trl_tokenize_row(feature)
        batch["img_path"] = feature["img_path"]
        for k in ("reference_chosen_logps", "reference_rejected_logps"):
            if k in feature:
                batch[k] = feature[k]
        return batch

    # ------------------------------------------------------------------------------------------ batching
    def concatenated_inputs(self, batch, is_encoder_decoder: bool = False, label_pad_token_id: int = -100,
                            padding_value: int = 0, device=None) -> Dict[str, torch.Tensor]:
        """reference base/trainer.py:124-146 (+ trl static base): chosen over rejected, padded to the common length;
        every image tensor / list is duplicated on the batch dimension.  The duplicate carries `_vlr_dup = 2` so the
        frozen vision tower is evaluated once per distinct image (results identical), and the result is memoised on
        the batch so the policy pass and the reference pass share the same device tensors."""
        cache = batch.get("_vlr_concat") if isinstance(batch, dict) else None
        src_ids = tuple(id(batch.get(k)) for k in ("chosen_input_ids", "rejected_input_ids", "chosen_labels", "rejected_labels",
                                                    "chosen_attention_mask", "rejected_attention_mask", "img_input_dict"))
        if cache is not None and cache.get("_vlr_src") == src_ids:     # a dict(batch) copy with replaced tensors must not hit
            return {k: v for k, v in cache.items() if k != "_vlr_src"}
        out = {}
        n = max(batch["chosen_input_ids"].shape[1], batch["rejected_input_ids"].shape[1])
        for field, pad in (("input_ids", padding_value), ("attention_mask", 0), ("labels", label_pad_token_id)):
            parts = [pad_to_length(batch[f"{s}_{field}"], n, pad) for s in ("chosen", "rejected")]
            t = torch.cat(parts, dim=0)
            out[f"concatenated_{field}"] = t.to(device) if device is not None else t
        # integer facts about THIS batch that every pass over it needs on the host (merged length, lm-head row count, DDPO
        # mask ...): the first pass computes them (one D2H each), later passes - the policy pass after the reference pass, the
        # next visit of a resident batch - read them here instead of stalling the device again
        out["concatenated_input_ids"]._vlr_meta = {}
        if "img_input_dict" in batch:
            d = {}
            for k, v in batch["img_input_dict"].items():
                if isinstance(v, torch.Tensor):
                    t = torch.cat([v, v], dim=0)
                    t = t.to(device) if device is not None else t
                    t._vlr_dup = 2
                    d[k] = t
                elif isinstance(v, list):
                    d[k] = v + v
                else:
                    raise ValueError(f"Unsupported type {type(v)} for concatenation.")
            out["concatenated_img_input_dict"] = d
        if isinstance(batch, dict):
            batch["_vlr_concat"] = dict(out, _vlr_src=src_ids)
        return out

    @staticmethod
    def get_batch_logps(logits, labels, average_log_prob: bool = False, label_pad_token_id: int = -100,
                        is_encoder_decoder: bool = False, mask_shared_tokens: bool = False) -> torch.Tensor:
        """reference base/trainer.py:148-188.  `logits` is either the lazy lm-head handle the MI355X model wrappers
        return (fused path: the [2B,S,V] tensor is never materialised) or a real [2B,S,V] tensor on the GPU."""
        if tuple(logits.shape[:-1]) != tuple(labels.shape):
            raise ValueError("Logits (batch and sequence length dim) and labels must have the same shape.")
        if is_encoder_decoder:
            raise NotImplementedError("encoder-decoder models are not on the MI355X DPO path")
        shared = None
        if mask_shared_tokens:
            assert labels.shape[0] % 2 == 0
            meta = getattr(labels, "_vlr_meta", None)
            shared = meta.get("ddpo_mask") if meta is not None else None
            if shared is None:
                shared = ddpo_shared_mask(labels, label_pad_token_id, min_match_size=3)
                if meta is not None:
                    meta["ddpo_mask"] = shared.to(labels.device) if labels.is_cuda else shared
        if hasattr(logits, "batch_logps"):
            return logits.batch_logps(labels, shared, average_log_prob, label_pad_token_id)
        if not logits.is_cuda:
            raise _hip.VlrError("get_batch_logps: logits must live on the MI355X (no CPU fallback)")
        return _TensorLogpsFn.apply(logits, labels, shared, bool(average_log_prob), int(label_pad_token_id))

    def concatenated_forward(self, model, batch):
        """reference base/trainer.py:190-242."""
        cb = self.concatenated_inputs(batch, is_encoder_decoder=self.is_encoder_decoder,
                                      label_pad_token_id=self.label_pad_token_id, padding_value=self.padding_value,
                                      device=self.accelerator.device)
        len_chosen = batch["chosen_labels"].shape[0]
        kwargs = {"use_cache": False}
        if "concatenated_img_input_dict" in cb:
            kwargs.update(cb["concatenated_img_input_dict"])
        output = model(input_ids=cb["concatenated_input_ids"], attention_mask=cb["concatenated_attention_mask"],
                       labels=cb["concatenated_labels"], **kwargs)
        all_logits = output.logits
        final_labels = output.labels if getattr(output, "labels", None) is not None else cb["concatenated_labels"]
        all_logps = self.get_batch_logps(all_logits, final_labels, average_log_prob=False,
                                         is_encoder_decoder=self.is_encoder_decoder,
                                         label_pad_token_id=self.label_pad_token_id,
                                         mask_shared_tokens=self.loss_type == "ddpo")
        return (all_logps[:len_chosen], all_logps[len_chosen:], all_logits[:len_chosen], all_logits[len_chosen:])

    def dpo_loss(self, policy_chosen_logps, policy_rejected_logps, reference_chosen_logps, reference_rejected_logps):
        """reference base/trainer.py:244-301 -> (losses, chosen_rewards, rejected_rewards)."""
        if self.loss_type not in LOSS_TYPE_IDS:
            raise ValueError(
                f"Unknown loss type: {self.loss_type}. Should be one of ['sigmoid', 'hinge', 'ipo', 'kto_pair']")
        dev = self.accelerator.device
        return _DpoLossFn.apply(policy_chosen_logps.to(dev), policy_rejected_logps.to(dev),
                                reference_chosen_logps.to(dev), reference_rejected_logps.to(dev), self.beta,
                                self.label_smoothing, LOSS_TYPE_IDS[self.loss_type], self.reference_free)

    # ------------------------------------------------------------------------------------------ loss + metrics
    def _reference_logps(self, batch):
        if "reference_chosen_logps" in batch and "reference_rejected_logps" in batch:
            dev = self.accelerator.device
            return batch["reference_chosen_logps"].to(dev).float(), batch["reference_rejected_logps"].to(dev).float()
        if self.ref_model is None:
            if not self.is_peft_model:
                raise ValueError("no reference model and no precomputed reference log-probs in the batch")
            with torch.no_grad(), self.null_ref_context():
                rc, rr, _, _ = self.concatenated_forward(self.model, batch)
            return rc, rr
        with torch.no_grad():
            rc, rr, _, _ = self.concatenated_forward(self.ref_model, batch)
        return rc, rr

    def null_ref_context(self):
        """trl==0.8.1 DPOTrainer.null_ref_context: the peft policy with its adapters disabled is the reference model."""
        return self.model.disable_adapter() if self.is_peft_model else contextlib.nullcontext()

    def _ref_owner(self):
        return self.ref_model if self.ref_model is not None else (self.model if self.is_peft_model else None)

    def _ref_side_ok(self, batch) -> bool:
        return bool(self.ref_on_side_stream and self._ref_owner() is not None and "reference_chosen_logps" not in batch
                    and torch.cuda.is_available())

    def _launch_reference(self, batch):
        """the frozen reference forward on the side HIP stream, ordered after everything queued on the current stream so far
        (the shared ViT features are evaluated there first)"""
        main = torch.cuda.current_stream()
        ref_owner = self._ref_owner()
        cb = self.concatenated_inputs(batch, False, self.label_pad_token_id, self.padding_value, self.accelerator.device)
        if "concatenated_img_input_dict" in cb and hasattr(ref_owner, "prefetch_vision"):
            ref_owner.prefetch_vision(cb["concatenated_img_input_dict"])
        if self._ref_stream is None:
            self._ref_stream = torch.cuda.Stream()
        self._ref_stream.wait_stream(main)
        with torch.cuda.stream(self._ref_stream):
            return self._reference_logps(batch)

    def prefetch_reference(self, inputs):
        """Issue the reference forward of the NEXT batch now.  Called by the training loop (and bench.py) after the backward of
        the current batch is queued and BEFORE the optimizer step: the reference model is frozen (under LoRA: the base weights
        with the adapters disabled), so its log-probs do not depend on the update, and the HBM-bound gradient-norm + AdamW pass
        (and, N > 1, the exposed tail of the gradient all-reduce) overlaps the compute-bound forward instead of idling the MFMA
        units.  Returns the device-resident batch to hand to training_step(); a batch that is never trained on just drops its
        result.  Work per optimizer step is unchanged: one reference forward, one policy forward + backward, one update."""
        inputs = self._prepare_inputs(inputs)
        if self.ref_pipeline and self._ref_side_ok(inputs) and "chosen_input_ids" in inputs:
            rc, rr = self._launch_reference(inputs)
            self._ref_pending = (inputs["chosen_input_ids"], rc, rr)
        return inputs

    def get_batch_loss_metrics(self, model, batch, train_eval: Literal["train", "eval"] = "train"):
        """trl==0.8.1 DPOTrainer.get_batch_loss_metrics.  The reference forward is issued on a side HIP stream ahead of
        the policy forward (it is frozen and shares only the cached vision features), then joined before the loss."""
        main = torch.cuda.current_stream()
        pending, use_side = self._ref_pending, False
        if pending is not None and pending[0] is batch.get("chosen_input_ids"):
            rc, rr = pending[1], pending[2]             # issued by prefetch_reference during the previous step
            self._ref_pending = None
            use_side = True
        elif self._ref_side_ok(batch):
            rc, rr = self._launch_reference(batch)
            use_side = True
        pc, pr, pcl, prl = self.concatenated_forward(model, batch)
        if use_side:
            main.wait_stream(self._ref_stream)
            rc.record_stream(main)
            rr.record_stream(main)
        else:
            rc, rr = self._reference_logps(batch)
        losses, chosen_rewards, rejected_rewards = self.dpo_loss(pc, pr, rc, rr)
        reward_accuracies = (chosen_rewards > rejected_rewards).float()
        prefix = "eval_" if train_eval == "eval" else ""
        metrics = {
            f"{prefix}rewards/chosen": chosen_rewards.mean(),
            f"{prefix}rewards/rejected": rejected_rewards.mean(),
            f"{prefix}rewards/accuracies": reward_accuracies.mean(),
            f"{prefix}rewards/margins": (chosen_rewards - rejected_rewards).mean(),
            f"{prefix}logps/rejected": pr.detach().mean(),
            f"{prefix}logps/chosen": pc.detach().mean(),
            f"{prefix}logits/rejected": prl.detach().mean(),
            f"{prefix}logits/chosen": pcl.detach().mean(),
        }
        return losses.mean(), metrics

    def compute_loss(self, model, inputs, return_outputs: bool = False):
        loss, metrics = self.get_batch_loss_metrics(model, inputs, train_eval="train")
        self.store_metrics(metrics, train_eval="train")
        if return_outputs:
            return loss, metrics
        return loss

    def store_metrics(self, metrics, train_eval="train"):
        for k, v in metrics.items():
            self._stored_metrics[train_eval][k].append(v)

    def log(self, logs: Dict[str, float]):
        """trl DPOTrainer.log + HF Trainer.log: the stored metrics are averaged over the micro-steps since the last log and -
        like the loss HF reports (`_nested_gather(tr_loss).mean()`) - over the data-parallel ranks, with ONE all-reduce of
        the <= 9 scalars.  Every rank must call log() at the same steps (they do: same step counter)."""
        from ..parallel import all_reduce_mean_scalars
        train_eval = "train" if "loss" in logs else "eval"
        keys, vals = [], []
        for k, v in self._stored_metrics[train_eval].items():
            keys.append(k)
            vals.append(torch.stack([torch.as_tensor(x, dtype=torch.float32, device=self.accelerator.device) for x in v]).mean())
        for k in ("loss", "eval_loss"):
            if k in logs:
                keys.append(k)
                vals.append(float(logs[k]))
        self._stored_metrics[train_eval].clear()
        if keys:
            logs.update(zip(keys, all_reduce_mean_scalars(vals, device=self.accelerator.device if _world() > 1 and _backend() == "nccl" else None)))
        logs = dict(logs, step=self.state.global_step)
        self.log_history.append(logs)
        if getattr(self.args, "local_rank", 0) in (0, -1) and _rank() == 0:
            print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in logs.items()}, flush=True)
        return logs

    # ------------------------------------------------------------------------------------------ step + loop
    def _prepare_inputs(self, inputs):
        dev = self.accelerator.device

        def mv(v):
            if isinstance(v, torch.Tensor):
                return v.to(dev, non_blocking=True)
            if isinstance(v, dict):
                return {k: mv(x) for k, x in v.items()}
            return v
        return {k: mv(v) for k, v in inputs.items()}

    def training_step(self, model, inputs) -> torch.Tensor:
        """reference base/trainer.py:303-308 + transformers Trainer.training_step: H2D, loss, backward, return
        loss.detach() / gradient_accumulation_steps.  (No empty_cache / gc.collect: that flush serialises the device.)"""
        model.train()
        inputs = self._prepare_inputs(inputs)
        loss = self.compute_loss(model, inputs)
        ga = max(1, int(getattr(self.args, "gradient_accumulation_steps", 1) or 1))
        loss.backward()
        return loss.detach() / ga

    # ------------------------------------------------------------------------------------------ evaluation
    def prediction_step(self, model, inputs, prediction_loss_only: bool = True, ignore_keys=None):
        """trl==0.8.1 DPOTrainer.prediction_step: no-grad loss + metrics of one batch, stored under the eval_ prefix;
        returns (loss, logits, labels) with logits = [eval_logits/chosen, eval_logits/rejected] as trl does."""
        with torch.no_grad():
            loss, metrics = self.get_batch_loss_metrics(model, self._prepare_inputs(inputs), train_eval="eval")
        self.store_metrics(metrics, train_eval="eval")
        if prediction_loss_only:
            return loss.detach(), None, None
        logits = torch.stack([metrics["eval_logits/chosen"], metrics["eval_logits/rejected"]]).mean(dim=0, keepdim=True)
        return loss.detach(), logits, torch.zeros(logits.shape[0], device=logits.device)

    def evaluate(self, eval_dataset=None, metric_key_prefix: str = "eval") -> Dict[str, float]:
        """HF Trainer.evaluate on the DPO objective: mean eval loss + the eight eval_ metrics over the (rank-sharded) eval set."""
        ds = eval_dataset if eval_dataset is not None else self.eval_dataset
        if not ds:
            return {}
        if self.precompute_ref_log_probs and not self._precomputed_eval_ref_log_probs and eval_dataset is None:
            self.precompute_reference_log_probs(ds)
            self._precomputed_eval_ref_log_probs = True
        bs = int(getattr(self.args, "per_device_eval_batch_size", None) or getattr(self.args, "per_device_train_batch_size", 4))
        # the index list is padded by wrapping to a multiple of the world size (as _train_row_batches does): every rank runs the same
        # number of prediction steps and reduces the same 9 scalars in log() - an eval set smaller than the world would otherwise
        # leave ranks without metrics and hang the collective (torch DistributedSampler semantics, as HF's eval dataloader)
        rows = list(ds)
        w = _world()
        if w > 1 and len(rows) % w:
            pad = (-len(rows)) % w                      # cycled: an eval set smaller than the padding still fills every rank
            rows = rows + (rows * (pad // len(rows) + 1))[:pad]
        rows = rows[_rank()::w]
        was_training = self.model.training
        self.model.eval()
        if self.generate_during_eval and rows:
            # trl==0.8.1 DPOTrainer.evaluation_loop: ONE random eval batch is sampled from the policy and the reference and logged as a
            # (prompt, policy, reference) table (wandb.Table there; a "game_log" entry of log_history + a rank-0 print here).  Every rank
            # draws its own batch from its shard - no collective is involved.
            import random
            pick = random.sample(range(len(rows)), k=min(bs, len(rows)))
            sample = self._prepare_inputs(self.data_collator([rows[i] for i in pick]))
            policy_txt, ref_txt = self.get_batch_samples(self.model, sample)
            prompts = sample.get("prompt") or self.tokenizer.batch_decode(sample["prompt_input_ids"], skip_special_tokens=True)
            table = [[pr, po[len(pr):], rf[len(pr):]] for pr, po, rf in zip(prompts, policy_txt, ref_txt)]
            if _rank() == 0:
                self.log_history.append({"game_log": {"columns": ["Prompt", "Policy", "Ref Model"], "rows": table}, "step": self.state.global_step})
                for row in table:
                    print({"prompt": row[0], "policy": row[1], "ref_model": row[2]}, flush=True)
        losses = []
        for i in range(0, len(rows), bs):
            loss, _, _ = self.prediction_step(self.model, self.data_collator(rows[i:i + bs]))
            losses.append(loss)
        self.model.train(was_training)
        out = {f"{metric_key_prefix}_loss": float(torch.stack(losses).mean()) if losses else float("nan")}
        return self.log(out)

    # ------------------------------------------------------------------------------------------ reference pre-pass
    def compute_reference_log_probs(self, padded_batch: Dict) -> Tuple[torch.Tensor, torch.Tensor]:
        """trl==0.8.1 DPOTrainer.compute_reference_log_probs: reference log-probs of one collated batch, no grad.  Without
        a ref_model the policy itself is the reference (its adapters disabled if it is a peft model): the pre-pass runs
        before the first optimizer step, so these are the initial weights."""
        with torch.no_grad():
            if self.ref_model is None:
                with self.null_ref_context():
                    was_training = self.model.training
                    self.model.eval()
                    rc, rr, _, _ = self.concatenated_forward(self.model, padded_batch)
                    self.model.train(was_training)
            else:
                rc, rr, _, _ = self.concatenated_forward(self.ref_model, padded_batch)
        return rc, rr

    def precompute_reference_log_probs(self, dataset, batch_size: Optional[int] = None):
        """trl==0.8.1 get_train_dataloader / get_eval_dataloader with precompute_ref_log_probs=True: one no-grad pass over
        the tokenised dataset, the two log-probs are stored on every row as `reference_chosen_logps` /
        `reference_rejected_logps`; the collator turns them into float tensors and the training step then skips the
        reference forward (21 % of the step at the 7B configuration).  Every rank walks the whole dataset, as in trl."""
        if dataset is None or not len(dataset) or "reference_chosen_logps" in dataset[0]:
            return dataset
        bs = int(batch_size or getattr(self.args, "per_device_eval_batch_size", None) or
                 getattr(self.args, "per_device_train_batch_size", 4))
        for i in range(0, len(dataset), bs):
            rows = dataset[i:i + bs]
            batch = self._prepare_inputs(self.data_collator(rows))
            rc, rr = self.compute_reference_log_probs(batch)
            rc, rr = rc.float().cpu().tolist(), rr.float().cpu().tolist()
            for r, c, j in zip(rows, rc, rr):
                r["reference_chosen_logps"], r["reference_rejected_logps"] = c, j
        return dataset

    def _train_row_batches(self, epoch: int):
        """One epoch of this rank's row batches.  torch DistributedSampler semantics (accelerate MULTI_GPU, ddp.yaml): the
        shuffled index list is padded by wrapping to a multiple of the world size so that EVERY rank gets the same number
        of batches (unequal counts would dead-lock the gradient all-reduce), then strided by rank.  HF's default
        dataloader_drop_last=False: the final partial batch is kept."""
        bs = int(getattr(self.args, "per_device_train_batch_size", 4))
        world, rank = _world(), _rank()
        idx = list(range(len(self.train_dataset)))
        random.Random(int(getattr(self.args, "seed", 42)) + epoch).shuffle(idx)
        if world > 1 and len(idx) % world:
            idx += idx[: world - len(idx) % world]
        idx = idx[rank::world]
        drop_last = bool(getattr(self.args, "dataloader_drop_last", False))
        for i in range(0, len(idx), bs):
            rows = idx[i:i + bs]
            if len(rows) < bs and drop_last:
                break
            yield [self.train_dataset[j] for j in rows]

    def _batches_per_epoch(self) -> int:
        bs = int(getattr(self.args, "per_device_train_batch_size", 4))
        n = -(-len(self.train_dataset) // _world())
        return n // bs if getattr(self.args, "dataloader_drop_last", False) else -(-n // bs)

    def get_train_batches(self, epoch: int, skip: int = 0):
        """collated batches of one epoch.  The collator (image decode + CLIP preprocess) runs `dataloader_prefetch` batches
        ahead on a background thread and the H2D copy goes through a copy stream (base/loader.py); 0 = collate inline."""
        import itertools
        depth = int(getattr(self.args, "dataloader_prefetch", 2) or 0)
        rows_iter = lambda: itertools.islice(self._train_row_batches(epoch), skip, None)   # noqa: E731  (resume: skip consumed batches un-collated)
        if depth <= 0:
            for rows in rows_iter():
                yield self.data_collator(rows)
            return
        from .loader import PrefetchLoader
        yield from PrefetchLoader(rows_iter, self.data_collator, self.accelerator.device, depth)

    def lr_at(self, step: int, total: int) -> float:
        """transformers get_scheduler('cosine' | 'linear' | 'constant') with warmup_ratio / warmup_steps."""
        a = self.args
        base = float(getattr(a, "learning_rate", 5e-5))
        warm = int(getattr(a, "warmup_steps", 0) or 0) or math.ceil(float(getattr(a, "warmup_ratio", 0.0) or 0.0) * total)
        kind = str(getattr(a, "lr_scheduler_type", "linear")).split(".")[-1].lower()
        if step < warm:
            return base * step / max(1, warm)
        prog = (step - warm) / max(1, total - warm)
        if kind == "cosine":
            return base * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
        if kind.startswith("constant"):
            return base
        return base * max(0.0, 1.0 - prog)

    # ------------------------------------------------------------------------------------------ checkpoints
    def _checkpoint_dirs(self):
        import os
        import re
        out_dir = str(getattr(self.args, "output_dir", "output"))
        if not os.path.isdir(out_dir):
            return []
        found = [(int(m.group(1)), os.path.join(out_dir, d)) for d in os.listdir(out_dir)
                 if (m := re.fullmatch(r"checkpoint-(\d+)", d)) and os.path.isfile(os.path.join(out_dir, d, "trainer_state.json"))]
        return [p for _, p in sorted(found)]

    def save_checkpoint(self, step: int, micro: int, epoch: int, window_len: int = 0):
        """HF Trainer._save_checkpoint for this path: `output_dir/checkpoint-<step>/` with the weights (adapters under LoRA),
        the optimizer state (fp32 master / m / v + step), trainer_state.json (step counters, log history, the dropout call counters
        that seed the counter-based masks, world size / accumulation steps the counters were taken under); rotated to
        `save_total_limit`.  Rank 0 writes (every rank holds identical state under DDP)."""
        import json
        import os
        import shutil
        from safetensors.torch import save_file
        if _rank() != 0:
            return None
        eng = self.model.engine
        path = os.path.join(str(getattr(self.args, "output_dir", "output")), f"checkpoint-{step}")
        tmp = path + ".tmp"
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        if self.is_peft_model:
            self.model.save_adapter(tmp)
        else:
            self.model.save_pretrained(tmp)
        st = eng.optimizer_state()
        if st is not None:
            for k in ("master", "m", "v"):          # one file per buffer: 27 GB each at 7B full fine-tuning
                save_file({k: st[k].detach().cpu()}, os.path.join(tmp, f"optimizer_{k}.safetensors"))
        with open(os.path.join(tmp, "trainer_state.json"), "w") as f:
            json.dump(dict(global_step=step, micro_step=micro, epoch=epoch, opt_step=eng.opt_step, log_history=self.log_history,
                           world_size=_world(), lora_calls=getattr(eng, "_lora_calls", 0), plora_calls=getattr(eng, "_plora_calls", 0),
                           gradient_accumulation_steps=max(1, int(getattr(self.args, "gradient_accumulation_steps", 1) or 1)),
                           per_device_train_batch_size=int(getattr(self.args, "per_device_train_batch_size", 4))), f, indent=1)
        shutil.rmtree(path, ignore_errors=True)
        os.replace(tmp, path)                        # a checkpoint directory is either complete or absent
        limit = int(getattr(self.args, "save_total_limit", 0) or 0)
        if limit > 0:
            for old in self._checkpoint_dirs()[:-limit]:
                shutil.rmtree(old, ignore_errors=True)
        return path

    def load_checkpoint(self, path: str) -> dict:
        import json
        import os
        from safetensors.torch import load_file
        eng = self.model.engine
        with open(os.path.join(path, "trainer_state.json")) as f:
            state = json.load(f)
        # micro_step -> (epoch, batches to skip) only means the same thing under the same sharding
        for key, now in (("world_size", _world()), ("gradient_accumulation_steps", max(1, int(getattr(self.args, "gradient_accumulation_steps", 1) or 1))),
                         ("per_device_train_batch_size", int(getattr(self.args, "per_device_train_batch_size", 4)))):
            if key in state and int(state[key]) != now:
                raise ValueError(f"resume_from_checkpoint: {path} was written with {key}={state[key]}, this run has {now}")
        if self.is_peft_model:
            self.model.load_adapter(path)
        else:
            sd = {}
            for fn in sorted(os.listdir(path)):
                if fn.startswith("model") and fn.endswith(".safetensors"):
                    sd.update(load_file(os.path.join(path, fn)))
            eng.policy.load_state_dict(sd)
        if os.path.isfile(os.path.join(path, "optimizer_master.safetensors")):
            bufs = {k: load_file(os.path.join(path, f"optimizer_{k}.safetensors"))[k] for k in ("master", "m", "v")}
            eng.load_optimizer_state(bufs["master"], bufs["m"], bufs["v"], state["opt_step"])
        if hasattr(eng, "_lora_calls"):
            eng._lora_calls = int(state.get("lora_calls", 0))
        if hasattr(eng, "_plora_calls"):
            eng._plora_calls = int(state.get("plora_calls", 0))
        self.log_history = list(state.get("log_history", []))
        return state

    def train(self, resume_from_checkpoint=None):
        a = self.args
        eng = self.model.engine
        ga = max(1, int(getattr(a, "gradient_accumulation_steps", 1) or 1))
        n_batches = self._batches_per_epoch()
        if n_batches == 0:
            raise ValueError(f"the training set ({len(self.train_dataset)} rows over {_world()} rank(s)) yields no batch of "
                             f"per_device_train_batch_size={getattr(a, 'per_device_train_batch_size', 4)} with dataloader_drop_last")
        per_epoch = max(1, n_batches // ga)
        max_steps = int(getattr(a, "max_steps", -1) or -1)
        epochs = float(getattr(a, "num_train_epochs", 1.0))
        total = max_steps if max_steps > 0 else int(math.ceil(per_epoch * epochs))
        logging_steps = max(1, int(getattr(a, "logging_steps", 10) or 10))
        save_strategy = str(getattr(a, "save_strategy", "no")).split(".")[-1].lower()
        save_steps = max(1, int(getattr(a, "save_steps", 500) or 500))
        if self.precompute_ref_log_probs and not self._precomputed_train_ref_log_probs:
            self.precompute_reference_log_probs(self.train_dataset)
            self._precomputed_train_ref_log_probs = True
        eng.init_optimizer() if eng.master is None else None
        eng.zero_grad()
        step, micro, ep = 0, 0, 0
        skip = 0              # micro-batches of the resumed epoch that were already consumed
        if resume_from_checkpoint:
            ckpt = resume_from_checkpoint if isinstance(resume_from_checkpoint, str) else (self._checkpoint_dirs() or [None])[-1]
            if ckpt is None:
                raise ValueError(f"resume_from_checkpoint: no checkpoint-* directory under {getattr(a, 'output_dir', 'output')}")
            st = self.load_checkpoint(ckpt)
            step, micro = int(st["global_step"]), int(st["micro_step"])
            ep, skip = divmod(micro, n_batches)
            self.state.global_step = step
        last_saved = [-1]

        def save(step_, micro_, ep_):
            last_saved[0] = step_
            return self.save_checkpoint(step_, micro_, ep_)

        window = []           # device scalars; only read back at logging time (no per-step host sync)
        epoch_save_due = False
        while step < total:
            it = iter(self.get_train_batches(ep, skip=skip))
            nxt = next(it, None)
            while nxt is not None:
                batch = nxt
                if eng.reducer is not None:        # DDP no_sync: reduce only with the last micro-batch of an accumulation window
                    eng.reducer.enabled = (micro + 1) % ga == 0
                window.append(self.training_step(self.model, batch))
                nxt = next(it, None)
                if nxt is not None:                # look-ahead of one batch: its reference forward runs under the optimizer step
                    nxt = self.prefetch_reference(nxt)
                micro += 1
                if micro % ga:
                    continue
                lr = self.lr_at(step, total)
                eng.optimizer_step(lr=lr, beta1=float(getattr(a, "adam_beta1", 0.9)), beta2=float(getattr(a, "adam_beta2", 0.999)),
                                   eps=float(getattr(a, "adam_epsilon", 1e-8)), weight_decay=float(getattr(a, "weight_decay", 0.0)),
                                   max_grad_norm=float(getattr(a, "max_grad_norm", 1.0) or 0.0),
                                   grad_scale=(1.0 / (ga * _world())) if eng.reducer is not None else 1.0 / ga)
                step += 1
                self.state.global_step = step
                if step % logging_steps == 0 or step >= total:
                    n_opt = max(1, len(window) // ga)
                    self.log({"loss": float(torch.stack(window).sum()) / n_opt, "learning_rate": lr,
                              "grad_norm": eng.grad_norm(), "epoch": micro / ga / per_epoch})
                    window = []
                ev = str(getattr(a, "evaluation_strategy", "no")).split(".")[-1].lower()
                if ev == "steps" and self.eval_dataset and step % max(1, int(getattr(a, "eval_steps", None) or logging_steps)) == 0:
                    self.evaluate()
                if save_strategy == "steps" and step % save_steps == 0:
                    save(step, micro, ep)
                elif epoch_save_due:                    # the epoch ended inside an accumulation window: saved at the first optimizer step after it
                    save(step, micro, ep)
                epoch_save_due = False
                if step >= total:
                    break
            # HF fires on_epoch_end - and with save_strategy="epoch" saves - at every epoch end INCLUDING the one training stops in, whether
            # the epoch was consumed or max_steps cut it short (the loop leaves on an optimizer-step boundary in both cases).  A checkpoint is
            # only state at such a boundary: an epoch that ends inside an accumulation window defers its save to the next optimizer step.
            if save_strategy == "epoch":
                if step >= total:
                    if last_saved[0] != step:
                        save(step, micro, ep)
                elif micro % ga == 0:
                    save(step, micro, ep)
                else:
                    epoch_save_due = True
                    if _rank() == 0:
                        print(f"[vlrlhf] epoch {ep} ended inside an accumulation window ({micro % ga} of {ga} micro-batches): its checkpoint is written at optimizer step {step + 1}", flush=True)
            ep += 1
            skip = 0
        return self.state

    def save_state(self):
        """HF Trainer.save_state: trainer_state.json (step counter + log history) in output_dir, rank 0."""
        import json
        import os
        if _rank() != 0:
            return
        out_dir = str(getattr(self.args, "output_dir", "output"))
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "trainer_state.json"), "w") as f:
            json.dump(dict(global_step=self.state.global_step, log_history=self.log_history), f, indent=1)

    def add_callback(self, cb):
        self.callbacks.append(cb)

    def get_batch_samples(self, model, batch):
        """reference base/trainer.py:310-360: sample a continuation of every prompt from the policy and from the reference (the batch's
        `reference_output`, else the reference model, else the policy with its adapters disabled), pad to max_length, decode.
        Evaluation-time only (`generate_during_eval`; the reference CLI never switches it on): `model.generate` re-runs the HIP forward
        per token, no KV cache."""
        others = dict(batch.get("img_input_dict", {}))
        kw = dict(input_ids=batch["prompt_input_ids"], attention_mask=batch["prompt_attention_mask"], max_length=self.max_length,
                  do_sample=True, pad_token_id=self.tokenizer.pad_token_id, **others)
        # the stop token is the TOKENIZER's (Qwen-VL's HF config has no eos_token_id: its end token is tokenizer.eod_id; the model-side
        # default of `generate` is LLaMA's id 2)
        eos = getattr(self.tokenizer, "eos_token_id", None)
        if eos is None:
            eos = getattr(self.tokenizer, "eod_id", None)
        if eos is not None:
            kw["eos_token_id"] = eos
        policy_output = model.generate(**kw)
        if "reference_output" in batch:
            reference_output = batch["reference_output"]
        elif self.ref_model is None:
            with self.null_ref_context():
                reference_output = self.model.generate(**kw)
        else:
            reference_output = self.ref_model.generate(**kw)
        pad = self.tokenizer.pad_token_id
        policy_output = pad_to_length(policy_output, self.max_length, pad)
        reference_output = pad_to_length(reference_output, self.max_length, pad)
        return (self.tokenizer.batch_decode(policy_output, skip_special_tokens=True),
                self.tokenizer.batch_decode(reference_output, skip_special_tokens=True))


class _State:
    global_step = 0


class _Accelerator:
    """the two attributes of accelerate.Accelerator the reference trainer touches on this path."""

    def __init__(self, model):
        eng = getattr(model, "engine", None)
        self.device = eng.dev if eng is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _backend():
    import torch.distributed as dist
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None
