"""LLaVA-1.5 wrapper for the MI355X DPO path - mirror of /root/reference/src/vlrlhf/models/Llava/__init__.py
(LlavaRLOutputWithPast :22-32, LlavaForRL :35-298, LlavaProcessor :315-432, LlavaDPODataCollatorWithPadding :435-443,
LlavaDPOTrainer :474, core_mapper :486-499).  `LlavaForRL.forward` keeps the reference call signature and output
fields; underneath it runs vlrlhf.engine (HIP kernels) and returns a LAZY logits handle so the [2B,S,V] fp32 tensor is
never written to HBM unless a caller asks for it."""
import contextlib
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Literal, Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import _hip
from ...base.collator import VLDPODataCollatorWithPadding
from ...base.processor import VLChatTemplate, VLProcessor
from ...base.trainer import VLDPOTrainer
from ...engine import BF16, LlavaHipEngine, WeightSet
from ...utils.common import flatten_list
from ..utils import ModelCoreMapper


@dataclass
class LlavaRLOutputWithPast:
    loss: Optional[torch.Tensor] = None
    logits: Any = None
    past_key_values: Optional[List[torch.Tensor]] = None
    hidden_states: Optional[Tuple[torch.Tensor]] = None
    attentions: Optional[Tuple[torch.Tensor]] = None
    image_hidden_states: Optional[Tuple[torch.Tensor]] = None
    labels: Optional[torch.Tensor] = None
    image_position_map: Optional[torch.Tensor] = None

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else (self.loss, self.logits)[k]


# ----------------------------------------------------------------------------------------------------------
class _HiddenFn(torch.autograd.Function):
    """embed -> ViT -> projector -> merge -> 32 decoder layers -> final norm, as ONE autograd node.  Parameter
    gradients are written by the HIP backward straight into the engine's flat gradient buffer (the .grad of every
    nn.Parameter is a view of it), so this node returns no tensor gradients."""

    @staticmethod
    def forward(ctx, anchor, model, input_ids, attention_mask, labels, pixel_values, dup, image_sizes=None):
        c = model.engine.forward_hidden(model.weights, input_ids, attention_mask, labels, pixel_values, image_dup=dup,
                                        save=True, tag="policy", image_sizes=image_sizes)
        ctx.c, ctx.engine = c, model.engine
        model._last_ctx = c
        return c["hidden"]

    @staticmethod
    def backward(ctx, dhidden):
        ctx.engine.hidden_backward(ctx.c, dhidden.contiguous())
        ctx.c = None
        return (None,) * 8


class _LogpsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, engine, c, labels, shared, average, label_pad):
        logps, lp = engine.logps_forward(c, labels, shared, average, label_pad)
        ctx.lp, ctx.engine = lp, engine
        return logps

    @staticmethod
    def backward(ctx, dlogps):
        dh = ctx.engine.logps_backward(ctx.lp, dlogps.contiguous())
        ctx.lp = None
        return (dh,) + (None,) * 6


class LazyLogits:
    """Stand-in for the [B,S,V] logits of the reference (`output.logits`): supports what the DPO path does with them -
    `get_batch_logps` (fused lm-head + log-softmax pick on the response rows), batch slicing, `.detach()`, `.mean()`
    (trl's logits/* metrics) - and `materialize()` for callers that need the tensor."""

    def __init__(self, engine, c, hidden, lo=0, hi=None):
        self.engine, self.c, self.hidden = engine, c, hidden
        self.lo, self.hi = lo, (c["Bn"] if hi is None else hi)

    @property
    def shape(self):
        return torch.Size((self.hi - self.lo, self.c["S"], self.engine.V))

    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        return self.engine.dev

    is_cuda = True

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            lo, hi, st = idx.indices(self.hi - self.lo)
            assert st == 1
            return LazyLogits(self.engine, self.c, self.hidden, self.lo + lo, self.lo + hi)
        return self.materialize()[idx]

    def detach(self):
        return LazyLogits(self.engine, self.c, self.hidden.detach(), self.lo, self.hi)

    def float(self):
        return self

    def to(self, *a, **k):
        return self

    def mean(self):
        return self.engine.logits_mean(self.c, self.lo, self.hi)

    def materialize(self):
        return self.engine.materialize_logits(self.c, self.lo, self.hi)

    def batch_logps(self, labels, shared, average, label_pad):
        assert self.lo == 0 and self.hi == self.c["Bn"], "get_batch_logps expects the whole concatenated batch"
        if self.hidden.requires_grad:
            return _LogpsFn.apply(self.hidden, self.engine, self.c, labels, shared, bool(average), int(label_pad))
        logps, _ = self.engine.logps_forward(self.c, labels, shared, bool(average), int(label_pad))
        return logps


class _Config(dict):
    """attribute + dict access; the handful of HF config fields the reference reads."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def _cfg_from_hf(hf: dict) -> dict:
    t, v = hf.get("text_config", {}), hf.get("vision_config", {})
    hidden = t.get("hidden_size", 4096)
    extra = {}
    if t.get("num_key_value_heads") and t["num_key_value_heads"] != t.get("num_attention_heads", 32):
        extra["kv_heads"] = t["num_key_value_heads"]
    if hf.get("image_grid_pinpoints"):
        extra["image_grid_pinpoints"] = [list(p) for p in hf["image_grid_pinpoints"]]
    return dict(
        **extra,
        vit_hidden=v.get("hidden_size", 1024), vit_mlp=v.get("intermediate_size", 4096),
        vit_layers=v.get("num_hidden_layers", 24), vit_heads=v.get("num_attention_heads", 16),
        image_size=v.get("image_size", 336), patch_size=v.get("patch_size", 14), vit_ln_eps=v.get("layer_norm_eps", 1e-5),
        hidden=hidden, inter=t.get("intermediate_size", 11008), layers=t.get("num_hidden_layers", 32),
        heads=t.get("num_attention_heads", 32), vocab=t.get("vocab_size", hf.get("vocab_size", 32064)),
        rms_eps=t.get("rms_norm_eps", 1e-5), rope_theta=t.get("rope_theta", 10000.0),
        image_token=hf.get("image_token_index", 32000), model_pad_token_id=hf.get("pad_token_id", 32001),
        ignore_index=hf.get("ignore_index", -100))


def _hf_from_cfg(c: dict) -> dict:
    """config.json (transformers==4.41.0 LlavaConfig / LlavaNextConfig layout) for a model that was not loaded from a checkpoint."""
    nxt = bool(c.get("image_grid_pinpoints"))
    return dict(
        **({"image_grid_pinpoints": c["image_grid_pinpoints"]} if nxt else {}),
        architectures=["LlavaNextForConditionalGeneration" if nxt else "LlavaForConditionalGeneration"],
        model_type="llava_next" if nxt else "llava", image_token_index=c["image_token"],
        pad_token_id=c.get("model_pad_token_id", c["image_token"] + 1), ignore_index=c.get("ignore_index", -100),
        projector_hidden_act="gelu", vision_feature_layer=-2, vision_feature_select_strategy="default",
        vocab_size=c["vocab"], torch_dtype="bfloat16",
        text_config=dict(model_type="mistral" if nxt and c.get("kv_heads") else "llama", hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["layers"],
                         num_attention_heads=c["heads"], num_key_value_heads=c.get("kv_heads", c["heads"]), vocab_size=c["vocab"],
                         rms_norm_eps=c.get("rms_eps", 1e-5), rope_theta=c.get("rope_theta", 10000.0)),
        vision_config=dict(model_type="clip_vision_model", hidden_size=c["vit_hidden"], intermediate_size=c["vit_mlp"],
                           num_hidden_layers=c["vit_layers"], num_attention_heads=c["vit_heads"], image_size=c["image_size"],
                           patch_size=c["patch_size"], layer_norm_eps=c.get("vit_ln_eps", 1e-5)))


def sampling_filter(logits, temperature=1.0, top_k=50, top_p=1.0):
    """transformers' logits warpers in their order (TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper): the filtered-out entries
    become -inf; at least one token always survives"""
    logits = logits / max(float(temperature), 1e-6)
    if top_k and top_k > 0:
        kth = torch.topk(logits, min(int(top_k), logits.shape[-1]), dim=-1).values[:, -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p < 1.0:
        srt, idx = torch.sort(logits, dim=-1, descending=False)
        drop = srt.softmax(-1).cumsum(-1) <= (1.0 - float(top_p))
        drop[:, -1] = False
        logits = logits.masked_fill(drop.scatter(1, idx, drop), float("-inf"))
    return logits


class LlavaForRL(nn.Module):
    engine_cls = LlavaHipEngine

    def __init__(self, cfg: dict, engine: Optional[LlavaHipEngine] = None, weights: Optional[WeightSet] = None,
                 trainable: bool = True):
        super().__init__()
        self.engine = engine if engine is not None else self.engine_cls(cfg)
        self.weights = weights if weights is not None else self.engine.policy
        self.config = _Config(cfg)
        self.config.setdefault("is_encoder_decoder", False)
        self.config.setdefault("image_token_index", cfg.get("image_token", -1))
        self.config.setdefault("ignore_index", -100)
        self.config.setdefault("use_cache", False)
        self.pad_token_id = cfg.get("model_pad_token_id", cfg.get("image_token", -1) + 1)
        self._trainable = trainable and self.weights is self.engine.policy
        self._params = nn.ParameterDict()
        self._hf_names = {}
        if self._trainable:
            sd = self.weights.state_dict()
            for i, (hf, v) in enumerate(sd.items()):
                p = nn.Parameter(v, requires_grad=True)
                self._params[f"p{i}"] = p
                self._hf_names[f"_params.p{i}"] = hf
            gsd = WeightSet(self.engine.layout, self.engine.dev, self.engine.grads).state_dict()
            for i, hf in enumerate(sd.keys()):
                self._params[f"p{i}"].grad = gsd[hf]
        self._anchor = nn.Parameter(torch.zeros(1, device=self.engine.dev), requires_grad=True)
        self._last_ctx = None
        self._vision_frozen = True

    # ---- construction ---------------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, cfg: dict, sd: Dict[str, torch.Tensor], **kw):
        m = cls(cfg, **kw)
        m.engine.load_state_dict(sd)
        return m

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, torch_dtype=None, use_flash_attention_2=None, **kwargs):
        """reads a transformers==4.41.0 LLaVA checkpoint directory (config.json + *.safetensors)."""
        from safetensors.torch import load_file
        path = pretrained_model_name_or_path
        with open(os.path.join(path, "config.json")) as f:
            cfg = _cfg_from_hf(json.load(f))
        m = cls(cfg)
        with open(os.path.join(path, "config.json")) as f:
            m.hf_config = json.load(f)
        sd = {}
        idx = os.path.join(path, "model.safetensors.index.json")
        files = sorted(set(json.load(open(idx))["weight_map"].values())) if os.path.exists(idx) else \
            [f for f in os.listdir(path) if f.endswith(".safetensors")]
        for fn in files:
            sd.update(load_file(os.path.join(path, fn)))
        m.engine.load_state_dict(sd)
        return m

    def save_pretrained(self, output_dir, max_shard_bytes: int = 5 << 30, state_dict=None):
        """Writes a directory `from_pretrained` (and transformers) can load: config.json + sharded safetensors of the WHOLE
        model - language model, projector and the frozen vision tower's original tensors (reference: HF Trainer._save ->
        model.save_pretrained).  `state_dict`: alternative LLM/projector tensors (e.g. merge_and_unload())."""
        from safetensors.torch import save_file
        os.makedirs(output_dir, exist_ok=True)
        hf = dict(getattr(self, "hf_config", None) or _hf_from_cfg(self.engine.cfg))
        hf.setdefault("architectures", ["LlavaForConditionalGeneration"])
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump(hf, f, indent=1)
        sd = dict(state_dict if state_dict is not None else self.state_dict())
        sd.update(self.engine.vision_sd)
        shards, cur, size = [], {}, 0
        for k, v in sd.items():
            nb = v.numel() * v.element_size()
            if cur and size + nb > max_shard_bytes:
                shards.append(cur)
                cur, size = {}, 0
            cur[k] = v
            size += nb
        shards.append(cur)
        if len(shards) == 1:
            save_file({k: v.detach().contiguous().cpu() for k, v in shards[0].items()}, os.path.join(output_dir, "model.safetensors"))
            return
        wmap = {}
        for i, sh in enumerate(shards):
            fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file({k: v.detach().contiguous().cpu() for k, v in sh.items()}, os.path.join(output_dir, fn))
            wmap.update({k: fn for k in sh})
        with open(os.path.join(output_dir, "model.safetensors.index.json"), "w") as f:
            json.dump(dict(metadata=dict(total_size=sum(v.numel() * v.element_size() for v in sd.values())), weight_map=wmap), f, indent=1)

    def save_adapter(self, output_dir, base_model_name_or_path=None):
        """peft PeftModel.save_pretrained layout: adapter_model.safetensors + adapter_config.json holding LoraConfig fields only
        (reference utils/common.py:97-98 saves exactly the adapter tensors)."""
        from safetensors.torch import save_file
        os.makedirs(output_dir, exist_ok=True)
        save_file({k: v.contiguous().cpu() for k, v in self.lora_state_dict().items()}, os.path.join(output_dir, "adapter_model.safetensors"))
        lo = self.engine.lora
        cfg = dict(peft_type="LORA", task_type="CAUSAL_LM", base_model_name_or_path=base_model_name_or_path, r=lo["r"],
                   lora_alpha=lo["alpha"], lora_dropout=lo["dropout"], target_modules=list(self.default_lora_target), bias="none",
                   fan_in_fan_out=False, inference_mode=True, modules_to_save=None, init_lora_weights=True)
        with open(os.path.join(output_dir, "adapter_config.json"), "w") as f:
            json.dump(cfg, f, indent=1)

    def load_adapter(self, path):
        from safetensors.torch import load_file
        self.engine.load_lora_state_dict(load_file(os.path.join(path, "adapter_model.safetensors")))

    # ---- LoRA (peft) ---------------------------------------------------------------------------------------
    def apply_lora(self, peft_config):
        """peft.get_peft_model(self, LoraConfig(...)) as the reference's trainer applies it (trl DPOTrainer.__init__ with
        peft_config from utils/auto_load.py:559-571): freezes the base weights, adds adapters on default_lora_target,
        and leaves only lora_A / lora_B trainable.  `peft_config` is a LoraConfig-like object or a dict."""
        get = (lambda k, d=None: peft_config.get(k, d)) if isinstance(peft_config, dict) else (lambda k, d=None: getattr(peft_config, k, d))
        targets = get("target_modules")
        if targets in (None, "auto"):
            targets = self.default_lora_target
        if isinstance(targets, str):
            targets = targets.split(",")
        if sorted(targets) != sorted(self.default_lora_target):
            raise NotImplementedError(f"LoRA target_modules {sorted(targets)}: the MI355X path adapts exactly the decoder "
                                      f"linears {self.default_lora_target} (LlavaForRL.default_lora_target)")
        if get("bias", "none") != "none":
            raise NotImplementedError("lora_bias other than 'none' is not supported on the MI355X path")
        if get("modules_to_save"):
            raise NotImplementedError("modules_to_save is not supported on the MI355X path")
        if not self._trainable:
            raise ValueError("apply_lora needs the trainable policy model")
        self.engine.enable_lora(int(get("r", 64)), float(get("lora_alpha", 16)), float(get("lora_dropout", 0.0) or 0.0),
                                seed=int(get("seed", 0) or 0))
        self.engine.training = self.training
        self._params = nn.ParameterDict()
        self._hf_names = {}
        names = self.engine.lora_layout.hf_names()
        for i, (hf, (k, lo, hi)) in enumerate(names.items()):
            p = nn.Parameter(self.engine.lv[k][lo:hi], requires_grad=True)
            p.grad = self.engine.lgv[k][lo:hi]
            self._params[f"p{i}"] = p
            self._hf_names[f"_params.p{i}"] = hf.replace(".weight", ".default.weight")
        self.peft_config = {"default": peft_config}
        return self

    @property
    def is_peft_model(self):
        return self.engine.lora is not None and self._trainable

    @contextlib.contextmanager
    def disable_adapter(self):
        """peft PeftModel.disable_adapter(): inside, forwards use the frozen base weights only - trl's null_ref_context
        turns the policy into its own reference model this way."""
        prev = self.engine.lora_active
        self.engine.lora_active = False
        try:
            yield
        finally:
            self.engine.lora_active = prev

    def lora_state_dict(self):
        self.engine.wait_optimizer()
        return self.engine.lora_state_dict()

    def merge_and_unload(self):
        """state dict of the base model with the adapters folded in (peft merge_and_unload)"""
        return self.engine.merged_weights().state_dict()

    def train(self, mode: bool = True):
        super().train(mode)
        if self._trainable:
            self.engine.training = bool(mode)     # lora_dropout is active in training mode only
        return self

    def create_reference_model(self):
        """trl.create_reference_model: a frozen deep copy of the policy weights sharing the engine (and the frozen ViT)."""
        ref = type(self)(dict(self.engine.cfg), engine=self.engine, weights=self.weights.clone(), trainable=False)
        ref.eval()
        return ref

    # ---- reference wrapper API (docs/CustomizedModel.md:7-11; Llava/__init__.py:273-298) ------------------
    def named_parameters(self, *a, **k):
        for n, p in super().named_parameters(*a, **k):
            if n == "_anchor":
                continue
            yield self._hf_names.get(n, n), p

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def state_dict(self, *a, **k):
        self.engine.wait_optimizer()
        out = dict(self.weights.state_dict())
        return out

    @property
    def default_lora_target(self):
        return ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]

    def get_vision_tower(self):
        return self.engine.vision

    def freeze_vision_tower(self):
        self._vision_frozen = True          # the MI355X path keeps the tower frozen (reference default, auto_load.py:554)

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        """transformers PreTrainedModel API (the HF Trainer calls it for --gradient_checkpointing True, reference dpo.py:99): the engine
        keeps only each decoder layer's input and re-runs the layer's forward right before its backward - bit-identical results"""
        self.engine.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.engine.gradient_checkpointing = False

    @property
    def is_gradient_checkpointing(self):
        return bool(self.engine.gradient_checkpointing)

    def prepare_default_generation_kwargs(self, generation_config):
        generation_config.max_new_tokens = 1024
        generation_config.do_sample = False
        return dict(generation_config=generation_config)

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, max_length=None, max_new_tokens=None, do_sample=False, temperature=1.0,
                 top_k=50, top_p=1.0, pad_token_id=None, eos_token_id=None, generation_config=None, generator=None, **img):
        """What the reference trainer's `get_batch_samples` calls (base/trainer.py:310-360: `model.generate(input_ids, attention_mask,
        max_length, do_sample=True, pad_token_id, **img_input_dict)`, transformers GenerationMixin defaults: temperature 1, top_k 50,
        top_p 1).  Evaluation-time sampling only and outside the DPO step, so there is NO KV cache: every new token re-runs the forward
        of the whole sequence on the HIP path (the vision features of the batch are cached by the engine) and the lm-head is evaluated
        on the last row of each sequence alone.  Prompts are LEFT-padded (trl's collator; the merge of the reference end-aligns such
        rows); the running batch is left-padded further to a multiple of 32 tokens so that the engine sees a new shape every 32 steps,
        not every step.  Finished rows keep receiving `pad_token_id` as in transformers.  Returns prompt + continuation ids."""
        if generation_config is not None:
            max_new_tokens = max_new_tokens if max_new_tokens is not None else getattr(generation_config, "max_new_tokens", None)
            do_sample = bool(getattr(generation_config, "do_sample", do_sample))
        if input_ids is None:
            raise ValueError("generate needs input_ids")
        dev = self.engine.dev
        ids = input_ids.to(dev).long()
        mask = (torch.ones_like(ids) if attention_mask is None else attention_mask.to(dev).long())
        B, T0 = ids.shape
        if max_new_tokens is not None:
            limit = T0 + int(max_new_tokens)
        elif max_length is not None:
            limit = int(max_length)
        else:
            limit = T0 + 20                       # transformers' default max_length is 20 NEW tokens when nothing is given
        eos = eos_token_id if eos_token_id is not None else self.config.get("eos_token_id", 2)
        eos = set(eos) if isinstance(eos, (list, tuple)) else {int(eos)}
        pad = int(pad_token_id if pad_token_id is not None else 0)
        img = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in img.items()}
        was_training = self.training
        self.eval()
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        lm_head = self.weights.v["lm_head"]
        try:
            while ids.shape[1] < limit and bool(unfinished.any()):
                T = ids.shape[1]
                Tp = (T + 31) // 32 * 32
                if Tp != T:                   # left padding: masked out, and the merged rows stay end-aligned
                    fill = torch.full((B, Tp - T), pad, dtype=ids.dtype, device=dev)
                    run_ids, run_mask = torch.cat([fill, ids], 1), torch.cat([torch.zeros_like(fill), mask], 1)
                else:
                    run_ids, run_mask = ids, mask
                out = self(input_ids=run_ids, attention_mask=run_mask, labels=None, use_cache=False, **img)
                c = out.logits.c
                S, H = c["S"], self.engine.H
                valid = c["mask"].view(B, S) != 0
                last = S - 1 - torch.flip(valid, dims=[1]).float().argmax(1)           # last attended position of every row
                rows = max(8, B)
                h_last = torch.zeros(rows, H, dtype=torch.bfloat16, device=dev)
                h_last[:B] = c["hidden"].view(B, S, H)[torch.arange(B, device=dev), last]
                logits = torch.empty(rows, self.engine.V, dtype=torch.float32, device=dev)
                _hip.call("vlr_gemm_bf16", 0, h_last, lm_head, logits, None, None, rows, self.engine.V, H, H, H, self.engine.V, 0, 0, 0, 1)
                logits = logits[:B]
                if do_sample:
                    nxt = torch.multinomial(sampling_filter(logits, temperature, top_k, top_p).softmax(-1), 1, generator=generator).squeeze(1)
                else:
                    nxt = logits.argmax(-1)
                nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
                ids = torch.cat([ids, nxt[:, None]], 1)
                mask = torch.cat([mask, torch.ones(B, 1, dtype=mask.dtype, device=dev)], 1)
                for e in eos:
                    unfinished = unfinished & (nxt != e)
        finally:
            self.train(was_training)
        return ids

    def zero_grad(self, set_to_none: bool = True):
        self.engine.zero_grad()

    def prefetch_vision(self, img_input_dict):
        pv = img_input_dict.get("pixel_values")
        if pv is not None:
            dup = getattr(pv, "_vlr_dup", 1)
            self.engine.vision_features(pv[: pv.shape[0] // dup] if dup > 1 else pv)

    def get_input_embeddings(self):
        return self.weights.v["embed"]

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, input_ids=None, pixel_values=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, vision_feature_layer=None, vision_feature_select_strategy=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, image_sizes=None):
        """reference Llava/__init__.py:111-271 on the training path.  Returns `logits` (lazy), the EXPANDED `labels`
        and `image_position_map`; the reference's internal cross-entropy `loss` (:246-257, unused by DPO) is None."""
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("generation / KV-cache inputs are outside the MI355X DPO training path")
        if pixel_values is None:
            raise ValueError("LlavaForRL.forward on the DPO path needs pixel_values")
        if vision_feature_layer not in (None, -2) or vision_feature_select_strategy not in (None, "default"):
            raise ValueError(f"Unexpected select feature strategy: {vision_feature_select_strategy}")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        dup = int(getattr(pixel_values, "_vlr_dup", 1))
        grad = torch.is_grad_enabled() and self._trainable and self.training
        if self.engine.anyres and image_sizes is None:
            raise ValueError("LLaVA-Next forward needs image_sizes (reference LlavaNext/__init__.py:216-222)")
        if grad:
            hidden = _HiddenFn.apply(self._anchor, self, input_ids, attention_mask, labels, pixel_values, dup, image_sizes)
            c = self._last_ctx
        else:
            c = self.engine.forward_hidden(self.weights, input_ids, attention_mask, labels, pixel_values, image_dup=dup,
                                           save=False, tag="policy_ng" if self.weights is self.engine.policy else "ref",
                                           image_sizes=image_sizes)
            hidden = c["hidden"]
        out_labels = c["labels"] if labels is not None else torch.full_like(c["mask"], -100, dtype=torch.long)
        if c.get("meta") is not None:
            out_labels._vlr_meta = c["meta"]          # per-batch host-side facts (trainer.concatenated_inputs)
        return LlavaRLOutputWithPast(loss=None, logits=LazyLogits(self.engine, c, hidden), labels=out_labels,
                                     image_position_map=c["img_map"])


# ----------------------------------------------------------------------------------------------------------
class LlavaProcessor(VLProcessor):
    def __init__(self, model_name_or_path=None, tokenizer=None, image_processor=None, **kwargs) -> None:
        if model_name_or_path is not None:
            import transformers
            self.processor = transformers.LlavaProcessor.from_pretrained(model_name_or_path, **kwargs)
            self._tok, self._ip = self.processor.tokenizer, self.processor.image_processor
        else:
            self.processor = None
            self._tok, self._ip = tokenizer, image_processor

    @property
    def tokenizer(self):
        return self._tok

    @property
    def chat_template(self):
        return VLChatTemplate(system_begin=None, system_end=None, user_begin="USER: ", user_end="",
                              assistant_begin="ASSISTANT: ", assistant_end="", image_placeholder="<image>\n")

    @property
    def image_processor(self):
        return self._ip

    def save_pretrained(self, output_dir):
        if self.processor is not None:
            return self.processor.save_pretrained(output_dir)

    def process_batch_conv(self, sources, system_message=None, add_end_for_empty_value=False):
        """reference Llava/__init__.py:343-388."""
        if not isinstance(sources, list) or not isinstance(sources[0], list):
            raise ValueError("sources must be a batch of conversations, eg. List[List[Dict]]")
        t = self.chat_template
        begin = {"user": t.user_begin, "assistant": t.assistant_begin}
        end = {"user": t.user_end, "assistant": t.assistant_end}
        raw_texts, b_ids, b_masks, b_labels = [], [], [], []
        for source in sources:
            raw, labels, prev = "", [], 0
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
            assert len(ids) == len(masks) == len(labels), f"input_ids:{len(ids)}, attention_masks:{len(masks)}, labels:{len(labels)}"
            b_ids.append(ids)
            b_masks.append(masks)
            b_labels.append(labels)
            raw_texts.append(raw)
        return {"prompt": None, "answer": None,
                "full": dict(input_ids=b_ids, attention_mask=b_masks, labels=b_labels), "raw_str": raw_texts}

    @staticmethod
    def format_multimodal_prompt(prompt: str, img_paths=None):
        if img_paths is None:
            return prompt
        if not isinstance(img_paths, list):
            img_paths = [img_paths]
        if len(img_paths) == 1 and "<image>" not in prompt:
            return "<image>\n" + prompt
        assert prompt.count("<image>") == len(img_paths), \
            f"The number of given image ({len(img_paths)}) does not match the number of image placeholders in the prompt: {prompt}"
        return prompt.replace("<image>", "<image>\n")

    @staticmethod
    def remove_image_placeholder(prompt: str):
        return prompt.replace("<image>\n", "")

    @staticmethod
    def is_multimodal_prompt_valid(prompt: str):
        return "<image>\n" in prompt

    def train(self):
        self.tokenizer.pad_token = self.tokenizer.unk_token

    def infer(self):
        self.tokenizer.pad_token = self.tokenizer.bos_token

    def __call__(self, texts=None, convs=None, images_path=None, padding=True, padding_side="left", check_format=True):
        inputs = super().__call__(texts, convs, images_path, padding, padding_side, check_format)
        if images_path is not None:
            inputs["pixel_values"] = load_pixel_values(flatten_list(images_path), self.image_processor)
        return inputs


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def load_pixel_values(items, image_processor=None) -> torch.Tensor:
    """paths / PIL images -> CLIP-normalised fp32 [n,3,S,S] via the HF CLIPImageProcessor (reference :435-443); items that
    already are [3,S,S] float tensors (synthetic data) pass through."""
    if all(isinstance(i, torch.Tensor) for i in items):
        return torch.stack([i.float() for i in items])
    from PIL import Image
    imgs = [Image.open(i).convert("RGB") if isinstance(i, str) else i for i in items]
    return image_processor(images=imgs, return_tensors="pt")["pixel_values"]


@dataclass
class LlavaDPODataCollatorWithPadding(VLDPODataCollatorWithPadding):
    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        padded = super().__call__(features)
        ip = self.processor.image_processor if self.processor is not None else None
        padded["img_input_dict"] = dict(pixel_values=load_pixel_values(padded["img_path"], ip))
        return padded


class LlavaDPOTrainer(VLDPOTrainer):
    ...


core_mapper = ModelCoreMapper(
    model=LlavaForRL,
    processor=LlavaProcessor,
    dpo_collator=LlavaDPODataCollatorWithPadding,
    dpo_trainer=LlavaDPOTrainer,
)
