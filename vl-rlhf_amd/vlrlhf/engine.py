"""MI355X engine for the LLaVA-1.5 DPO step: owns the flat parameter / gradient / optimizer buffers and the
activation workspaces, and sequences libvlr_hip.so (include/vlr.h) for the forward, the backward and the optimizer.

PyTorch is used for device memory, streams and the autograd *boundary* only - every FLOP below goes through the C ABI.
There is no fallback path: without the HIP library (or a GPU) constructing the engine raises.

Replaces, for the DPO hot path, what the reference reaches through
  LlavaForRL.forward                    /root/reference/src/vlrlhf/models/Llava/__init__.py:111-271
  VLDPOTrainer.get_batch_logps          /root/reference/src/vlrlhf/base/trainer.py:148-188
  Trainer.training_step / optimizer     (transformers 4.41.0 + torch AdamW; flags scripts/dpo_llava.sh:35-41)
"""
import math
import os
from typing import Dict, List, Optional

import torch

from . import _hip

BF16 = torch.bfloat16


def _align(n, a=8):
    return (n + a - 1) // a * a


class ParamLayout:
    """Flat layout of the TRAINABLE parameters (LLM + projector; the vision tower is frozen, auto_load.py:554-555).

    Weight-decay region first, in the order the backward finishes them (lm_head, layer L-1 .. 0, projector,
    embed_tokens) so contiguous slices are the DDP buckets; then the no-decay region (norm weights, biases)."""

    def __init__(self, cfg):
        H, I, V, D, L = cfg["hidden"], cfg["inter"], cfg["vocab"], cfg.get("vit_hidden", 8), cfg["layers"]
        assert H % 8 == 0 and I % 8 == 0 and V % 8 == 0 and D % 8 == 0
        nh = cfg.get("heads") or 1
        nkv = cfg.get("kv_heads") or nh                       # grouped-query attention (Mistral, InternLM2)
        hd = cfg.get("head_dim") or H // nh
        Nq, Nkv = nh * hd, nkv * hd
        self.entries = []   # (name, shape, [(hf_name, row0, rows)])
        e = self.entries
        qwen = cfg.get("family") == "qwen_vl"
        ilm = cfg.get("family") == "internlm_xc2"
        self.row_perm = {}          # entry name -> row permutation applied when loading (engine row i = checkpoint row perm[i])
        if ilm:
            # InternLM-XComposer2 (reference models/InternLMXC2/modeling_internlm2.py): fused grouped-query wqkv with rows laid out per
            # K/V head as [q_0..q_{g-1} | k | v] (re-ordered to q | k | v blocks at load time), w2(silu(w1(x)) * w3(x)), and a PLoRA pair
            # (build_mlp.py:158-203) on every linear - base-model weights, trained by a full fine-tune
            lm, layer = "", "model.layers.{}."
            nm = dict(lm_head="output.weight", embed="model.tok_embeddings.weight", norm="model.norm.weight", down="feed_forward.w2.weight",
                      gate="feed_forward.w1.weight", up="feed_forward.w3.weight", o="attention.wo.weight", ln1="attention_norm.weight",
                      ln2="ffn_norm.weight")
            g_ = nh // nkv
            idx = torch.arange((nh + 2 * nkv) * hd).view(nkv, g_ + 2, hd)
            qkv_perm = torch.cat([idx[:, :g_].reshape(-1), idx[:, g_].reshape(-1), idx[:, g_ + 1].reshape(-1)])
        elif qwen:
            # Qwen-VL (reference models/QwenVL/modeling_qwen.py): fused biased c_attn, MLP c_proj(w1(x) * silu(w2(x))) -> gate = w2,
            # up = w1; no projector (the resampler is part of the frozen vision tower)
            lm, layer = "", "transformer.h.{}."
            nm = dict(lm_head="lm_head.weight", embed="transformer.wte.weight", norm="transformer.ln_f.weight", down="mlp.c_proj.weight",
                      gate="mlp.w2.weight", up="mlp.w1.weight", o="attn.c_proj.weight", ln1="ln_1.weight", ln2="ln_2.weight")
        else:
            lm, layer = "language_model.", "language_model.model.layers.{}."
            nm = dict(lm_head=lm + "lm_head.weight", embed=lm + "model.embed_tokens.weight", norm=lm + "model.norm.weight",
                      down="mlp.down_proj.weight", gate="mlp.gate_proj.weight", up="mlp.up_proj.weight", o="self_attn.o_proj.weight",
                      ln1="input_layernorm.weight", ln2="post_attention_layernorm.weight")
        e.append(("lm_head", (V, H), [(nm["lm_head"], 0, V)]))
        for l in range(L - 1, -1, -1):
            p = layer.format(l)
            e.append((f"l{l}.wdown", (H, I), [(p + nm["down"], 0, H)]))
            e.append((f"l{l}.wgu", (2 * I, H), [(p + nm["gate"], 0, I), (p + nm["up"], I, I)]))
            e.append((f"l{l}.wo", (H, Nq), [(p + nm["o"], 0, H)]))
            if qwen:
                e.append((f"l{l}.wqkv", (Nq + 2 * Nkv, H), [(p + "attn.c_attn.weight", 0, Nq + 2 * Nkv)]))
            elif ilm:
                pr = int(cfg.get("plora_r", 256))
                e.append((f"l{l}.wqkv", (Nq + 2 * Nkv, H), [(p + "attention.wqkv.weight", 0, Nq + 2 * Nkv)]))
                self.row_perm[f"l{l}.wqkv"] = qkv_perm
                for key, mod, din, dout in (("qkv", "attention.wqkv", H, Nq + 2 * Nkv), ("o", "attention.wo", Nq, H), ("d", "feed_forward.w2", I, H)):
                    e.append((f"l{l}.pa_{key}", (pr, din), [(p + mod + ".Plora_A.weight", 0, pr)]))
                    e.append((f"l{l}.pb_{key}", (dout, pr), [(p + mod + ".Plora_B.weight", 0, dout)]))
                # gate (w1) and up (w3): Plora_A stacked [2r][H], Plora_B stacked [2I][r] like the fused gate|up weight, so that the pair
                # can ride the K loop of the fused SwiGLU GEMM as its adapter segment
                e.append((f"l{l}.pa_gu", (2 * pr, H), [(p + "feed_forward.w1.Plora_A.weight", 0, pr), (p + "feed_forward.w3.Plora_A.weight", pr, pr)]))
                e.append((f"l{l}.pb_gu", (2 * I, pr), [(p + "feed_forward.w1.Plora_B.weight", 0, I), (p + "feed_forward.w3.Plora_B.weight", I, I)]))
                self.row_perm[f"l{l}.pb_qkv"] = qkv_perm
            else:
                e.append((f"l{l}.wqkv", (Nq + 2 * Nkv, H), [(p + "self_attn.q_proj.weight", 0, Nq), (p + "self_attn.k_proj.weight", Nq, Nkv),
                                                            (p + "self_attn.v_proj.weight", Nq + Nkv, Nkv)]))
        self.tail_start = "embed" if (qwen or ilm) else "proj.w2"
        if not (qwen or ilm):
            e.append(("proj.w2", (H, H), [("multi_modal_projector.linear_2.weight", 0, H)]))
            e.append(("proj.w1", (H, D), [("multi_modal_projector.linear_1.weight", 0, H)]))
        if cfg.get("image_grid_pinpoints"):                   # LLaVA-Next: the row appended to every line of the un-padded tile grid
            e.append(("image_newline", (H,), [("image_newline", 0, H)]))
        e.append(("embed", (V, H), [(nm["embed"], 0, V)]))
        ap = "transformer.visual.attn_pool."
        if qwen and cfg.get("visual"):
            # the resampler of the Qwen-VL vision tower stays trainable in a full fine-tune (QwenVLForRL.freeze_vision_tower,
            # reference models/QwenVL/__init__.py:33-37): its weights live in the trainable buffer; the ViT trunk, ln_post and proj do not
            vq = cfg["visual"]
            E_, W_, nq_ = vq["output_dim"], vq["width"], int(vq.get("n_queries", 256))
            e.append(("ap.win", (3 * E_, E_), [(ap + "attn.in_proj_weight", 0, 3 * E_)]))
            e.append(("ap.wo", (E_, E_), [(ap + "attn.out_proj.weight", 0, E_)]))
            e.append(("ap.kv", (E_, W_), [(ap + "kv_proj.weight", 0, E_)]))
            e.append(("ap.query", (nq_, E_), [(ap + "query", 0, nq_)]))
        # HF Trainer.get_decay_parameter_names excludes nn.LayerNorm / LlamaRMSNorm (ALL_LAYERNORM_LAYERS) and `bias` parameters from weight
        # decay.  The RMSNorm classes VENDORED with Qwen-VL (modeling_qwen.py) and InternLM2 (modeling_internlm2.py) are not in that list:
        # the reference decays their weights (weight_decay 0.05 / 0.1 in the shipped scripts), so for these families the norm weights
        # sit in the decay region and only biases (and the resampler's nn.LayerNorm weights) stay outside it.
        norms_decay = qwen or ilm
        def _norms():
            e.append(("norm", (H,), [(nm["norm"], 0, H)]))
            for l in range(L - 1, -1, -1):
                p = layer.format(l)
                e.append((f"l{l}.ln2", (H,), [(p + nm["ln2"], 0, H)]))
                e.append((f"l{l}.ln1", (H,), [(p + nm["ln1"], 0, H)]))
        if norms_decay:
            _norms()
        self.n_decay_entries = len(e)
        if not norms_decay:
            _norms()
        for l in range(L - 1, -1, -1):
            p = layer.format(l)
            if qwen:
                e.append((f"l{l}.bqkv", (Nq + 2 * Nkv,), [(p + "attn.c_attn.bias", 0, Nq + 2 * Nkv)]))
        if qwen and cfg.get("visual"):
            E_ = cfg["visual"]["output_dim"]
            e.append(("ap.bin", (3 * E_,), [(ap + "attn.in_proj_bias", 0, 3 * E_)]))
            e.append(("ap.bo", (E_,), [(ap + "attn.out_proj.bias", 0, E_)]))
            for k_, n_ in (("lnq_w", "ln_q.weight"), ("lnq_b", "ln_q.bias"), ("lnkv_w", "ln_kv.weight"), ("lnkv_b", "ln_kv.bias")):
                e.append((f"ap.{k_}", (E_,), [(ap + n_, 0, E_)]))
        if not (qwen or ilm):
            e.append(("proj.b2", (H,), [("multi_modal_projector.linear_2.bias", 0, H)]))
            e.append(("proj.b1", (H,), [("multi_modal_projector.linear_1.bias", 0, H)]))
        self.n_trainable_entries = len(e)
        if ilm:      # --freeze_vision_tower freezes the projector too (InternLMXC2/__init__.py:252-255): kept OUTSIDE the optimizer's range
            e.append(("proj.w2", (H, H), [("vision_proj.2.weight", 0, H)]))
            e.append(("proj.w1", (H, D), [("vision_proj.0.weight", 0, H)]))
            e.append(("proj.b2", (H,), [("vision_proj.2.bias", 0, H)]))
            e.append(("proj.b1", (H,), [("vision_proj.0.bias", 0, H)]))
        self.offset = {}
        off = 0
        self.n_opt = None
        for i, (name, shape, _) in enumerate(e):
            if i == self.n_decay_entries:
                self.n_decay = off
            if i == self.n_trainable_entries:
                self.n_opt = off
            self.offset[name] = off
            off += _align(int(math.prod(shape)))
        self.numel = off
        if self.n_opt is None:
            self.n_opt = off           # elements [0, n_opt) are trainable (optimizer, gradient norm, DDP buckets); the rest is frozen
        self.shape = {name: shape for name, shape, _ in e}
        # DDP buckets = contiguous slices of the flat gradient in backward-completion order
        self.bucket_after = {}   # event name -> (start, end)
        ts = self.offset[self.tail_start]
        self.bucket_after["lm_head"] = (0, self.offset[f"l{L - 1}.wdown"] if L else ts)
        for l in range(L - 1, -1, -1):
            end = self.offset[f"l{l - 1}.wdown"] if l > 0 else ts
            self.bucket_after[f"layer{l}"] = (self.offset[f"l{l}.wdown"], end)
        self.bucket_after["tail"] = (ts, self.n_opt)

    def hf_names(self):
        for name, shape, parts in self.entries:
            for hf, r0, rows in parts:
                yield hf, name, r0, rows


LORA_KEYS = ("a_qkv", "b_qkv", "a_o", "b_o", "a_gu", "b_gu", "a_down", "b_down")
# (group, peft module path, sub-target names in stacking order)
LORA_GROUPS = (("qkv", "self_attn", ("q_proj", "k_proj", "v_proj")), ("o", "self_attn", ("o_proj",)),
               ("gu", "mlp", ("gate_proj", "up_proj")), ("down", "mlp", ("down_proj",)))
LORA_TARGETS = tuple(t for _, _, ts in LORA_GROUPS for t in ts)
# Qwen-VL (QwenVLForRL.default_lora_target, reference models/QwenVL/__init__.py:26-28): ONE adapter over the fused c_attn, attn.c_proj,
# w2 (= gate) and w1 (= up); mlp.c_proj has none
QWEN_LORA_GROUPS = (("qkv", "attn", ("c_attn",)), ("o", "attn", ("c_proj",)), ("gu", "mlp", ("w2", "w1")))
# InternLM-XComposer2 (InternLMXC2ForRL.default_lora_target, reference models/InternLMXC2/__init__.py:244-245): peft adapters on the five
# PLoRA linears - one over the fused grouped-query wqkv (lora_B rows in the checkpoint's per-K/V-head order)
ILM_LORA_GROUPS = (("qkv", "attention", ("wqkv",)), ("o", "attention", ("wo",)), ("gu", "feed_forward", ("w1", "w3")), ("down", "feed_forward", ("w2",)))


class LoraLayout:
    """Flat bf16 layout of the peft adapters on the decoder linears.  LLaVA / LLaVA-Next (LlavaForRL.default_lora_target,
    /root/reference src/vlrlhf/models/Llava/__init__.py:273-286), per layer: a_qkv [3r,H] | b_qkv [3H,r] | a_o [r,H] | b_o [H,r] |
    a_gu [2r,H] | b_gu [2I,r] | a_down [r,I] | b_down [H,r]; Qwen-VL: a_qkv [r,H] | b_qkv [3H,r] | a_o | b_o | a_gu | b_gu.
    Sub-targets of a fused group are stacked rows, matching vlr_lora_weights (include/vlr.h)."""

    def __init__(self, cfg, r):
        H, I, L = cfg["hidden"], cfg["inter"], cfg["layers"]
        nh = cfg.get("heads") or 1
        nkv = cfg.get("kv_heads") or nh
        hd = cfg.get("head_dim") or H // nh
        Nq, Nkv = nh * hd, nkv * hd
        self.r, self.L = r, L
        self.qwen = cfg.get("family") == "qwen_vl"
        self.row_perm = {}
        if cfg.get("family") == "internlm_xc2":
            self.groups, self.prefix = ILM_LORA_GROUPS, "base_model.model.model.layers."
            per = dict(a_qkv=(r, H), b_qkv=(Nq + 2 * Nkv, r), a_o=(r, Nq), b_o=(H, r), a_gu=(2 * r, H), b_gu=(2 * I, r), a_down=(r, I), b_down=(H, r))
            self.out_dim = dict(wqkv=Nq + 2 * Nkv, wo=H, w1=I, w3=I, w2=H)
            g_ = nh // nkv
            idx = torch.arange((nh + 2 * nkv) * hd).view(nkv, g_ + 2, hd)
            self.row_perm["b_qkv"] = torch.cat([idx[:, :g_].reshape(-1), idx[:, g_].reshape(-1), idx[:, g_ + 1].reshape(-1)])
        elif self.qwen:
            self.groups, self.prefix = QWEN_LORA_GROUPS, "base_model.model.transformer.h."
            per = dict(a_qkv=(r, H), b_qkv=(Nq + 2 * Nkv, r), a_o=(r, Nq), b_o=(H, r), a_gu=(2 * r, H), b_gu=(2 * I, r))
            self.out_dim = dict(c_attn=Nq + 2 * Nkv, c_proj=H, w2=I, w1=I)
        else:
            self.groups, self.prefix = LORA_GROUPS, "base_model.model.language_model.model.layers."
            per = dict(a_qkv=(3 * r, H), b_qkv=(Nq + 2 * Nkv, r), a_o=(r, Nq), b_o=(H, r), a_gu=(2 * r, H), b_gu=(2 * I, r),
                       a_down=(r, I), b_down=(H, r))
            self.out_dim = dict(q_proj=Nq, k_proj=Nkv, v_proj=Nkv, o_proj=H, gate_proj=I, up_proj=I, down_proj=H)
        self.keys = tuple(k for k in LORA_KEYS if k in per)
        self.qkv_targets = 1 if (self.qwen or cfg.get("family") == "internlm_xc2") else 3
        self.offset, self.shape = {}, {}
        o = 0
        for l in range(L):
            for k in self.keys:
                self.offset[f"l{l}.{k}"] = o
                self.shape[f"l{l}.{k}"] = per[k]
                o += _align(per[k][0] * per[k][1])
        self.numel = o

    def hf_names(self, prefix=None):
        """peft adapter-file name -> (flat key, row_lo, row_hi)"""
        prefix = self.prefix if prefix is None else prefix
        out = {}
        r = self.r
        for l in range(self.L):
            for g, mod, ts in self.groups:
                row = 0
                for i, t in enumerate(ts):
                    od = self.out_dim[t]
                    out[f"{prefix}{l}.{mod}.{t}.lora_A.weight"] = (f"l{l}.a_{g}", i * r, (i + 1) * r)
                    out[f"{prefix}{l}.{mod}.{t}.lora_B.weight"] = (f"l{l}.b_{g}", row, row + od)
                    row += od
        return out


class WeightSet:
    """One set of LLM + projector weights in a flat bf16 buffer, with named 2-D views."""

    def __init__(self, layout: ParamLayout, device, flat: Optional[torch.Tensor] = None):
        self.layout = layout
        self.flat = flat if flat is not None else torch.zeros(layout.numel, dtype=BF16, device=device)
        self.version = 0          # bumped by load_state_dict (caches derived from the weights key on it)
        self.v = {n: self.flat[layout.offset[n]: layout.offset[n] + int(math.prod(s))].view(*s)
                  for n, s in layout.shape.items()}

    def clone(self):
        return WeightSet(self.layout, self.flat.device, self.flat.clone())

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict=True):
        self.version += 1
        seen = set()
        for hf, name, r0, rows in self.layout.hf_names():
            if hf not in sd:
                if strict:
                    raise KeyError(f"missing weight {hf}")
                continue
            t = sd[hf]
            dst = self.v[name]
            perm = self.layout.row_perm.get(name)
            if perm is not None:
                t = t[perm.to(t.device)]
            if dst.dim() == 1:
                dst.copy_(t.to(BF16))
            else:
                dst[r0:r0 + rows].copy_(t.to(BF16))
            seen.add(hf)
        return seen

    def state_dict(self):
        out = {}
        for hf, name, r0, rows in self.layout.hf_names():
            v = self.v[name]
            perm = self.layout.row_perm.get(name)
            if perm is not None:                  # back to the checkpoint's row order (a copy, not a view)
                inv = torch.empty_like(perm)
                inv[perm] = torch.arange(perm.numel())
                out[hf] = v[inv.to(v.device)]
                continue
            out[hf] = v if v.dim() == 1 else v[r0:r0 + rows]
        return out


class VisionWeights:
    """Frozen CLIP ViT weights, bf16, q|k|v fused, patch-embedding kernel flattened and K-padded to a multiple of 8."""

    def __init__(self, cfg, sd, device, prefix="vision_tower.vision_model."):
        D, P = cfg["vit_hidden"], cfg["patch_size"]
        self.Kp = _align(3 * P * P, 8)
        dv = lambda t: t.to(BF16).to(device).contiguous()  # noqa: E731
        w = torch.zeros(D, self.Kp)
        w[:, : 3 * P * P] = sd[prefix + "embeddings.patch_embedding.weight"].float().reshape(D, -1)
        self.patch_w = dv(w)
        self.cls = dv(sd[prefix + "embeddings.class_embedding"].reshape(D))
        self.pos = dv(sd[prefix + "embeddings.position_embedding.weight"])
        self.pre_w, self.pre_b = dv(sd[prefix + "pre_layrnorm.weight"]), dv(sd[prefix + "pre_layrnorm.bias"])
        self.layers = []
        self._keep = []
        for i in range(cfg["vit_layers"] + 1 + int(cfg.get("vit_feature_layer", -2))):   # vision_feature_layer -2 (LLaVA): the last layer is never evaluated; -1: all
            p = f"{prefix}encoder.layers.{i}."
            t = dict(
                ln1_w=dv(sd[p + "layer_norm1.weight"]), ln1_b=dv(sd[p + "layer_norm1.bias"]),
                wqkv=dv(torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)),
                bqkv=dv(torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)),
                wo=dv(sd[p + "self_attn.out_proj.weight"]), bo=dv(sd[p + "self_attn.out_proj.bias"]),
                ln2_w=dv(sd[p + "layer_norm2.weight"]), ln2_b=dv(sd[p + "layer_norm2.bias"]),
                w1=dv(sd[p + "mlp.fc1.weight"]), b1=dv(sd[p + "mlp.fc1.bias"]),
                w2=dv(sd[p + "mlp.fc2.weight"]), b2=dv(sd[p + "mlp.fc2.bias"]))
            self._keep.append(t)
            self.layers.append(_hip.VitLayerWeights(*(t[k].data_ptr() for k in (
                "ln1_w", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2"))))


class LlavaHipEngine:
    custom_layers = False          # True: the subclass composes the decoder layer itself (_layer_forward / _hidden_backward_custom)
    supports_resid_f32 = True      # False: the subclass adds to the residual stream with bf16 primitives
    supports_ckpt = True           # False: the subclass's backward cannot re-run a layer's forward (gradient checkpointing is ignored)
    proj_out_f32 = True            # with the fp32 stream the projector writes fp32 rows for the merge (VLR_PROJ_F32=0: bf16)

    def __init__(self, cfg: dict, device="cuda", max_positions: int = 4096):
        if not torch.cuda.is_available():
            raise _hip.VlrError("LlavaHipEngine needs an MI355X (torch.cuda.is_available() is False); "
                                "there is no CPU fallback for the DPO hot path")
        _hip.lib()
        self.cfg = dict(cfg)
        self.dev = torch.device(device)
        c = self.cfg
        self.H, self.I, self.V, self.L = c["hidden"], c["inter"], c["vocab"], c["layers"]
        self.nh = c["heads"]
        self.nkv = c.get("kv_heads") or self.nh
        self.hd = c.get("head_dim") or self.H // self.nh
        if self.hd != 128:
            raise ValueError(f"decoder head_dim must be 128 for the gfx950 attention kernels, got {self.hd}")
        if self.nh % self.nkv:
            raise ValueError(f"heads ({self.nh}) must be a multiple of kv_heads ({self.nkv})")
        self.Nq, self.Nkv = self.nh * self.hd, self.nkv * self.hd
        self.Nqkv = self.Nq + 2 * self.Nkv
        self.anyres = bool(c.get("image_grid_pinpoints"))      # LLaVA-Next tiles
        self.D = c.get("vit_hidden", 8)
        self.P = (c["image_size"] // c["patch_size"]) ** 2 if "image_size" in c else 0
        self.layout = ParamLayout(c)
        self.max_pos = max_positions
        self.cos = torch.empty(max_positions, self.hd // 2, dtype=torch.float32, device=self.dev)
        self.sin = torch.empty_like(self.cos)
        _hip.call("vlr_rope_table", self.cos, self.sin, max_positions, self.hd, float(c.get("rope_theta", 10000.0)))
        # fp32 residual stream (include/vlr.h vlr_llama_cfg.resid_f32): x0 and every layer's x_mid / x_out are fp32, never rounded.
        # Default ON (VLR_RESID_F32=0 or cfg["resid_f32"] = False: the bf16 stream of ABI v3); engines that compose their own layers
        # from bf16 primitives opt out (supports_resid_f32).
        self.resid_f32 = bool(c.get("resid_f32", os.environ.get("VLR_RESID_F32", "1") != "0")) and self.supports_resid_f32
        self.RDT = torch.float32 if self.resid_f32 else BF16      # dtype of the residual stream
        self.proj_out_f32 = self.proj_out_f32 and os.environ.get("VLR_PROJ_F32", "1") != "0"
        # gradient checkpointing (reference scripts: --gradient_checkpointing True, dpo.py:99 non-reentrant): only the layer inputs
        # are kept by the forward; the backward re-runs each layer's forward into one scratch set right before its backward
        self.gradient_checkpointing = bool(c.get("gradient_checkpointing", False))
        self.llama_cfg = _hip.LlamaCfg(self.H, self.I, self.nh, self.hd, float(c.get("rms_eps", 1e-5)), max_positions,
                                       self.cos.data_ptr(), self.sin.data_ptr(), self.nkv, int(self.resid_f32))
        self._init_vision_cfg()
        self.vision: Optional[VisionWeights] = None
        self.policy = WeightSet(self.layout, self.dev)
        self.grads = torch.zeros(self.layout.numel, dtype=BF16, device=self.dev)
        self.gv = {n: self.grads[self.layout.offset[n]: self.layout.offset[n] + int(math.prod(s))].view(*s)
                   for n, s in self.layout.shape.items()}
        self.master = self.m = self.v = None     # fp32 optimizer state, allocated by init_optimizer()
        self.opt_step = 0
        self._weights_version = 0                # bumped whenever weights are (re)loaded: keys caches derived from them
        self.vision_sd = {}
        self.grad_fresh = True                   # next backward overwrites instead of accumulating
        self._ws = {}
        self._vit_cache = None
        self.reducer = None                      # parallel.GradReducer for DDP
        self.lora = None                         # dict(r, scale, dropout) once enable_lora() ran
        self.lora_active = True                  # False inside LlavaForRL.disable_adapter() (reference pass)
        self.training = True                     # lora_dropout only in training mode
        self._norm_ws = torch.empty(_hip.helper("vlr_rmsnorm_bwd_workspace_bytes", self.H), dtype=torch.uint8, device=self.dev)
        self._colsum_ws = torch.empty(_hip.helper("vlr_colsum_workspace_bytes", max(self.H, self.Nqkv, 8)), dtype=torch.uint8, device=self.dev)
        self._sq_ws = torch.empty(_hip.helper("vlr_grad_sqnorm_workspace_bytes"), dtype=torch.uint8, device=self.dev)
        self.norm_out = torch.zeros(3, dtype=torch.float32, device=self.dev)
        # Optional (VLR_ASYNC_OPT=1): clip + AdamW (HBM-bound, 32 ms) on their own stream so that they overlap the next step's
        # FROZEN reference forward; the policy forward waits for the `_opt_done` event.  Measured on MI355X: no gain (650.7 /
        # 652.6 ms without vs 648.5 / 653.1 ms with) - the two kernels do not co-schedule - so it is off by default.
        self._opt_stream = torch.cuda.Stream(self.dev) if os.environ.get("VLR_ASYNC_OPT", "0") == "1" else None
        self._opt_done = None
        # split-K scratch of the GEMM dispatcher (ragged last tile rows, LoRA adapter gradients): two 64 MiB slots (main + side stream)
        _hip.ensure_splitk_workspace(self.dev, force=True)      # a new engine brings new streams: forget the old slot assignment

    # ------------------------------------------------------------------------------------------------ weights
    vision_prefix = "vision_tower."

    def _init_vision_cfg(self):
        c = self.cfg
        self.vit_cfg = _hip.VitCfg(self.D, c["vit_mlp"], c["vit_heads"], self.D // c["vit_heads"],
                                   float(c.get("vit_ln_eps", 1e-5)))
        if self.D // c["vit_heads"] != 64:
            raise ValueError("ViT head_dim must be 64 for the gfx950 attention kernels")

    def _load_vision(self, sd):
        return VisionWeights(self.cfg, sd, self.dev, prefix=self.vision_prefix + "vision_model.")

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        self.policy.load_state_dict(sd)
        self.vision = self._load_vision(sd)
        # the frozen tower's ORIGINAL tensors (bf16, ~0.6 GB for CLIP-L): save_pretrained writes them back so that the
        # output directory reloads (reference trainer._save writes the whole model)
        self.vision_sd = {k: v.detach().to(device=self.dev, dtype=BF16) for k, v in sd.items() if k.startswith(self.vision_prefix)}
        self._vit_cache = None
        self._weights_version += 1
        if self.master is not None:
            self.init_optimizer()

    def layer_weights(self, ws: WeightSet, l):
        v = ws.v
        return _hip.LayerWeights(*(v[f"l{l}.{k}"].data_ptr() for k in ("ln1", "wqkv", "wo", "ln2", "wgu", "wdown")),
                                 v[f"l{l}.bqkv"].data_ptr() if f"l{l}.bqkv" in v else None)

    def layer_grads(self, l):
        return _hip.LayerGrads(*(self.gv[f"l{l}.{k}"].data_ptr() for k in ("ln1", "wqkv", "wo", "ln2", "wgu", "wdown")))

    # ------------------------------------------------------------------------------------------------ LoRA
    def enable_lora(self, r: int, alpha: float, dropout: float = 0.0, seed: int = 0):
        """peft get_peft_model(LoraConfig(r, lora_alpha, lora_dropout, target_modules=default_lora_target, bias='none')):
        base weights frozen, A ~ kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)), B = 0.  From here on the only
        trainable parameters (gradients, optimizer state, DDP bucket) are the adapters."""
        if r <= 0 or r % 8:
            raise ValueError(f"lora_r must be a positive multiple of 8 for the gfx950 GEMM tiles, got {r}")
        if self.Nq != self.H:
            raise NotImplementedError("LoRA needs heads * head_dim == hidden_size on the MI355X path")
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"lora_dropout must be in [0, 1), got {dropout}")
        self.lora = dict(r=int(r), scale=float(alpha) / r, dropout=float(dropout), alpha=float(alpha))
        self.lora_layout = LoraLayout(self.cfg, r)
        n = self.lora_layout.numel
        self.lora_flat = torch.zeros(n, dtype=BF16, device=self.dev)
        self.lora_grads = torch.zeros(n, dtype=BF16, device=self.dev)
        view = lambda flat: {k: flat[o: o + int(math.prod(self.lora_layout.shape[k]))].view(*self.lora_layout.shape[k])
                             for k, o in self.lora_layout.offset.items()}
        self.lv, self.lgv = view(self.lora_flat), view(self.lora_grads)
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(seed)
        for k, t in self.lv.items():
            if ".a_" in k:
                bound = 1.0 / math.sqrt(t.shape[1])
                t.copy_((torch.rand(t.shape, generator=gen, device=self.dev) * 2 - 1) * bound)
        self.lora_seed = int(seed)
        self._lora_calls = 0

        self.grads = None                         # full-parameter gradient / optimizer buffers are not needed any more
        self.gv = None
        self.master = self.m = self.v = None
        self.opt_step = 0
        self.grad_fresh = True

    def lora_state_dict(self):
        """adapter tensors under their peft adapter-file names (adapter_model.safetensors layout)"""
        out = {}
        for n, (k, lo, hi) in self.lora_layout.hf_names().items():
            t = self.lv[k][lo:hi].clone()
            perm = self.lora_layout.row_perm.get(k.split(".", 1)[1])
            if perm is not None:                  # back to the checkpoint's row order
                inv = torch.empty_like(perm)
                inv[perm] = torch.arange(perm.numel())
                t = t[inv.to(t.device)]
            out[n] = t
        return out

    def load_lora_state_dict(self, sd):
        names = self.lora_layout.hf_names()
        norm = {k.replace(".default.weight", ".weight"): v for k, v in sd.items()}
        missing = [n for n in names if n not in norm]
        if missing:
            raise KeyError(f"missing LoRA tensors: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        for n, (k, lo, hi) in names.items():
            t = norm[n]
            if tuple(t.shape) != tuple(self.lv[k][lo:hi].shape):
                raise ValueError(f"shape mismatch for {n}: {tuple(t.shape)} vs {tuple(self.lv[k][lo:hi].shape)}")
            perm = self.lora_layout.row_perm.get(k.split(".", 1)[1])
            if perm is not None:
                t = t[perm.to(t.device)]
            self.lv[k][lo:hi].copy_(t)
        if self.master is not None:
            self.init_optimizer()

    def _mask_bits(self, l, M, p, acts=None, key="lora_bits"):
        """packed lora_dropout keep masks of layer l (include/vlr.h vlr_lora_weights::mask_bits): drawn by the layer's forward, read by
        its adapter GEMMs and again by its backward - 57 MB per layer at the 7B shapes.  The buffer lives IN the activation set of the
        pass that drew it (`acts`: per tag and layer, or the one shared set of a checkpointed pass), so a second training-mode forward
        of the same size cannot overwrite masks a pending backward will read, and checkpointing keeps one buffer instead of one per
        layer.  VLR_LORA_BITS=0: every kernel hashes instead."""
        if p <= 0.0 or os.environ.get("VLR_LORA_BITS", "1") == "0" or (M * self.H) % 32 or (M * self.I) % 32:
            return None
        n = _hip.helper("vlr_lora_mask_bytes", self.H, self.I, M)
        if acts is None:
            return self._buf((key, l, M), (n,), torch.uint8).data_ptr()
        sh = acts.get("shared", acts)
        t = sh.get(key)
        if t is None or t.numel() != n:
            t = sh[key] = torch.empty(n, dtype=torch.uint8, device=self.dev)
        return t.data_ptr()

    def _lora_structs(self, l, train, M=None, acts=None):
        lo = self.lora
        p = lo["dropout"] if (train and self.training) else 0.0
        ptr = lambda views, k: views[f"l{l}.{k}"].data_ptr() if f"l{l}.{k}" in views else None   # noqa: E731  (no down adapter: NULL)
        w = _hip.LoraWeights(lo["r"], lo["scale"], p, *(ptr(self.lv, k) for k in LORA_KEYS), self.lora_layout.qkv_targets,
                             self._mask_bits(l, M, p, acts) if M else None)
        g = _hip.LoraGrads(*(ptr(self.lgv, k) for k in LORA_KEYS))
        return w, g

    def merged_weights(self) -> WeightSet:
        """W + (alpha/r) B A for every adapted linear (peft merge_and_unload): a new WeightSet for export / inference."""
        ws = self.policy.clone()
        r, sc = self.lora["r"], self.lora["scale"]
        for l in range(self.L):
            for g, _, targets in self.lora_layout.groups:
                W = ws.v[f"l{l}.w{g}"]
                A, B = self.lv[f"l{l}.a_{g}"], self.lv[f"l{l}.b_{g}"]
                inn = W.shape[1]
                row = 0
                for t, name in enumerate(targets):
                    out = self.lora_layout.out_dim[name]
                    Wt = W[row:row + out]
                    _hip.call("vlr_gemm_bf16_scaled", 1, B[row:row + out], A[t * r:(t + 1) * r], Wt, None, Wt,
                              out, inn, r, r, inn, inn, inn, 0, 0, 0, sc)
                    row += out
        return ws

    # ------------------------------------------------------------------------------------------------ workspaces
    def _buf(self, key, shape, dtype=BF16, zero=False):
        t = self._ws.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.dev)
            self._ws[key] = t
        return t

    def _layer_acts(self, tag, l, Bn, S):
        M, H, I = Bn * S, self.H, self.I
        Sp = _align(S, 64)
        k = (tag, l, Bn, S)
        t = self._ws.get(k)
        if t is None:
            t = dict(xn1=torch.empty(M, H, dtype=BF16, device=self.dev), rstd1=torch.empty(M, dtype=torch.float32, device=self.dev),
                     qkv=torch.empty(M, self.Nqkv, dtype=BF16, device=self.dev), attn=torch.empty(M, self.Nq, dtype=BF16, device=self.dev),
                     lse=torch.empty(Bn, self.nh, Sp, dtype=torch.float32, device=self.dev),
                     x_mid=torch.empty(M, H, dtype=self.RDT, device=self.dev), xn2=torch.empty(M, H, dtype=BF16, device=self.dev),
                     rstd2=torch.empty(M, dtype=torch.float32, device=self.dev), gu=torch.empty(M, 2 * I, dtype=BF16, device=self.dev),
                     act=torch.empty(M, I, dtype=BF16, device=self.dev), x_out=torch.empty(M, H, dtype=self.RDT, device=self.dev))
            t["struct"] = _hip.LayerActs(*(t[n].data_ptr() for n in ("xn1", "rstd1", "qkv", "attn", "lse", "x_mid", "xn2", "rstd2", "gu", "act", "x_out")))
            self._ws[k] = t
        return t

    def _ckpt_acts(self, tag, l, Bn, S):
        """gradient checkpointing: every layer shares ONE scratch activation set; only x_out (= the next layer's input) is per layer"""
        k = (tag, "ckpt", l, Bn, S)
        t = self._ws.get(k)
        if t is None:
            base = self._layer_acts(tag + "/ckpt", 0, Bn, S)
            t = {n: v for n, v in base.items() if n != "struct"}
            t["x_out"] = torch.empty(Bn * S, self.H, dtype=self.RDT, device=self.dev)
            t["struct"] = _hip.LayerActs(*(t[n].data_ptr() for n in ("xn1", "rstd1", "qkv", "attn", "lse", "x_mid", "xn2", "rstd2", "gu", "act", "x_out")))
            t["shared"] = base            # LoRA: u / xd live in the shared set
            self._ws[k] = t
        return t

    def _norm_fwd(self, x, w, y, rstd, M):
        _hip.call("vlr_rmsnorm_fwd_f32" if x.dtype == torch.float32 else "vlr_rmsnorm_fwd", x, w, y, rstd, M, self.H, self.llama_cfg.rms_eps)

    def _norm_bwd(self, dy, x, w, rstd, dres, dx, dw, acc, M):
        _hip.call("vlr_rmsnorm_bwd_f32" if x.dtype == torch.float32 else "vlr_rmsnorm_bwd", dy, x, w, rstd, dres, dx, dw, acc, self._norm_ws, M, self.H)

    def _layer_fwd_call(self, ws, l, a, x, e, Bn, S, keep, use_lora, lora_seed):
        """one decoder layer forward into the activation set `a` (keep: also write what only the backward reads)"""
        M = Bn * S
        if self.custom_layers:
            self._layer_forward(ws, l, a, x, e, Bn, S, keep, use_lora, lora_seed)
        elif use_lora:
            r = self.lora["r"]
            sh = a.get("shared", a)
            if "u" not in sh or sh["u"].shape[1] != 7 * r:
                sh["u"] = torch.empty(M, 7 * r, dtype=BF16, device=self.dev)
            lw, _ = self._lora_structs(l, train=True, M=M, acts=a)
            # (lora_dropout: the keep mask is applied to x while the adapter GEMMs stage it and regenerated in the backward - no dropped
            # copies of the seven inputs are kept any more: 0.9 GB per layer at the 7B shapes)
            _hip.call("vlr_decoder_layer_fwd_lora", self.llama_cfg, self.layer_weights(ws, l), lw, a["struct"], sh["u"], None,
                      lora_seed + 8 * l, x, e["pos"], e["mask"], Bn, S)
        else:
            _hip.call("vlr_decoder_layer_fwd_ex", self.llama_cfg, self.layer_weights(ws, l), a["struct"], x, e["pos"], e["mask"], Bn, S,
                      int(keep))

    # ------------------------------------------------------------------------------------------------ vision
    def vision_features(self, pixel_values: torch.Tensor, key=None) -> torch.Tensor:
        """CLIP ViT hidden_states[-2] without CLS -> [n*P, D] bf16 (frozen tower: cached per pixel_values tensor so the
        reference pass and the policy pass share one evaluation).  `pixel_values` may be a callable producing the [n,3,s,s]
        tensor (LLaVA-Next selects the evaluated tiles of a padded 5-D batch) - it is only called on a cache miss."""
        if key is None:
            key = (pixel_values.data_ptr(), tuple(pixel_values.shape), pixel_values._version)
        if self._vit_cache is not None and self._vit_cache[0] == key:
            return self._vit_cache[1]
        if callable(pixel_values):
            pixel_values = pixel_values()
        c = self.cfg
        n = pixel_values.shape[0]
        g = c["image_size"] // c["patch_size"]
        T = g * g + 1
        D, F = self.D, c["vit_mlp"]
        vw = self.vision
        pv = pixel_values.to(device=self.dev, dtype=torch.float32).contiguous()
        patches = self._buf(("vit_patches", n), (n * g * g, vw.Kp))
        _hip.call("vlr_im2col", pv, patches, n, c["image_size"], c["patch_size"], vw.Kp)
        pe = self._buf(("vit_pe", n), (n * g * g, D))
        _hip.call("vlr_gemm_bf16", 0, patches, vw.patch_w, pe, None, None, n * g * g, D, vw.Kp, vw.Kp, vw.Kp, D, 0, 0, 0, 0)
        x = self._buf(("vit_x", n), (n * T, D))
        _hip.call("vlr_vit_embed_ln", pe, vw.cls, vw.pos, vw.pre_w, vw.pre_b, x, n, T, D, self.vit_cfg.ln_eps)
        wsb = dict(xn=self._buf(("vit_xn", n), (n * T, D)), qkv=self._buf(("vit_qkv", n), (n * T, 3 * D)),
                   attn=self._buf(("vit_attn", n), (n * T, D)), h=self._buf(("vit_h", n), (n * T, F)))
        ws = _hip.VitWs(*(wsb[k].data_ptr() for k in ("xn", "qkv", "attn", "h")))
        for lw in vw.layers:
            _hip.call("vlr_vit_layer_fwd", self.vit_cfg, lw, ws, x, n, T)
        rows = self._ws.get(("vit_rows", n))
        if rows is None:
            rows = (torch.arange(n * T, device=self.dev, dtype=torch.int32).view(n, T)[:, 1:]).reshape(-1).contiguous()
            self._ws[("vit_rows", n)] = rows
        feat = torch.empty(n * (T - 1), D, dtype=BF16, device=self.dev)
        _hip.call("vlr_gather_rows", x, rows, feat, n * (T - 1), D)
        self._vit_cache = (key, feat, pixel_values)
        return feat

    def anyres_vision_features(self, pixel_values, image_sizes, image_dup=1):
        """LLaVA-Next: the first num_patches(image_size) tiles of every DISTINCT image through the frozen ViT (cached like
        vision_features).  -> (features [tiles*P, D], sizes of the distinct images, tiles per image)"""
        from .models.LlavaNext import anyres as AR
        c = self.cfg
        if image_sizes is None:
            raise ValueError("LLaVA-Next forward needs image_sizes")
        n_img = pixel_values.shape[0]
        if n_img % image_dup:
            raise ValueError(f"{n_img} images cannot be {image_dup} identical halves")
        sizes = [tuple(int(v) for v in sz) for sz in (image_sizes.tolist() if isinstance(image_sizes, torch.Tensor) else image_sizes)]
        if len(sizes) != n_img:
            raise ValueError(f"{len(sizes)} image_sizes for {n_img} images")
        uniq = pixel_values[: n_img // image_dup]
        usz = sizes[: n_img // image_dup]
        npatch = [AR.image_size_to_num_patches(sz, c["image_grid_pinpoints"], c["image_size"]) for sz in usz]
        if uniq.dim() == 5:
            if max(npatch) > uniq.shape[1]:
                raise ValueError(f"pixel_values holds {uniq.shape[1]} tiles per image, image_sizes need {max(npatch)}")
            flat = lambda: torch.cat([uniq[i, :k] for i, k in enumerate(npatch)], dim=0)   # noqa: E731
        elif uniq.dim() == 4:
            flat = lambda: uniq                                                              # noqa: E731
        else:
            raise ValueError(f"pixel_values of shape {tuple(pixel_values.shape)}, expect to be of 4 or 5 dimensions")
        vit_feat = self.vision_features(flat, key=(pixel_values.data_ptr(), tuple(pixel_values.shape), pixel_values._version, tuple(usz)))
        if vit_feat.shape[0] != sum(npatch) * self.P:
            raise ValueError(f"{vit_feat.shape[0] // self.P} image tiles given, image_sizes need {sum(npatch)}")
        return vit_feat, usz, npatch

    def projector_fwd(self, ws: WeightSet, vit_feat, tag, extra_rows=0):
        R, H, D = vit_feat.shape[0], self.H, self.D
        z = self._buf((tag, "proj_z", R), (R, H))
        h = self._buf((tag, "proj_h", R), (R, H))
        # fp32 residual stream: the projector's output = the image rows of the merged embeddings stays fp32 (never rounded to bf16)
        f32 = self.resid_f32 and self.proj_out_f32
        out = self._buf((tag, "proj_out", R, extra_rows, f32), (R + extra_rows, H), torch.float32 if f32 else BF16)    # LLaVA-Next appends the image_newline row
        _hip.call("vlr_gemm_bf16", 0, vit_feat, ws.v["proj.w1"], z, ws.v["proj.b1"], None, R, H, D, D, D, H, 0, 0, 0, 0)
        _hip.call("vlr_gelu_fwd", z, h, z.numel())
        _hip.call("vlr_gemm_bf16", 0, h, ws.v["proj.w2"], out, ws.v["proj.b2"], None, R, H, H, H, H, H, 0, 0, 0, int(f32))
        return out, z, h

    # ------------------------------------------------------------------------------------------------ forward
    def _embed_inputs(self, ws, ids, am, lab, pixel_values, image_dup, tag, image_sizes, meta):
        """vision tower -> projector -> merge index: everything in front of the decoder.  Returns the merged-sequence geometry
        (S, src / inv maps, mask, positions, merged labels, image map) and the feature rows `feats` the merge gathers from."""
        c = self.cfg
        Bn, T = ids.shape
        n_img = pixel_values.shape[0]
        if image_dup > 1:
            assert n_img % image_dup == 0
            uniq = pixel_values[: n_img // image_dup]
        else:
            uniq = pixel_values
        pack = None
        if self.anyres:
            # ---- LLaVA-Next (reference LlavaNext/__init__.py:205-265): tiles -> ViT -> projector -> anyres pack (row gather)
            from .models.LlavaNext import anyres as AR
            vit_feat, usz, npatch = self.anyres_vision_features(pixel_values, image_sizes, image_dup)
            n_rows = vit_feat.shape[0]
            ext, z, h = self.projector_fwd(ws, vit_feat, tag, extra_rows=1)
            ext[n_rows].copy_(ws.v["image_newline"])
            cached = meta.get("anyres") if meta is not None else None
            if cached is None:
                pidx, lens, nl_pos = AR.pack_index(usz, npatch, c["image_grid_pinpoints"], c["image_size"], c["patch_size"])
                mi = AR.merge_index(ids.cpu().numpy(), am.cpu().numpy(), lab.cpu().numpy() if lab is not None else None,
                                    list(lens) * image_dup, int(c["image_token"]), c.get("padding_side", "left"), dup=image_dup)
                dv = lambda a_: torch.from_numpy(a_).to(self.dev)    # noqa: E731
                cached = dict(S=mi["S"], src=dv(mi["src"]), mask=dv(mi["mask"]), labels=dv(mi["labels"]), pos=dv(mi["pos"]),
                              img_map=dv(mi["img_map"]), inv=dv(mi["inv"]),
                              pack=dict(idx=dv(pidx), nl=dv(nl_pos), F=int(pidx.shape[0]), rows=n_rows, feature_lens=lens))
                if meta is not None:
                    meta["anyres"] = cached
            pack = cached["pack"]
            F = pack["F"]
            feats = self._buf((tag, "packed", F, ext.dtype), (F, self.H), ext.dtype)
            _hip.call("vlr_gather_rows", ext, pack["idx"], feats, F, self.H * (2 if ext.dtype == torch.float32 else 1))   # (row copy: fp32 rows = 2H 16-bit columns)
            S = cached["S"]
            M = Bn * S
            src, mask, pos, img_map, inv = cached["src"], cached["mask"], cached["pos"], cached["img_map"], cached["inv"]
            mlabels = cached["labels"].clone()                 # handed out as `output.labels`: every pass gets its own tensor
            n_feat = F
        else:
            vit_feat = self.vision_features(uniq)
            feats, z, h = self.projector_fwd(ws, vit_feat, tag)
            n_rows = feats.shape[0]
            n_feat = n_rows
            P = self.P
            n_img_tok = (ids == c["image_token"]).sum(-1)
            if meta is not None and "S" in meta:
                S = meta["S"]
            else:
                S = int(n_img_tok.max()) * (P - 1) + T                   # one small D2H sync (shape of the merged batch)
            M = Bn * S
            src = self._buf((tag, "src", Bn, S), (Bn, S), torch.int32)
            mask = torch.empty(Bn, S, dtype=torch.int32, device=self.dev)
            mlabels = torch.empty(Bn, S, dtype=torch.int64, device=self.dev)
            pos = torch.empty(Bn, S, dtype=torch.int32, device=self.dev)
            img_map = torch.empty(Bn, S, dtype=torch.uint8, device=self.dev)
            inv = self._buf((tag, "inv", image_dup, n_rows), (image_dup, n_rows), torch.int32)
            info = torch.zeros(2, dtype=torch.int32, device=self.dev)
            _hip.call("vlr_merge_index", ids, am, lab, Bn, T, S, P, int(c["image_token"]),
                      int(c.get("model_pad_token_id", c["image_token"] + 1)), n_rows, image_dup, src, mask, mlabels, pos,
                      img_map, inv, info)
            found = n_rows * image_dup if (meta is not None and meta.get("merge_ok")) else int(info[0])   # checked once per batch
            if meta is not None:
                meta["S"], meta["merge_ok"] = S, found == n_rows * image_dup
            if found != n_rows * image_dup:
                raise ValueError(
                    f"The input provided to the model are wrong. The number of image tokens is {int(n_img_tok.sum())} while"
                    f" the number of image given to the model is {n_img}. This prevents correct indexing and breaks batch"
                    " generation.")
        return dict(S=S, M=M, src=src, mask=mask, pos=pos, labels=mlabels, img_map=img_map, inv=inv, feats=feats, vit_feat=vit_feat,
                    proj_z=z, proj_h=h, n_rows=n_rows, n_feat=n_feat, pack=pack)

    def forward_hidden(self, ws: WeightSet, input_ids, attention_mask, labels, pixel_values, image_dup=1, save=False,
                       tag="ref", image_sizes=None):
        """embed -> ViT -> projector -> merge -> decoder -> final RMSNorm.  Returns a context dict with the final
        hidden states [Bn*S, H] and the merged labels / mask / positions."""
        c = self.cfg
        if ws is self.policy:
            self.wait_optimizer()
        Bn, T = input_ids.shape
        meta = getattr(input_ids, "_vlr_meta", None)      # per-batch cache of host-side integers (None: always recompute)
        ids = input_ids.to(self.dev).contiguous()
        am = attention_mask.to(self.dev).contiguous()
        lab = labels.to(self.dev).contiguous() if labels is not None else None
        e = self._embed_inputs(ws, ids, am, lab, pixel_values, image_dup, tag, image_sizes, meta)
        e["tag"] = tag               # scratch buffers of custom layers are keyed per pass (the reference pass runs on a side stream)
        e["grad_pass"] = bool(save)  # this pass will be back-propagated (training-mode dropout applies; a checkpointed forward included)
        S, M = e["S"], e["M"]
        src, mask, pos, mlabels, img_map, inv = e["src"], e["mask"], e["pos"], e["labels"], e["img_map"], e["inv"]
        feats, vit_feat, z, h, n_rows, n_feat, pack = e["feats"], e["vit_feat"], e["proj_z"], e["proj_h"], e["n_rows"], e["n_feat"], e["pack"]
        if self.resid_f32:                 # fp32 stream: embedding rows widened exactly, the projector's output rows unrounded (fp32 feats)
            x0 = self._buf((tag, "x0f", Bn, S), (M, self.H), torch.float32)
            _hip.call("vlr_merge_fwd_f32", src, ids, ws.v["embed"], feats, int(feats.dtype == torch.float32), x0, Bn, T, S, self.H)
        else:
            x0 = self._buf((tag, "x0", Bn, S), (M, self.H))
            _hip.call("vlr_merge_fwd", src, ids, ws.v["embed"], feats, x0, Bn, T, S, self.H)
        x = x0
        acts = []
        use_lora = self.lora is not None and self.lora_active and ws is self.policy
        lora_seed = None
        if use_lora:
            self._lora_calls += 1
            lora_seed = (self.lora_seed << 40) + (self._lora_calls << 16)        # + 8*layer + target inside the library
        ckpt = bool(save and self.gradient_checkpointing and self.supports_ckpt)      # (layers composed in Python keep their activations)
        for l in range(self.L):
            if ckpt:
                a = self._ckpt_acts(tag, l, Bn, S)
            else:
                a = self._layer_acts(tag if save else tag + "/scratch", l if save else (l % 2), Bn, S)   # scratch per pass tag (side stream)
            # checkpointing: this pass keeps x_out only - what the backward reads is written by the recompute (hidden_backward)
            self._layer_fwd_call(ws, l, a, x, e, Bn, S, save and not ckpt, use_lora, lora_seed)
            acts.append(a)
            x = a["x_out"]
        hidden = torch.empty(M, self.H, dtype=BF16, device=self.dev)
        rstd_f = self._buf((tag, "rstd_f", M), (M,), torch.float32)
        self._norm_fwd(x, ws.v["norm"], hidden, rstd_f, M)
        return dict(ws=ws, Bn=Bn, T=T, S=S, M=M, ids=ids, src=src, inv=inv, mask=mask, labels=mlabels, pos=pos,
                    img_map=img_map.bool(), hidden=hidden, rstd_f=rstd_f, x_last=x, x0=x0, acts=acts if save else None,
                    vit_feat=vit_feat, feats=feats, proj_z=z, proj_h=h, image_dup=image_dup, n_rows=n_rows, n_feat=n_feat,
                    pack=pack, tag=tag, lora_seed=lora_seed, meta=meta, extra=e.get("extra"), ckpt=ckpt, use_lora=use_lora,
                    embed=dict(pos=e["pos"], mask=e["mask"], extra=e.get("extra"), tag=tag, img_map=e["img_map"], plora_seed=e.get("plora_seed"), grad_pass=bool(save)))

    # ------------------------------------------------------------------------------------------------ log-probs
    def logps_forward(self, ctx, labels, shared_mask=None, average=False, label_pad=-100):
        """get_batch_logps on the lm-head restricted to the response rows (identical result: every other row is
        multiplied by a zero mask in the reference, base/trainer.py:185-188).  Returns (logps [Bn], lp_ctx)."""
        Bn, S, M, H, V = ctx["Bn"], ctx["S"], ctx["M"], self.H, self.V
        ws = ctx["ws"]
        if tuple(labels.shape) != (Bn, S):
            raise ValueError("Logits (batch and sequence length dim) and labels must have the same shape.")
        lab = labels.to(self.dev).contiguous()
        rows = torch.empty(M, dtype=torch.int32, device=self.dev)
        tgt = torch.empty(M, dtype=torch.int32, device=self.dev)
        seq_off = torch.empty(Bn + 1, dtype=torch.int32, device=self.dev)
        sm = shared_mask.to(device=self.dev, dtype=torch.uint8).contiguous() if shared_mask is not None else None
        _hip.call("vlr_build_rows", lab, sm, Bn, S, label_pad, rows, tgt, seq_off)
        meta, rkey = ctx.get("meta"), ("R", sm is not None, int(label_pad))
        if meta is not None and rkey in meta:
            R = meta[rkey]
        else:
            R = int(seq_off[-1])                                      # one small D2H sync (row count of the lm-head GEMM)
            if meta is not None:
                meta[rkey] = R
        logps = torch.zeros(Bn, dtype=torch.float32, device=self.dev)
        lp = dict(R=R, rows=rows, tgt=tgt, seq_off=seq_off, average=average, ctx=ctx)
        if R == 0:
            return logps, lp
        hg = torch.empty(R, H, dtype=BF16, device=self.dev)
        _hip.call("vlr_gather_rows", ctx["hidden"], rows, hg, R, H)
        # fused lm-head + log-softmax pick: at the 7B shapes the [R][V] logits never reach HBM (include/vlr.h); small shapes go
        # through an fp32 logits buffer (per pass tag: the reference pass runs on a side stream)
        fused = bool(_hip.helper("vlr_lmhead_is_fused", R, V, H))
        lws = self._buf(("lmhead_ws", ctx["tag"], R), (int(_hip.lib().vlr_lmhead_workspace_bytes(R, V)),), torch.uint8)
        logits = None if fused else self._buf(("logits", ctx["tag"], R), (R, V), torch.float32)
        tok = torch.empty(R, dtype=torch.float32, device=self.dev)
        lse = torch.empty(R, dtype=torch.float32, device=self.dev)
        _hip.call("vlr_lmhead_logps_fwd", hg, ws.v["lm_head"], tgt, tok, lse, lws, logits, R, V, H)
        _hip.call("vlr_seq_sum", tok, seq_off, Bn, int(average), logps)
        lp.update(hg=hg, lse=lse, tok=tok)
        return logps, lp

    def logits_mean(self, ctx, lo, hi):
        """mean over [lo:hi] sequences, all positions, all vocabulary entries of the logits = mean_rows(h . sum_v W_v)/V
        (the `logits/chosen|rejected` metrics of trl's get_batch_loss_metrics, never materialising [B,S,V])."""
        ws = ctx["ws"]
        # keyed on the optimizer step / load counter: vlr_adamw_step writes the weights through raw pointers, so torch's
        # tensor version counter never moves
        key = ("wsum", id(ws), self.opt_step if ws is self.policy else -1, (self._weights_version, ws.version))
        wsum = self._ws.get(key)
        if wsum is None:
            wsum = torch.empty(self.H, dtype=torch.float32, device=self.dev)
            _hip.call("vlr_colsum_f32", ws.v["lm_head"], self.V, self.H, self.H, wsum, self._colsum_ws)
            self._ws = {k: v for k, v in self._ws.items() if not (isinstance(k, tuple) and len(k) == 4 and k[0] == "wsum" and k[1] == id(ws))}
            self._ws[key] = wsum
        S = ctx["S"]
        rd = torch.empty(ctx["M"], dtype=torch.float32, device=self.dev)
        _hip.call("vlr_rowdot", ctx["hidden"], wsum, rd, ctx["M"], self.H)
        return rd[lo * S: hi * S].mean() / self.V

    def materialize_logits(self, ctx, lo=0, hi=None):
        """Full fp32 logits [hi-lo, S, V] (debug / small shapes / callers that insist on a tensor)."""
        hi = ctx["Bn"] if hi is None else hi
        S = ctx["S"]
        n = (hi - lo) * S
        out = torch.empty(n, self.V, dtype=torch.float32, device=self.dev)
        hsl = ctx["hidden"][lo * S: hi * S]
        _hip.call("vlr_gemm_bf16", 0, hsl, ctx["ws"].v["lm_head"], out, None, None, n, self.V, self.H, self.H, self.H, self.V, 0, 0, 0, 1)
        return out.view(hi - lo, S, self.V)

    # ------------------------------------------------------------------------------------------------ backward
    def logps_backward(self, lp, dlogps):
        """d logps -> d hidden (dense [M,H], zero outside the response rows) and the lm_head weight gradient."""
        ctx = lp["ctx"]
        M, H, V, R = ctx["M"], self.H, self.V, lp["R"]
        acc = int(not self.grad_fresh)
        dhidden = torch.zeros(M, H, dtype=BF16, device=self.dev)
        if R == 0:
            if not acc and self.lora is None:
                self.gv["lm_head"].zero_()
            return dhidden
        # d logits (bf16 [R][V]): the lm-head GEMM is recomputed and its epilogue writes the gradient directly (fused shapes)
        fused = bool(_hip.helper("vlr_lmhead_is_fused", R, V, H))
        lws = self._buf(("lmhead_ws", ctx["tag"], R), (int(_hip.lib().vlr_lmhead_workspace_bytes(R, V)),), torch.uint8)
        logits = None if fused else self._buf(("logits", ctx["tag"], R), (R, V), torch.float32)
        dl = self._buf(("dlogits", R), (R, V))
        _hip.call("vlr_lmhead_logps_bwd", lp["hg"], ctx["ws"].v["lm_head"], lp["tgt"], lp["lse"], lp["seq_off"], ctx["Bn"],
                  dlogps.to(torch.float32).contiguous(), int(lp["average"]), dl, lws, logits, R, V, H)
        dhg = torch.empty(R, H, dtype=BF16, device=self.dev)
        _hip.call("vlr_gemm_bf16", 1, dl, ctx["ws"].v["lm_head"], dhg, None, None, R, H, V, V, H, H, 0, 0, 0, 0)
        if self.lora is None:                 # under LoRA the lm_head is frozen (not a target module)
            _hip.call("vlr_gemm_bf16", 2, dl, lp["hg"], self.gv["lm_head"], None, None, V, H, R, V, H, H, 0, 0, acc, 0)
        _hip.call("vlr_scatter_rows", dhg, lp["rows"], dhidden, R, H)
        if self.reducer is not None and self.lora is None:
            self.reducer.bucket_ready("lm_head")
        return dhidden

    def hidden_backward(self, ctx, dhidden):
        """Backward of forward_hidden(save=True): final norm, decoder layers L-1..0, merge, projector.  Gradients
        land in the flat bf16 gradient buffer (overwrite when `grad_fresh`, else accumulate)."""
        assert ctx["acts"] is not None, "forward_hidden(save=True) required"
        ws = ctx["ws"]
        Bn, S, M, H, I = ctx["Bn"], ctx["S"], ctx["M"], self.H, self.I
        acc = int(not self.grad_fresh)
        Sp = _align(S, 64)
        dxa = self._buf(("dxa", M), (M, H))
        dxb = self._buf(("dxb", M), (M, H))
        if self.custom_layers:
            return self._hidden_backward_custom(ctx, dhidden, dxa, dxb)
        if self.lora is not None:
            return self._hidden_backward_lora(ctx, dhidden, dxa, dxb)
        self._norm_bwd(dhidden, ctx["x_last"], ws.v["norm"], ctx["rstd_f"], None, dxa, self.gv["norm"], acc, M)
        wsb = dict(dact=self._buf(("dact", M), (M, I)), dxn=self._buf(("dxn", M), (M, H)), dattn=self._buf(("dattn", M), (M, self.Nq)),
                   dqkv=self._buf(("dqkv", M), (M, self.Nqkv)), dx_mid=self._buf(("dx_mid", M), (M, H)),
                   delta=self._buf(("delta", Bn, S), (Bn, self.nh, Sp), torch.float32))
        lws = _hip.LayerBwdWs(wsb["dact"].data_ptr(), wsb["dxn"].data_ptr(), wsb["dattn"].data_ptr(), wsb["dqkv"].data_ptr(),
                              wsb["dx_mid"].data_ptr(), wsb["delta"].data_ptr(), self._norm_ws.data_ptr())
        cur, nxt = dxa, dxb
        for l in range(self.L - 1, -1, -1):
            a = ctx["acts"][l]
            x_in = ctx["acts"][l - 1]["x_out"] if l > 0 else ctx["x0"]
            if ctx["ckpt"]:
                _hip.call("vlr_layers_join")          # the previous layer's weight-gradient GEMMs still read the shared activation set
                self._layer_fwd_call(ws, l, a, x_in, ctx["embed"], Bn, S, True, False, None)
            _hip.call("vlr_decoder_layer_bwd", self.llama_cfg, self.layer_weights(ws, l), self.layer_grads(l), acc,
                      a["struct"], lws, x_in, cur, nxt, ctx["pos"], ctx["mask"], Bn, S)
            if f"l{l}.bqkv" in self.gv:              # bias of the fused q|k|v projection: column sum of this layer's d qkv (post rope-transpose)
                _hip.call("vlr_colsum", wsb["dqkv"], M, self.Nqkv, self.Nqkv, self.gv[f"l{l}.bqkv"], acc, self._colsum_ws)
            cur, nxt = nxt, cur
            if self.reducer is not None:
                _hip.call("vlr_layers_join")          # wgrad GEMMs of this layer run on the library's side stream
                self.reducer.bucket_ready(f"layer{l}")
        _hip.call("vlr_layers_join")
        self._embed_backward(ctx, cur, acc)
        self.grad_fresh = False
        if self.reducer is not None:
            self.reducer.bucket_ready("tail")

    def _embed_backward(self, ctx, cur, acc):
        """gradient of the merged embeddings `cur` [M,H] -> embed_tokens rows, projector (and image_newline) gradients"""
        ws = ctx["ws"]
        Bn, S, H = ctx["Bn"], ctx["S"], self.H
        # ---- merge + projector
        n_rows, dup = ctx["n_rows"], ctx["image_dup"]
        if not acc:
            self.gv["embed"].zero_()          # rows of tokens that do not occur keep a zero gradient
        if ctx["pack"] is None:
            dfeats = self._buf(("dfeats", n_rows), (n_rows, H))
            _hip.call("vlr_merge_bwd", cur, ctx["src"], ctx["inv"], ctx["ids"], dfeats, self.gv["embed"], Bn, ctx["T"], S, H, n_rows, dup)
        else:
            # LLaVA-Next: gradient of the packed rows, then un-pack: every projector row feeds at most one packed row (tiles the
            # un-padding dropped get zero), the image_newline row feeds one packed row per grid line -> fixed-order column sum
            pk = ctx["pack"]
            F = pk["F"]
            dpacked = self._buf(("dpacked", F), (F, H))
            _hip.call("vlr_merge_bwd", cur, ctx["src"], ctx["inv"], ctx["ids"], dpacked, self.gv["embed"], Bn, ctx["T"], S, H, F, dup)
            dext = self._buf(("dfeats_ext", n_rows), (n_rows + 1, H))
            dext.zero_()
            _hip.call("vlr_scatter_rows", dpacked, pk["idx"], dext, F, H)
            n_nl = int(pk["nl"].shape[0])
            dnl = self._buf(("dnewline", n_nl), (n_nl, H))
            _hip.call("vlr_gather_rows", dpacked, pk["nl"], dnl, n_nl, H)
            _hip.call("vlr_colsum", dnl, n_nl, H, H, self.gv["image_newline"], acc, self._colsum_ws)
            dfeats = dext[:n_rows]
        D = self.D
        _hip.call("vlr_colsum", dfeats, n_rows, H, H, self.gv["proj.b2"], acc, self._colsum_ws)
        _hip.call("vlr_gemm_bf16", 2, dfeats, ctx["proj_h"], self.gv["proj.w2"], None, None, H, H, n_rows, H, H, H, 0, 0, acc, 0)
        dh = self._buf(("proj_dh", n_rows), (n_rows, H))
        _hip.call("vlr_gemm_bf16", 1, dfeats, ws.v["proj.w2"], dh, None, None, n_rows, H, H, H, H, H, 0, 0, 0, 0)
        dz = self._buf(("proj_dz", n_rows), (n_rows, H))
        _hip.call("vlr_gelu_bwd", ctx["proj_z"], dh, dz, dz.numel())
        _hip.call("vlr_colsum", dz, n_rows, H, H, self.gv["proj.b1"], acc, self._colsum_ws)
        _hip.call("vlr_gemm_bf16", 2, dz, ctx["vit_feat"], self.gv["proj.w1"], None, None, H, D, n_rows, H, D, D, 0, 0, acc, 0)

    def _hidden_backward_lora(self, ctx, dhidden, dxa, dxb):
        """LoRA backward: data gradients through the frozen decoder + adapter gradients only; nothing below the first
        decoder layer is trainable (embedding, projector and vision tower are not target modules), so it stops there."""
        if ctx.get("lora_seed") is None:
            raise RuntimeError("backward through a pass that ran with the adapters disabled")
        ws = ctx["ws"]
        Bn, S, M, H, I = ctx["Bn"], ctx["S"], ctx["M"], self.H, self.I
        acc = int(not self.grad_fresh)
        Sp = _align(S, 64)
        r = self.lora["r"]
        self._norm_bwd(dhidden, ctx["x_last"], ws.v["norm"], ctx["rstd_f"], None, dxa, None, 0, M)
        wsb = dict(dact=self._buf(("dact", M), (M, I)), dxn=self._buf(("dxn", M), (M, H)), dattn=self._buf(("dattn", M), (M, H)),
                   dqkv=self._buf(("dqkv", M), (M, 3 * H)), dx_mid=self._buf(("dx_mid", M), (M, H)),
                   delta=self._buf(("delta", Bn, S), (Bn, self.nh, Sp), torch.float32))
        lws = _hip.LayerBwdWs(wsb["dact"].data_ptr(), wsb["dxn"].data_ptr(), wsb["dattn"].data_ptr(), wsb["dqkv"].data_ptr(),
                              wsb["dx_mid"].data_ptr(), wsb["delta"].data_ptr(), self._norm_ws.data_ptr())
        ws_v = self._buf(("lora_v", M), (M, 3 * r))
        drop = self.lora["dropout"] > 0 and self.training
        scratch = self._buf(("lora_scratch", M), (M, max(H, I))) if drop else None     # fallback path of the fused dropout-accumulate
        cur, nxt = dxa, dxb
        for l in range(self.L - 1, -1, -1):
            a = ctx["acts"][l]
            x_in = ctx["acts"][l - 1]["x_out"] if l > 0 else ctx["x0"]
            lw, lg = self._lora_structs(l, train=True, M=M, acts=a)
            if ctx["ckpt"]:
                self._layer_fwd_call(ws, l, a, x_in, ctx["embed"], Bn, S, True, True, ctx["lora_seed"])
            sh = a.get("shared", a)
            _hip.call("vlr_decoder_layer_bwd_lora", self.llama_cfg, self.layer_weights(ws, l), lw, lg, acc, a["struct"], sh["u"],
                      lws, ws_v, scratch, ctx["lora_seed"] + 8 * l, x_in, cur, nxt, ctx["pos"], ctx["mask"], Bn, S)
            cur, nxt = nxt, cur
        self.grad_fresh = False
        if self.reducer is not None:
            self.reducer.bucket_ready("lora")

    def make_reducer(self, group=None):
        """DDP gradient reducer over the trainable flat gradient buffer (full fine-tuning: one bucket per decoder layer in
        backward order; LoRA: the adapters are small, one bucket when the backward is done)."""
        from .parallel import GradReducer
        if self.lora is not None:
            self.reducer = GradReducer(self.lora_grads, {"lora": (0, self.lora_layout.numel)}, group=group)
        else:
            self.reducer = GradReducer(self.grads, self.layout.bucket_after, group=group)
        return self.reducer

    # ------------------------------------------------------------------------------------------------ optimizer
    def init_optimizer(self):
        """fp32 master copy + Adam moments for the flat parameter buffer (28 B of HBM traffic per parameter and step).
        The reference leaves the precision policy to DeepSpeed/DDP; fp32 master + fp32 moments is the documented choice."""
        self.master = (self.lora_flat if self.lora is not None else self.policy.flat).float()
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        self.opt_step = 0

    def optimizer_state(self):
        """fp32 master / m / v + step counter (what torch.optim.AdamW.state_dict() carries), for checkpoints."""
        if self.master is None:
            return None
        self.wait_optimizer()
        return dict(master=self.master, m=self.m, v=self.v, opt_step=self.opt_step)

    def load_optimizer_state(self, master, m, v, opt_step):
        if self.master is None:
            self.init_optimizer()
        for dst, src in ((self.master, master), (self.m, m), (self.v, v)):
            if dst.numel() != src.numel():
                raise ValueError(f"optimizer state has {src.numel()} elements, the trainable buffer {dst.numel()}")
            dst.copy_(src.to(self.dev))
        self.opt_step = int(opt_step)
        # the bf16 working copy is the rounding of the master weights
        (self.lora_flat if self.lora is not None else self.policy.flat).copy_(self.master)
        self._weights_version += 1

    def zero_grad(self):
        self.grad_fresh = True

    def wait_optimizer(self):
        """make the current stream wait for the last optimizer step (no host synchronisation)"""
        if self._opt_done is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._opt_done)

    def grad_norm(self) -> float:
        """total gradient norm of the last optimizer step (host value; synchronises on the optimizer stream)"""
        if self._opt_done is not None:
            self._opt_done.synchronize()
        return float(self.norm_out[0])

    def optimizer_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, grad_scale=1.0):
        """clip + AdamW, issued on the optimizer stream behind everything queued on the caller's stream so far."""
        if self.master is None:
            self.init_optimizer()
        if self.reducer is not None:
            self.reducer.wait()
        if self._opt_stream is None:
            return self._optimizer_step(lr, beta1, beta2, eps, weight_decay, max_grad_norm, grad_scale)
        main = torch.cuda.current_stream(self.dev)
        self._opt_stream.wait_stream(main)
        with torch.cuda.stream(self._opt_stream):
            out = self._optimizer_step(lr, beta1, beta2, eps, weight_decay, max_grad_norm, grad_scale)
            self._opt_done = torch.cuda.Event()
            self._opt_done.record(self._opt_stream)
        return out

    def _optimizer_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, grad_scale=1.0):
        """clip_grad_norm_(max_grad_norm) + AdamW on the flat buffers; no host synchronisation (the clip coefficient
        stays on the device).  grad_scale multiplies the raw gradients first (1/world_size after a sum all-reduce,
        1/gradient_accumulation_steps ...).  Returns the device tensor [norm, coef, sum g^2]."""
        if self.master is None:
            self.init_optimizer()
        if self.reducer is not None:
            self.reducer.wait()
        if self.lora is not None:
            # HF's decay grouping puts every lora_A / lora_B weight in the decay group.
            n = self.lora_layout.numel
            _hip.call("vlr_grad_sqnorm", self.lora_grads, n, float(max_grad_norm if max_grad_norm else 0.0),
                      float(grad_scale), 0.0, self._sq_ws, self.norm_out)
            self.opt_step += 1
            _hip.call("vlr_adamw_step", self.master, self.m, self.v, self.lora_grads, self.lora_flat, n, float(lr),
                      float(beta1), float(beta2), float(eps), float(weight_decay), self.opt_step, self.norm_out)
            self.grad_fresh = True
            return self.norm_out
        n = self.layout.n_opt
        _hip.call("vlr_grad_sqnorm", self.grads, n, float(max_grad_norm if max_grad_norm else 0.0), float(grad_scale), 0.0,
                  self._sq_ws, self.norm_out)
        self.opt_step += 1
        nd = self.layout.n_decay
        for lo, hi, wd in ((0, nd, weight_decay), (nd, n, 0.0)):
            if hi <= lo:              # (InternLM-XComposer2: no bias / nn.LayerNorm parameter is trainable - the no-decay region is empty)
                continue
            _hip.call("vlr_adamw_step", self.master[lo:hi], self.m[lo:hi], self.v[lo:hi], self.grads[lo:hi],
                      self.policy.flat[lo:hi], hi - lo, float(lr), float(beta1), float(beta2), float(eps), float(wd),
                      self.opt_step, self.norm_out)
        self.grad_fresh = True
        return self.norm_out
