"""InternLM-XComposer2 on the MI355X engine (reference /root/reference/src/vlrlhf/models/InternLMXC2/).

What is shared with LLaVA: the CLIP tower (here its LAST hidden state, 35 x 35 patches at 490 px), the mlp2x_gelu projector (frozen
together with the tower, __init__.py:252-255), the <ImageHere> expansion (= the LLaVA merge kernels), grouped-query attention.
What is not:
  * the fused `wqkv` rows are re-ordered from the checkpoint's per-K/V-head layout to q | k | v blocks at load time (engine.ParamLayout);
  * rotary positions are the index in the merged sequence (the vendored apply_rotary_pos_emb ignores position_ids);
  * every decoder linear carries a PLoRA pair that acts on the IMAGE rows only: y[img] += Plora_B(Plora_A(dropout(x[img]))) (build_mlp.py
    :158-203).  They are base-model weights - trained by a full fine-tune, frozen but ACTIVE (reference pass included) under peft LoRA,
    which the shipped script stacks on top of the same five linears.
The PLoRA term is a low-rank update of a row subset, so the decoder layer is composed HERE from the library's primitives instead of
the fused `vlr_decoder_layer_*` calls: the image rows of the batch are gathered into a compact [R, in] matrix (vlr_gather_rows /
vlr_rows_gather), run through two skinny GEMMs and added back (vlr_rows_add).  Correctness first: the projections run un-fused (the
adapter terms must reach the accumulator before RoPE / SwiGLU), ~110 launches per layer and pass.  DESIGN.md lists what a fused version
needs (a row-masked adapter segment in the GEMM K loop).
Dropout convention (the reference draws from torch's RNG): target t of layer l masks the compact [R, in] matrix with the counter-based
mask of vlr_dropout(seed + 8 l + t), t = 0 wqkv, 3 wo, 4 w1, 5 w3, 6 w2; PLoRA and LoRA use different seed bases."""
import math
import os
from typing import Dict

import torch

from . import _hip
from .engine import BF16, LlavaHipEngine, _align

PLORA_T = dict(qkv=0, o=3, g=4, u=5, d=6)
PLORA_SEED_XOR = 0x2A5A5A5A5A


class InternLMHipEngine(LlavaHipEngine):
    custom_layers = True
    vision_prefix = "vit.vision_tower."

    @property
    def supports_ckpt(self):       # the C layer passes (full fine-tune / reference / LoRA over PLoRA) can be re-run; the Python-composed peft-LoRA layer keeps its activations
        return (self.lora is None or bool(getattr(self, "lora_fused", False))) and bool(getattr(self, "fused_forward", True))      # (getattr: read by the base __init__)

    def __init__(self, cfg: dict, device="cuda", max_positions: int = 8192):
        c = dict(cfg, family="internlm_xc2")
        c.setdefault("vit_feature_layer", -1)
        c.setdefault("rope_theta", 1000000.0)
        super().__init__(c, device=device, max_positions=max_positions)
        self.plora_r = int(c.get("plora_r", 256))
        self.plora_scale = float(c.get("plora_alpha", 256)) / self.plora_r
        self.plora_p = float(c.get("plora_dropout", 0.05))
        self._plora_calls = 0
        self.plora_seed = int(c.get("seed", 0))
        self.fused_forward = os.environ.get("VLR_ILM_FUSED", "1") != "0"     # PLoRA-only passes on the C layer calls (0: the Python-composed layer everywhere)
        self.lora_fused = False                  # set by enable_lora: peft LoRA over PLoRA on vlr_decoder_layer_*_lora2
        self.last_train_plora_seed = None        # PLoRA dropout seed of the last training-mode policy pass
        if not self.fused_forward:
            self._to_bf16_stream()

    # ------------------------------------------------------------------------------------------------ embed
    def _embed_inputs(self, ws, ids, am, lab, pixel_values, image_dup, tag, image_sizes, meta):
        e = super()._embed_inputs(ws, ids, am, lab, pixel_values, image_dup, tag, image_sizes, meta)
        Bn, S = ids.shape[0], e["S"]
        cached = meta.get("ilm") if meta is not None else None
        if cached is None:
            rows = e["img_map"].reshape(-1).nonzero().reshape(-1).to(torch.int32).contiguous()       # one D2H sync per batch (row count)
            cached = dict(rows=rows, R=int(rows.numel()),
                          pos=torch.arange(S, dtype=torch.int32, device=self.dev)[None].expand(Bn, S).contiguous())
            if meta is not None:
                meta["ilm"] = cached
        e["pos"] = cached["pos"]                       # rotary position = index in the merged sequence
        e["extra"] = cached
        self._plora_calls += 1                         # one PLoRA dropout stream per forward pass (a recompute re-uses the pass's seed)
        e["plora_seed"] = ((self.plora_seed << 40) + (self._plora_calls << 16)) ^ PLORA_SEED_XOR
        return e

    def enable_lora(self, r, alpha, dropout=0.0, seed=0):
        """peft LoRA stacked on the frozen PLoRA decoder (the configuration the reference ships for this family).  Default: the C layer
        passes with TWO adapters per projection (vlr_decoder_layer_fwd_lora2 / bwd_lora2, include/vlr.h) - fused qkv + RoPE / SwiGLU /
        residual projections, one adapter segment [B_lora | B_plora] in the K loop, fp32 residual stream.  VLR_ILM_LORA_FUSED=0 composes
        the layer from bf16 primitives instead (rows_add, accumulate GEMMs; bf16 residual stream) - the cross-check of the fused path."""
        super().enable_lora(r, alpha, dropout, seed)
        self.lora_fused = bool(self.fused_forward) and os.environ.get("VLR_ILM_LORA_FUSED", "1") != "0"
        if not self.lora_fused:
            self._to_bf16_stream()

    def _bcomb(self, ws, l):
        """[B_lora | B_plora] of layer l for the four fused projections (vlr_lora_bcomb): rebuilt from the live adapter weights in front
        of every layer pass (4 copies of 26 MB at the 7B shapes, ~10 us) so an optimizer step or a loaded adapter is never stale"""
        rl, rp = self.lora["r"], self.plora_r
        ptrs = []
        for grp, pk in (("qkv", "pb_qkv"), ("o", "pb_o"), ("gu", "pb_gu"), ("down", "pb_d")):
            Bl, Bp = self.lv[f"l{l}.b_{grp}"], ws.v[f"l{l}.{pk}"]
            rows = Bl.shape[0]
            out = self._buf(("bcomb", grp), (rows, rl + rp))
            _hip.call("vlr_lora_concat_b", Bl, rl, Bp, rp, out, rows)
            ptrs.append(out.data_ptr())
        return _hip.LoraBcomb(*ptrs)

    def _to_bf16_stream(self):
        if self.resid_f32:
            self.resid_f32, self.RDT = False, BF16
            self.llama_cfg = _hip.LlamaCfg(self.H, self.I, self.nh, self.hd, self.llama_cfg.rms_eps, self.max_pos, self.cos.data_ptr(),
                                           self.sin.data_ptr(), self.nkv, 0)
            self._ws = {}

    def _plora_structs(self, ws, l, p, M=None, acts=None):
        """the PLoRA pairs of layer l as the adapter structs of the C layer passes (include/vlr.h vlr_lora_weights): ONE adapter over the
        fused wqkv, wo, w1 | w3 stacked, w2; scale = alpha / r; gradients into the flat gradient buffer (full fine-tune) or nowhere"""
        v = ws.v
        names = ("pa_qkv", "pb_qkv", "pa_o", "pb_o", "pa_gu", "pb_gu", "pa_d", "pb_d")
        w = _hip.LoraWeights(self.plora_r, self.plora_scale, float(p), *(v[f"l{l}.{n}"].data_ptr() for n in names), 1,
                             self._mask_bits(l, M, float(p), acts, "plora_bits") if M else None)
        g = _hip.LoraGrads(*(self.gv[f"l{l}.{n}"].data_ptr() for n in names)) if self.gv is not None else None
        return w, g


    def _embed_backward(self, ctx, cur, acc):
        """the projector is frozen with the tower: only tok_embeddings receives a gradient in front of the decoder"""
        if not acc:
            self.gv["embed"].zero_()
        _hip.call("vlr_merge_bwd", cur, ctx["src"], ctx["inv"], ctx["ids"], None, self.gv["embed"], ctx["Bn"], ctx["T"], ctx["S"], self.H,
                  ctx["n_rows"], ctx["image_dup"])

    # ------------------------------------------------------------------------------------------------ adapters
    def _pab(self, views, l, key):
        """(Plora_A [r][in], Plora_B [out][r]) of target `key`; gate (w1) / up (w3) are the two halves of the stacked `gu` pair"""
        r, I = self.plora_r, self.I
        if key == "g":
            return views[f"l{l}.pa_gu"][:r], views[f"l{l}.pb_gu"][:I]
        if key == "u":
            return views[f"l{l}.pa_gu"][r:], views[f"l{l}.pb_gu"][I:]
        return views[f"l{l}.pa_{key}"], views[f"l{l}.pb_{key}"]

    def _gemm(self, layout, A, B, C, M, N, K, lda, ldb, ldc, residual=None, ldr=0, accumulate=0, alpha=1.0):
        _hip.call("vlr_gemm_bf16_scaled", layout, A, B, C, None, residual, M, N, K, lda, ldb, ldc, ldr, 0, accumulate, 0, float(alpha))

    def _plora_fwd(self, ws, l, key, x_in, n_in, y, ldy, n_out, ex, train, seed, keep, tag):
        """y[img rows] += scale * (drop(x_in[img rows]) A^T) B^T;  y is a [M, n_out] column block with row stride ldy"""
        R, rows, r = ex["R"], ex["rows"], self.plora_r
        if R == 0:
            return None
        A, B = self._pab(ws.v, l, key)
        xs = torch.empty(R, n_in, dtype=BF16, device=self.dev) if keep else self._buf((tag, "pl_xs", R, n_in), (R, n_in))
        _hip.call("vlr_gather_rows", x_in, rows, xs, R, n_in)
        if train and self.plora_p > 0:
            _hip.call("vlr_dropout", xs, xs, R * n_in, self.plora_p, seed + PLORA_T[key], 1.0, 0)
        up = torch.empty(R, r, dtype=BF16, device=self.dev) if keep else self._buf((tag, "pl_up", R), (R, r))
        self._gemm(0, xs, A, up, R, r, n_in, n_in, n_in, r, alpha=self.plora_scale)
        yp = self._buf((tag, "pl_y", R, n_out), (R, n_out))
        self._gemm(0, up, B, yp, R, n_out, r, r, r, n_out)
        _hip.call("vlr_rows_add", yp, rows, y, ldy, R, n_out)
        return (xs, up) if keep else None

    def _plora_u(self, ws, l, key, x_in, n_in, u, ldu, ex, train, seed, keep, tag):
        """u [M][ldu] (column block of r) = scale * drop(x_in[img rows]) A^T on the image rows, ZERO on the text rows: the adapter-segment
        GEMMs (vlr_gemm_*_lora) then add u B^T inside the K loop of the base projection - PLoRA without a separate pass over y"""
        R, rows, r = ex["R"], ex["rows"], self.plora_r
        M = u.shape[0]
        if ldu == r:
            u.zero_()
        else:
            u[:, :r].zero_()
        if R == 0:
            return None
        A, _ = self._pab(ws.v, l, key)
        xs = torch.empty(R, n_in, dtype=BF16, device=self.dev) if keep else self._buf((tag, "pl_xs", R, n_in), (R, n_in))
        _hip.call("vlr_gather_rows", x_in, rows, xs, R, n_in)
        if train and self.plora_p > 0:
            _hip.call("vlr_dropout", xs, xs, R * n_in, self.plora_p, seed + PLORA_T[key], 1.0, 0)
        up = torch.empty(R, r, dtype=BF16, device=self.dev) if keep else self._buf((tag, "pl_up", R), (R, r))
        self._gemm(0, xs, A, up, R, r, n_in, n_in, n_in, r, alpha=self.plora_scale)
        _hip.call("vlr_rows_add", up, rows, u, ldu, R, r)
        return (xs, up) if keep else None

    def _layer_forward_fused(self, ws, l, a, x, e, Bn, S, save, train, pseed, keep_p):
        """policy (full fine-tune) / reference pass: only PLoRA sits on the linears, and it rides the fused projections"""
        c, H, I, N, M, r = self.llama_cfg, self.H, self.I, self.Nqkv, Bn * S, self.plora_r
        ex, pos, mask, tag = e["extra"], e["pos"], e["mask"], e["tag"]
        kept = {}
        ub = self._buf((tag, "pl_ufull", M), (M, 2 * r))
        u1 = ub[:, :r]
        _hip.call("vlr_rmsnorm_fwd", x, ws.v[f"l{l}.ln1"], a["xn1"], a["rstd1"], M, H, c.rms_eps)
        kept["p_qkv"] = self._plora_u(ws, l, "qkv", a["xn1"], H, u1, 2 * r, ex, train, pseed, keep_p, tag)
        _hip.call("vlr_gemm_qkv_rope_lora", a["xn1"], ws.v[f"l{l}.wqkv"], None, a["qkv"], pos, self.cos, self.sin, M, N, self.Nq + self.Nkv, H, H,
                  self.hd, self.max_pos, u1, 2 * r, ws.v[f"l{l}.pb_qkv"], r, N, 0)
        _hip.call("vlr_attn_fwd_gqa", a["qkv"], a["qkv"][:, self.Nq:], a["qkv"][:, self.Nq + self.Nkv:], N, a["attn"], self.Nq, a["lse"], mask,
                  Bn, S, self.nh, self.nkv, self.hd, 1, 1.0 / math.sqrt(self.hd))
        kept["p_o"] = self._plora_u(ws, l, "o", a["attn"], self.Nq, u1, 2 * r, ex, train, pseed, keep_p, tag)
        _hip.call("vlr_gemm_lora", a["attn"], self.Nq, ws.v[f"l{l}.wo"], a["x_mid"], H, x, H, M, H, self.Nq, u1, 2 * r, ws.v[f"l{l}.pb_o"], r)
        _hip.call("vlr_rmsnorm_fwd", a["x_mid"], ws.v[f"l{l}.ln2"], a["xn2"], a["rstd2"], M, H, c.rms_eps)
        kept["p_g"] = self._plora_u(ws, l, "g", a["xn2"], H, ub, 2 * r, ex, train, pseed, keep_p, tag)
        kept["p_u"] = self._plora_u(ws, l, "u", a["xn2"], H, ub[:, r:], 2 * r, ex, train, pseed, keep_p, tag)
        _hip.call("vlr_gemm_swiglu_lora", a["xn2"], ws.v[f"l{l}.wgu"], a["gu"], a["act"], M, I, H, H, ub, 2 * r, ws.v[f"l{l}.pb_gu"], r)
        kept["p_d"] = self._plora_u(ws, l, "d", a["act"], I, u1, 2 * r, ex, train, pseed, keep_p, tag)
        _hip.call("vlr_gemm_lora", a["act"], I, ws.v[f"l{l}.wdown"], a["x_out"], H, a["x_mid"], H, M, H, I, u1, 2 * r, ws.v[f"l{l}.pb_d"], r)
        return kept

    def _plora_bwd(self, ws, l, key, kept, dy, lddy, n_out, dx, n_in, ex, train, seed, acc, trainable):
        R, rows, r = ex["R"], ex["rows"], self.plora_r
        if R == 0:
            return
        A, B = self._pab(ws.v, l, key)
        dyr = self._buf(("pl_dy", R, n_out), (R, n_out))
        _hip.call("vlr_rows_gather", dy, lddy, rows, dyr, R, n_out)
        v = self._buf(("pl_v", R), (R, r))
        self._gemm(1, dyr, B, v, R, r, n_out, n_out, r, r)                                   # v = dy B
        if trainable:
            xs, up = kept
            gA, gB = self._pab(self.gv, l, key)
            self._gemm(2, dyr, up, gB, n_out, r, R, n_out, r, r, accumulate=acc)            # dB = dy^T (s u)
            self._gemm(2, v, xs, gA, r, n_in, R, r, n_in, n_in, accumulate=acc, alpha=self.plora_scale)   # dA = s v^T drop(x)
        dxr = self._buf(("pl_dx", R, n_in), (R, n_in))
        self._gemm(1, v, A, dxr, R, n_in, r, r, n_in, n_in, alpha=self.plora_scale)
        if train and self.plora_p > 0:
            _hip.call("vlr_dropout", dxr, dxr, R * n_in, self.plora_p, seed + PLORA_T[key], 1.0, 0)
        _hip.call("vlr_rows_add", dxr, rows, dx, n_in, R, n_in)

    def _lora_fwd(self, l, grp, t, x_in, n_in, y, ldy, n_out, row0, seed, keep, M, tag):
        """peft adapter of sub-target t of group grp on ALL rows: y += (s drop(x) A_t^T) B_t^T"""
        lo = self.lora
        r = lo["r"]
        A = self.lv[f"l{l}.a_{grp}"][t * r:(t + 1) * r]
        B = self.lv[f"l{l}.b_{grp}"][row0:row0 + n_out]
        p = lo["dropout"] if self.training else 0.0
        xd = x_in
        if p > 0:
            xd = torch.empty(M, n_in, dtype=BF16, device=self.dev) if keep else self._buf((tag, "lo_xd", M, n_in), (M, n_in))
            _hip.call("vlr_dropout", x_in, xd, M * n_in, p, seed, 1.0, 0)
        u = torch.empty(M, r, dtype=BF16, device=self.dev) if keep else self._buf((tag, "lo_u", M), (M, r))
        self._gemm(0, xd, A, u, M, r, n_in, n_in, n_in, r, alpha=lo["scale"])
        self._gemm(0, u, B, y, M, n_out, r, r, r, ldy, accumulate=1)
        return (xd, u) if keep else None

    def _lora_bwd(self, l, grp, t, kept, dy, lddy, n_out, row0, dx, n_in, seed, acc, M):
        lo = self.lora
        r = lo["r"]
        A = self.lv[f"l{l}.a_{grp}"][t * r:(t + 1) * r]
        B = self.lv[f"l{l}.b_{grp}"][row0:row0 + n_out]
        xd, u = kept
        p = lo["dropout"] if self.training else 0.0
        self._gemm(2, dy, u, self.lgv[f"l{l}.b_{grp}"][row0:row0 + n_out], n_out, r, M, lddy, r, r, accumulate=acc)           # dB = dy^T (s u)
        v = self._buf(("lo_v", M), (M, r))
        self._gemm(1, dy, B, v, M, r, n_out, lddy, r, r)
        self._gemm(2, v, xd, self.lgv[f"l{l}.a_{grp}"][t * r:(t + 1) * r], r, n_in, M, r, n_in, n_in, accumulate=acc, alpha=lo["scale"])
        if p > 0:
            scratch = self._buf(("lo_dx", M, n_in), (M, n_in))
            _hip.call("vlr_gemm_dropout_acc", v, r, A, dx, scratch, M, n_in, r, p, seed, lo["scale"])
        else:
            self._gemm(1, v, A, dx, M, n_in, r, r, n_in, n_in, accumulate=1, alpha=lo["scale"])

    # ------------------------------------------------------------------------------------------------ layer forward
    def _targets(self):
        H, I, N = self.H, self.I, self.Nqkv
        #        key   lora group, sub-target, lora_B row0, in, out
        return dict(qkv=("qkv", 0, 0, H, N), o=("o", 0, 0, self.Nq, H), g=("gu", 0, 0, H, I), u=("gu", 1, I, H, I), d=("down", 0, 0, I, H))

    def _layer_forward(self, ws, l, a, x, e, Bn, S, save, use_lora, lora_seed):
        c, H, I, N, M = self.llama_cfg, self.H, self.I, self.Nqkv, Bn * S
        ex, pos, mask = e["extra"], e["pos"], e["mask"]
        train = self.training and ws is self.policy and bool(e.get("grad_pass", save))     # (a checkpointed forward keeps nothing but is the same pass)
        pseed = e["plora_seed"]
        keep_p = save and self.lora is None          # PLoRA weights are trainable only in a full fine-tune
        if train and l == 0:
            self.last_train_plora_seed = pseed       # (tests: the oracle regenerates the pass's masks from it)
        if not use_lora and self.fused_forward:
            # only PLoRA sits on the linears (reference pass; policy pass of a full fine-tune): the C layer pass with the PLoRA pairs as
            # its adapters and the image rows as the row mask - fused qkv + RoPE / SwiGLU / residual projections, fp32 residual stream
            M = Bn * S
            sh = a.get("shared", a)
            if "u" not in sh or sh["u"].shape[1] != 7 * self.plora_r:
                sh["u"] = torch.empty(M, 7 * self.plora_r, dtype=BF16, device=self.dev)
            pw, _ = self._plora_structs(ws, l, self.plora_p if train else 0.0, M, a)
            _hip.call("vlr_decoder_layer_fwd_lora_ex", self.llama_cfg, self.layer_weights(ws, l), pw, a["struct"], sh["u"], None,
                      pseed + 8 * l, e["img_map"], x, e["pos"], e["mask"], Bn, S)
            if save:
                a["pseed"], a["train"] = pseed + 8 * l, train
            return
        if use_lora and self.lora_fused:
            R = self.lora["r"] + self.plora_r
            sh = a.get("shared", a)
            if "u2" not in sh or sh["u2"].shape[1] != 7 * R:
                sh["u2"] = torch.empty(M, 7 * R, dtype=BF16, device=self.dev)
            lw, _ = self._lora_structs(l, train=True, M=M, acts=a)
            pw, _ = self._plora_structs(ws, l, self.plora_p if train else 0.0, M, a)
            bc = self._bcomb(ws, l)
            _hip.call("vlr_decoder_layer_fwd_lora2", self.llama_cfg, self.layer_weights(ws, l), lw, pw, bc, a["struct"], sh["u2"],
                      lora_seed + 8 * l, pseed + 8 * l, e["img_map"], x, e["pos"], e["mask"], Bn, S)
            if save:
                a["pseed"], a["train"] = pseed + 8 * l, train
            return
        kept: Dict[str, object] = {}
        tg = self._targets()

        def adapters(key, x_in, y, ldy):
            grp, t, row0, n_in, n_out = tg[key]
            kept["p_" + key] = self._plora_fwd(ws, l, key, x_in, n_in, y, ldy, n_out, ex, train, pseed + 8 * l, keep_p, e["tag"])
            if use_lora:
                kept["l_" + key] = self._lora_fwd(l, grp, t, x_in, n_in, y, ldy, n_out, row0, lora_seed + 8 * l + PLORA_T[key], save, M, e["tag"])

        _hip.call("vlr_rmsnorm_fwd", x, ws.v[f"l{l}.ln1"], a["xn1"], a["rstd1"], M, H, c.rms_eps)
        self._gemm(0, a["xn1"], ws.v[f"l{l}.wqkv"], a["qkv"], M, N, H, H, H, N)
        adapters("qkv", a["xn1"], a["qkv"], N)
        _hip.call("vlr_rope_heads", a["qkv"], pos, self.cos, self.sin, M, self.nh + self.nkv, self.hd, N, self.max_pos, 0)
        _hip.call("vlr_attn_fwd_gqa", a["qkv"], a["qkv"][:, self.Nq:], a["qkv"][:, self.Nq + self.Nkv:], N, a["attn"], self.Nq, a["lse"], mask,
                  Bn, S, self.nh, self.nkv, self.hd, 1, 1.0 / math.sqrt(self.hd))
        self._gemm(0, a["attn"], ws.v[f"l{l}.wo"], a["x_mid"], M, H, self.Nq, self.Nq, self.Nq, H, residual=x, ldr=H)
        adapters("o", a["attn"], a["x_mid"], H)
        _hip.call("vlr_rmsnorm_fwd", a["x_mid"], ws.v[f"l{l}.ln2"], a["xn2"], a["rstd2"], M, H, c.rms_eps)
        self._gemm(0, a["xn2"], ws.v[f"l{l}.wgu"], a["gu"], M, 2 * I, H, H, H, 2 * I)
        adapters("g", a["xn2"], a["gu"], 2 * I)
        adapters("u", a["xn2"], a["gu"][:, I:], 2 * I)
        _hip.call("vlr_swiglu_fwd", a["gu"], a["act"], M, I)
        self._gemm(0, a["act"], ws.v[f"l{l}.wdown"], a["x_out"], M, H, I, I, I, H, residual=a["x_mid"], ldr=H)
        adapters("d", a["act"], a["x_out"], H)
        if save:
            a["kept"] = kept
            a["pseed"] = pseed + 8 * l
            a["train"] = train

    # ------------------------------------------------------------------------------------------------ backward
    def _hidden_backward_full(self, ctx, dxa, dxb, acc):
        """full fine-tune: vlr_decoder_layer_bwd_lora_ex per layer - base AND PLoRA gradients, the adapters restricted to the image rows;
        with gradient checkpointing each layer's forward is re-run (same dropout seed) right before its backward"""
        ws = ctx["ws"]
        Bn, S, M, H, I, N = ctx["Bn"], ctx["S"], ctx["M"], self.H, self.I, self.Nqkv
        Sp = _align(S, 64)
        r = self.plora_r
        wsb = dict(dact=self._buf(("dact", M), (M, I)), dxn=self._buf(("dxn", M), (M, H)), dattn=self._buf(("dattn", M), (M, self.Nq)),
                   dqkv=self._buf(("dqkv", M), (M, N)), dx_mid=self._buf(("dx_mid", M), (M, H)),
                   delta=self._buf(("delta", Bn, S), (Bn, self.nh, Sp), torch.float32))
        lws = _hip.LayerBwdWs(wsb["dact"].data_ptr(), wsb["dxn"].data_ptr(), wsb["dattn"].data_ptr(), wsb["dqkv"].data_ptr(),
                              wsb["dx_mid"].data_ptr(), wsb["delta"].data_ptr(), self._norm_ws.data_ptr())
        ws_v = self._buf(("plora_v", M), (M, 3 * r))
        scratch = self._buf(("plora_scratch", M), (M, max(H, I)))
        e = ctx["embed"]
        cur, nxt = dxa, dxb
        for l in range(self.L - 1, -1, -1):
            a = ctx["acts"][l]
            x_in = ctx["acts"][l - 1]["x_out"] if l > 0 else ctx["x0"]
            if ctx["ckpt"]:
                self._layer_forward(ws, l, a, x_in, e, Bn, S, True, False, None)
            train = a["train"]
            pw, pg = self._plora_structs(ws, l, self.plora_p if train else 0.0, M, a)
            sh = a.get("shared", a)
            _hip.call("vlr_decoder_layer_bwd_lora_ex", self.llama_cfg, self.layer_weights(ws, l), self.layer_grads(l), pw, pg, acc, a["struct"],
                      sh["u"], lws, ws_v, scratch, a["pseed"], e["img_map"], x_in, cur, nxt, ctx["pos"], ctx["mask"], Bn, S)
            cur, nxt = nxt, cur
            if self.reducer is not None:
                self.reducer.bucket_ready(f"layer{l}")
        self._embed_backward(ctx, cur, acc)
        self.grad_fresh = False
        if self.reducer is not None:
            self.reducer.bucket_ready("tail")

    def _hidden_backward_lora2(self, ctx, dxa, dxb, acc):
        """peft LoRA over the frozen PLoRA decoder: vlr_decoder_layer_bwd_lora2 per layer - gradients of the LoRA pairs only, dx through the
        base projections and BOTH adapters (PLoRA on the image rows, its own dropout stream)"""
        ws = ctx["ws"]
        Bn, S, M, H, I, N = ctx["Bn"], ctx["S"], ctx["M"], self.H, self.I, self.Nqkv
        Sp = _align(S, 64)
        R = self.lora["r"] + self.plora_r
        wsb = dict(dact=self._buf(("dact", M), (M, I)), dxn=self._buf(("dxn", M), (M, H)), dattn=self._buf(("dattn", M), (M, self.Nq)),
                   dqkv=self._buf(("dqkv", M), (M, N)), dx_mid=self._buf(("dx_mid", M), (M, H)),
                   delta=self._buf(("delta", Bn, S), (Bn, self.nh, Sp), torch.float32))
        lws = _hip.LayerBwdWs(wsb["dact"].data_ptr(), wsb["dxn"].data_ptr(), wsb["dattn"].data_ptr(), wsb["dqkv"].data_ptr(),
                              wsb["dx_mid"].data_ptr(), wsb["delta"].data_ptr(), self._norm_ws.data_ptr())
        ws_v = self._buf(("lora2_v", M), (M, 3 * R))
        e = ctx["embed"]
        cur, nxt = dxa, dxb
        for l in range(self.L - 1, -1, -1):
            a = ctx["acts"][l]
            x_in = ctx["acts"][l - 1]["x_out"] if l > 0 else ctx["x0"]
            if ctx["ckpt"]:
                self._layer_forward(ws, l, a, x_in, e, Bn, S, True, True, ctx["lora_seed"])
            lw, lg = self._lora_structs(l, train=True, M=M, acts=a)
            pw, _ = self._plora_structs(ws, l, self.plora_p if a["train"] else 0.0, M, a)
            bc = self._bcomb(ws, l)
            sh = a.get("shared", a)
            _hip.call("vlr_decoder_layer_bwd_lora2", self.llama_cfg, self.layer_weights(ws, l), lw, lg, pw, bc, acc, a["struct"], sh["u2"], lws,
                      ws_v, ctx["lora_seed"] + 8 * l, a["pseed"], e["img_map"], x_in, cur, nxt, ctx["pos"], ctx["mask"], Bn, S)
            cur, nxt = nxt, cur
        self.grad_fresh = False
        if self.reducer is not None:
            self.reducer.bucket_ready("lora")

    def _hidden_backward_custom(self, ctx, dhidden, dxa, dxb):
        ws = ctx["ws"]
        Bn, S, M, H, I, N = ctx["Bn"], ctx["S"], ctx["M"], self.H, self.I, self.Nqkv
        full = self.lora is None
        acc = int(not self.grad_fresh)
        Sp = _align(S, 64)
        ex = ctx["extra"]
        self._norm_bwd(dhidden, ctx["x_last"], ws.v["norm"], ctx["rstd_f"], None, dxa, self.gv["norm"] if full else None, acc if full else 0, M)
        if full and self.fused_forward:
            return self._hidden_backward_full(ctx, dxa, dxb, acc)
        if not full and self.lora_fused:
            return self._hidden_backward_lora2(ctx, dxa, dxb, acc)
        dact, dxn = self._buf(("dact", M), (M, I)), self._buf(("dxn", M), (M, H))
        dattn, dqkv = self._buf(("dattn", M), (M, self.Nq)), self._buf(("dqkv", M), (M, N))
        dx_mid = self._buf(("dx_mid", M), (M, H))
        delta = self._buf(("delta", Bn, S), (Bn, self.nh, Sp), torch.float32)
        tg = self._targets()
        cur, nxt = dxa, dxb
        for l in range(self.L - 1, -1, -1):
            a = ctx["acts"][l]
            x_in = ctx["acts"][l - 1]["x_out"] if l > 0 else ctx["x0"]
            kept, pseed, train = a["kept"], a["pseed"], a["train"]
            g = (lambda k: self.gv[f"l{l}.{k}"]) if full else (lambda k: None)

            def adapters_bwd(key, dy, lddy, dx):
                grp, t, row0, n_in, n_out = tg[key]
                self._plora_bwd(ws, l, key, kept["p_" + key], dy, lddy, n_out, dx, n_in, ex, train, pseed, acc, full)
                if not full:
                    self._lora_bwd(l, grp, t, kept["l_" + key], dy, lddy, n_out, row0, dx, n_in, ctx["lora_seed"] + 8 * l + PLORA_T[key], acc, M)

            # ---- MLP
            if full:
                self._gemm(2, cur, a["act"], g("wdown"), H, I, M, H, I, I, accumulate=acc)
            self._gemm(1, cur, ws.v[f"l{l}.wdown"], dact, M, I, H, H, I, I)
            adapters_bwd("d", cur, H, dact)
            _hip.call("vlr_swiglu_bwd", a["gu"], dact, M, I)                  # gu now holds [d gate | d up]
            if full:
                self._gemm(2, a["gu"], a["xn2"], g("wgu"), 2 * I, H, M, 2 * I, H, H, accumulate=acc)
            self._gemm(1, a["gu"], ws.v[f"l{l}.wgu"], dxn, M, H, 2 * I, 2 * I, H, H)
            adapters_bwd("g", a["gu"], 2 * I, dxn)
            adapters_bwd("u", a["gu"][:, I:], 2 * I, dxn)
            _hip.call("vlr_rmsnorm_bwd", dxn, a["x_mid"], ws.v[f"l{l}.ln2"], a["rstd2"], cur, dx_mid, g("ln2"), acc if full else 0, self._norm_ws, M, H)
            # ---- attention
            if full:
                self._gemm(2, dx_mid, a["attn"], g("wo"), H, self.Nq, M, H, self.Nq, self.Nq, accumulate=acc)
            self._gemm(1, dx_mid, ws.v[f"l{l}.wo"], dattn, M, self.Nq, H, H, self.Nq, self.Nq)
            adapters_bwd("o", dx_mid, H, dattn)
            _hip.call("vlr_attn_bwd_gqa", a["qkv"], a["qkv"][:, self.Nq:], a["qkv"][:, self.Nq + self.Nkv:], N, a["attn"], dattn, self.Nq, a["lse"],
                      delta, ctx["mask"], dqkv, dqkv[:, self.Nq:], dqkv[:, self.Nq + self.Nkv:], N, Bn, S, self.nh, self.nkv, self.hd, 1,
                      1.0 / math.sqrt(self.hd))
            _hip.call("vlr_rope_heads", dqkv, ctx["pos"], self.cos, self.sin, M, self.nh + self.nkv, self.hd, N, self.max_pos, 1)
            if full:
                self._gemm(2, dqkv, a["xn1"], g("wqkv"), N, H, M, N, H, H, accumulate=acc)
            self._gemm(1, dqkv, ws.v[f"l{l}.wqkv"], dxn, M, H, N, N, H, H)
            adapters_bwd("qkv", dqkv, N, dxn)
            _hip.call("vlr_rmsnorm_bwd", dxn, x_in, ws.v[f"l{l}.ln1"], a["rstd1"], dx_mid, nxt, g("ln1"), acc if full else 0, self._norm_ws, M, H)
            cur, nxt = nxt, cur
            a["kept"] = None
            if full and self.reducer is not None:
                self.reducer.bucket_ready(f"layer{l}")
        if full:
            self._embed_backward(ctx, cur, acc)
        self.grad_fresh = False
        if self.reducer is not None:
            self.reducer.bucket_ready("tail" if full else "lora")
