"""Qwen-VL on the MI355X engine (reference /root/reference/src/vlrlhf/models/QwenVL/: modeling_qwen.py, visual.py).

The decoder is the LLaMA-shaped one of vlrlhf.engine with a biased fused q|k|v projection (`c_attn`), rotary positions =
index in the sequence, and the MLP `c_proj(w1(x) * silu(w2(x)))` stored as gate = w2 | up = w1.  What differs is everything in
front of it:
  * vision tower (visual.py:393-415): 14x14 patch embedding without bias or CLS token, a 16x16 position table bicubically
    resized to the patch grid, pre-LN transformer blocks with exact GELU and head_dim 104 - run on the 128-wide attention kernel by
    laying q|k|v out with 128 features per head (zero weight rows / columns for the padding, softmax scale 1/sqrt(104));
  * resampler (visual.py:99-155): ONE cross-attention from 256 learned queries to the 1024 patch tokens.  It runs on the
    self-attention kernel over a combined sequence [queries | keys] with a key mask that hides the query rows; the constant query
    projection and the position terms of the keys are folded into tables at load time;
  * no token expansion: the 256 slots between <img> and </img> already sit in the ids (modeling_qwen.py:616-625) and are
    overwritten with the resampler output -> the merge is a row gather with S = T.
The ViT trunk, ln_post and proj are frozen.  The resampler is trainable in a full fine-tune (QwenVLForRL.freeze_vision_tower re-enables
`attn_pool`, reference models/QwenVL/__init__.py:33-37; under the shipped LoRA script peft freezes it again): its weights live in the
engine's trainable buffer (`ap.*`) and `resampler_bwd` back-propagates through the same combined-sequence construction.
"""
import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import _hip
from .engine import BF16, LlavaHipEngine, _align

SRC_ZERO = -(2 ** 30)


def get_abs_pos(abs_pos: torch.Tensor, tgt_n: int) -> torch.Tensor:
    """visual.py:24-45: [L, C] table of a sqrt(L)^2 grid -> bicubic resize to sqrt(tgt_n)^2 positions (host, fp32, once per load)"""
    src, tgt = int(math.sqrt(abs_pos.shape[0])), int(math.sqrt(tgt_n))
    if src == tgt:
        return abs_pos.float()
    t = F.interpolate(abs_pos.float().reshape(1, src, src, -1).permute(0, 3, 1, 2), size=(tgt, tgt), mode="bicubic", align_corners=False)
    return t.permute(0, 2, 3, 1).flatten(0, 2)


class QwenVisionWeights:
    """Frozen Qwen-VL vision tower in the layouts the kernels want (bf16 on the device)."""

    def __init__(self, vcfg: dict, sd: Dict[str, torch.Tensor], device, prefix="transformer.visual."):
        W, nh, L, P = vcfg["width"], vcfg["heads"], vcfg["layers"], vcfg["patch_size"]
        hd, E = W // nh, vcfg["output_dim"]
        self.T = (vcfg["image_size"] // P) ** 2
        self.hd, self.hdp = hd, 128
        if hd > 128 or W % 8 or E % 128:
            raise ValueError(f"Qwen-VL vision tower: width {W} / heads {nh} / output_dim {E} do not fit the 128-wide attention kernel")
        dv = lambda t: t.to(BF16).to(device).contiguous()  # noqa: E731
        g = lambda k: sd[prefix + k].float()                # noqa: E731
        src_dev = sd[prefix + "conv1.weight"].device         # packing runs where the checkpoint tensors live (host or device)
        self.Kp = _align(3 * P * P, 8)
        w = torch.zeros(W, self.Kp, device=src_dev)
        w[:, : 3 * P * P] = g("conv1.weight").reshape(W, -1)
        self.patch_w = dv(w)
        self.pos = dv(get_abs_pos(g("positional_embedding"), self.T))                       # [T, W]
        self.pre_w, self.pre_b = dv(g("ln_pre.weight")), dv(g("ln_pre.bias"))
        # q|k|v rows: the reference stores them per head as [q_h | k_h | v_h] (visual.py:203-211); here q | k | v blocks of
        # heads x 128 rows, rows d >= head_dim of every head zero
        A = nh * self.hdp
        ar = lambda n_: torch.arange(n_, device=src_dev)     # noqa: E731
        idx = ar(nh)[:, None] * 3 * hd + ar(hd)[None, :]                                   # row of (head, d) of the q block in the reference
        self.layers, self._keep = [], []
        for l in range(L):
            p = f"transformer.resblocks.{l}."
            wi, bi = g(p + "attn.in_proj.weight"), g(p + "attn.in_proj.bias")
            wqkv, bqkv = torch.zeros(3 * A, W, device=src_dev), torch.zeros(3 * A, device=src_dev)
            for blk in range(3):
                rows = (idx + blk * hd).reshape(-1)
                dst = (blk * A + ar(nh)[:, None] * self.hdp + ar(hd)[None, :]).reshape(-1)
                wqkv[dst], bqkv[dst] = wi[rows], bi[rows]
            wo = torch.zeros(W, A, device=src_dev)
            cols = (ar(nh)[:, None] * self.hdp + ar(hd)[None, :]).reshape(-1)
            wo[:, cols] = g(p + "attn.out_proj.weight")
            t = dict(ln1_w=dv(g(p + "ln_1.weight")), ln1_b=dv(g(p + "ln_1.bias")), wqkv=dv(wqkv), bqkv=dv(bqkv), wo=dv(wo),
                     bo=dv(g(p + "attn.out_proj.bias")), ln2_w=dv(g(p + "ln_2.weight")), ln2_b=dv(g(p + "ln_2.bias")),
                     w1=dv(g(p + "mlp.c_fc.weight")), b1=dv(g(p + "mlp.c_fc.bias")), w2=dv(g(p + "mlp.c_proj.weight")),
                     b2=dv(g(p + "mlp.c_proj.bias")))
            self._keep.append(t)
            self.layers.append(_hip.VitLayerWeights(*(t[k].data_ptr() for k in (
                "ln1_w", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2"))))
        self.mlp = int(self._keep[0]["w1"].shape[0]) if L else int(W * vcfg["mlp_ratio"])
        # ---- resampler: its weights are TRAINABLE in a full fine-tune and live in the engine's WeightSet (`ap.*`); frozen here are only
        # the two position tables (2-D sin-cos for the queries, bicubically resized to the patch grid for the keys), ln_post and proj
        a = "attn_pool."
        self.E, self.nq, self.heads = E, int(sd[prefix + a + "query"].shape[0]), E // 128
        if (prefix + a + "kv_proj.weight") not in sd:
            raise ValueError("Qwen-VL vision tower without attn_pool.kv_proj (width == output_dim) is not supported")
        pos_q = g(a + "pos_embed")
        self.pos_q, self.pos_k = dv(pos_q), dv(get_abs_pos(pos_q, self.T))                 # [nq, E], [T, E]
        self.post_w, self.post_b = dv(g("ln_post.weight")), dv(g("ln_post.bias"))
        self.proj_t = dv(g("proj").t())                                                     # x @ proj  ==  NT GEMM with proj^T


class QwenVLHipEngine(LlavaHipEngine):
    vision_prefix = "transformer.visual."

    def __init__(self, cfg: dict, device="cuda", max_positions: int = 4096):
        c = dict(cfg, family="qwen_vl")
        c.setdefault("rms_eps", 1e-6)
        super().__init__(c, device=device, max_positions=max_positions)

    # ------------------------------------------------------------------------------------------------ vision tower
    def _init_vision_cfg(self):
        v = self.cfg["visual"]
        W, nh = v["width"], v["heads"]
        self.vit_cfg = _hip.VitCfg(W, int(W * v["mlp_ratio"]), nh, W // nh, 1e-6, 2, 128, 1.0 / math.sqrt(W // nh))
        self.nq = int(v.get("n_queries", 256))

    def _load_vision(self, sd):
        vw = QwenVisionWeights(self.cfg["visual"], sd, self.dev)
        self.vit_cfg.mlp = vw.mlp
        return vw

    def vit_trunk(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[n,3,s,s] -> output of the 48 frozen ViT blocks [n * T, width] bf16 (visual.py:393-409), cached per pixel tensor so that the
        reference pass and the policy pass share one evaluation"""
        key = (pixel_values.data_ptr(), tuple(pixel_values.shape), pixel_values._version)
        if self._vit_cache is not None and self._vit_cache[0] == key:
            return self._vit_cache[1]
        vw, v = self.vision, self.cfg["visual"]
        n, T, W = pixel_values.shape[0], vw.T, v["width"]
        if pixel_values.shape[-1] != v["image_size"]:
            raise ValueError(f"Qwen-VL vision tower expects {v['image_size']} x {v['image_size']} images, got {tuple(pixel_values.shape)}")
        A = v["heads"] * vw.hdp
        M = n * T
        pv = pixel_values.to(device=self.dev, dtype=torch.float32).contiguous()
        patches = self._buf(("qv_patches", n), (M, vw.Kp))
        _hip.call("vlr_im2col", pv, patches, n, v["image_size"], v["patch_size"], vw.Kp)
        pe = self._buf(("qv_pe", n), (M, W))
        for i in range(n):           # patch embedding + position table (the residual operand of the GEMM epilogue)
            _hip.call("vlr_gemm_bf16", 0, patches[i * T:(i + 1) * T], vw.patch_w, pe[i * T:(i + 1) * T], None, vw.pos,
                      T, W, vw.Kp, vw.Kp, vw.Kp, W, W, 0, 0, 0)
        x = torch.empty(M, W, dtype=BF16, device=self.dev)
        _hip.call("vlr_layernorm_fwd", pe, vw.pre_w, vw.pre_b, x, M, W, 1e-6)
        wsb = dict(xn=self._buf(("qv_xn", n), (M, W)), qkv=self._buf(("qv_qkv", n), (M, 3 * A)),
                   attn=self._buf(("qv_attn", n), (M, A)), h=self._buf(("qv_h", n), (M, vw.mlp)))
        ws = _hip.VitWs(*(wsb[k].data_ptr() for k in ("xn", "qkv", "attn", "h")))
        for lw in vw.layers:
            _hip.call("vlr_vit_layer_fwd", self.vit_cfg, lw, ws, x, n, T)
        self._vit_cache = (key, x, pixel_values)
        return x

    def _g(self, layout, A, B, C, M, N, K, lda, ldb, ldc, bias=None, residual=None, ldr=0, accumulate=0):
        _hip.call("vlr_gemm_bf16", layout, A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, 0, accumulate, 0)

    def resampler_fwd(self, ws, x, n, tag, save):
        """attn_pool + ln_post + proj on the trunk output x [n*T, width] with the resampler weights of `ws` (visual.py:136-150,410-413):
        one cross-attention from the learned queries to the patch tokens, run on the SELF-attention kernel over [queries | keys] per image
        with a key mask hiding the query rows.  Position terms enter as residual operands: Q = ln_q(query) Wq^T + (pos_q Wq^T + bq),
        K_i = ln_kv(kv_i) Wk^T + (pos_k Wk^T + bk).  Returns (features [n*nq, E], state kept for the backward or None)."""
        vw = self.vision
        T, W, E, nq = vw.T, self.cfg["visual"]["width"], vw.E, vw.nq
        M, Sx = n * T, nq + T
        v_ = ws.v
        win, b_in = v_["ap.win"], v_["ap.bin"]
        kv = self._buf((tag, "qv_kv", n), (M, E))
        self._g(0, x, v_["ap.kv"], kv, M, E, W, W, W, E)
        kvn = self._buf((tag, "qv_kvn", n), (M, E))
        _hip.call("vlr_layernorm_fwd", kv, v_["ap.lnkv_w"], v_["ap.lnkv_b"], kvn, M, E, 1e-6)
        qn = self._buf((tag, "qv_qn"), (nq, E))
        _hip.call("vlr_layernorm_fwd", v_["ap.query"], v_["ap.lnq_w"], v_["ap.lnq_b"], qn, nq, E, 1e-6)
        tq, tk = self._buf((tag, "qv_tq"), (nq, E)), self._buf((tag, "qv_tk"), (T, E))
        self._g(0, vw.pos_q, win[:E], tq, nq, E, E, E, E, E, bias=b_in[:E])              # pos_q Wq^T + bq
        self._g(0, vw.pos_k, win[E:2 * E], tk, T, E, E, E, E, E, bias=b_in[E:2 * E])     # pos_k Wk^T + bk
        key = (tag, "qv_xattn", n)
        if key not in self._ws:        # zero once: K / V of the query rows and Q of the key rows are never written and must stay finite
            self._ws[key] = torch.zeros(n * Sx, 3 * E, dtype=BF16, device=self.dev)
            km = torch.ones(n, Sx, dtype=torch.int32, device=self.dev)
            km[:, :nq] = 0                 # queries attend to the patch tokens only
            self._ws[(tag, "qv_xmask", n)] = km
            self._ws[(tag, "qv_xrows", n)] = (torch.arange(n, device=self.dev, dtype=torch.int32)[:, None] * Sx
                                         + torch.arange(nq, device=self.dev, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
            self._ws[(tag, "qv_krows", n)] = (torch.arange(n, device=self.dev, dtype=torch.int32)[:, None] * Sx + nq
                                         + torch.arange(T, device=self.dev, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
        qkv, km, rows = self._ws[key], self._ws[(tag, "qv_xmask", n)], self._ws[(tag, "qv_xrows", n)]
        for i in range(n):
            self._g(0, qn, win[:E], qkv[i * Sx: i * Sx + nq, :E], nq, E, E, E, E, 3 * E, residual=tq, ldr=E)
            kr = qkv[i * Sx + nq:(i + 1) * Sx]
            self._g(0, kvn[i * T:(i + 1) * T], win[E:2 * E], kr[:, E:2 * E], T, E, E, E, E, 3 * E, residual=tk, ldr=E)
            self._g(0, kvn[i * T:(i + 1) * T], win[2 * E:], kr[:, 2 * E:], T, E, E, E, E, 3 * E, bias=b_in[2 * E:])
        o = self._buf((tag, "qv_xo", n), (n * Sx, E))
        Sp = _align(Sx, 64)
        lse = self._buf((tag, "qv_lse", n), (n, vw.heads, Sp), torch.float32) if save else None
        _hip.call("vlr_attn_fwd", qkv, qkv[:, E:], qkv[:, 2 * E:], 3 * E, o, E, lse, km, n, Sx, vw.heads, 128, 0, 1.0 / math.sqrt(128.0))
        oq = self._buf((tag, "qv_oq", n), (n * nq, E))
        _hip.call("vlr_gather_rows", o, rows, oq, n * nq, E)
        op = self._buf((tag, "qv_op", n), (n * nq, E))
        self._g(0, oq, v_["ap.wo"], op, n * nq, E, E, E, E, E, bias=v_["ap.bo"])
        on = self._buf((tag, "qv_on", n), (n * nq, E))
        _hip.call("vlr_layernorm_fwd", op, vw.post_w, vw.post_b, on, n * nq, E, 1e-6)
        feat = torch.empty(n * nq, E, dtype=BF16, device=self.dev)
        self._g(0, on, vw.proj_t, feat, n * nq, E, E, E, E, E)
        if not save:
            return feat, None
        return feat, dict(n=n, x=x, kv=kv, kvn=kvn, qn=qn, qkv=qkv, km=km, o=o, lse=lse, oq=oq, op=op, tag=tag)

    def resampler_bwd(self, ws, st, dfeats, acc):
        """gradients of the resampler weights (`ap.*`) from d features [n*nq, E]; the ViT trunk is frozen, so nothing flows past kv_proj"""
        vw, v_, gv = self.vision, ws.v, self.gv
        n, tag = st["n"], st["tag"]
        T, W, E, nq = vw.T, self.cfg["visual"]["width"], vw.E, vw.nq
        M, Sx, R = n * T, nq + T, n * nq
        win = v_["ap.win"]
        B = lambda k, shape, dt=BF16: self._buf((tag, "qvb_" + k, n), shape, dt)   # noqa: E731
        lnws = self._buf(("qvb_lnws",), (_hip.helper("vlr_layernorm_bwd_workspace_bytes", E),), torch.uint8)
        d_on = B("don", (R, E))
        self._g(1, dfeats, vw.proj_t, d_on, R, E, E, E, E, E)                               # proj is frozen: data gradient only
        d_op = B("dop", (R, E))
        _hip.call("vlr_layernorm_bwd", d_on, st["op"], vw.post_w, 1e-6, d_op, None, None, 0, lnws, R, E)
        self._g(2, d_op, st["oq"], gv["ap.wo"], E, E, R, E, E, E, accumulate=acc)           # out_proj
        _hip.call("vlr_colsum", d_op, R, E, E, gv["ap.bo"], acc, self._colsum_ws)
        d_oq = B("doq", (R, E))
        self._g(1, d_op, v_["ap.wo"], d_oq, R, E, E, E, E, E)
        do = B("do", (n * Sx, E))
        do.zero_()
        _hip.call("vlr_scatter_rows", d_oq, self._ws[(tag, "qv_xrows", n)], do, R, E)
        dqkv = B("dqkv", (n * Sx, 3 * E))
        delta = B("delta", (n, vw.heads, _align(Sx, 64)), torch.float32)
        qkv = st["qkv"]
        _hip.call("vlr_attn_bwd", qkv, qkv[:, E:], qkv[:, 2 * E:], 3 * E, st["o"], do, E, st["lse"], delta, st["km"], dqkv, dqkv[:, E:],
                  dqkv[:, 2 * E:], 3 * E, n, Sx, vw.heads, 128, 0, 1.0 / math.sqrt(128.0))
        # ---- queries: dQ summed over the images (the same learned queries serve every image)
        dQ = B("dQ", (nq, E))
        ar = self._ws.get(("qv_ar", nq))
        if ar is None:
            ar = self._ws[("qv_ar", nq)] = torch.arange(nq, device=self.dev, dtype=torch.int32)
        art = self._ws.get(("qv_ar", T))
        if art is None:
            art = self._ws[("qv_ar", T)] = torch.arange(T, device=self.dev, dtype=torch.int32)
        tmpq = B("tmpq", (nq, E))
        for i in range(n):
            dst = dQ if i == 0 else tmpq
            _hip.call("vlr_rows_gather", dqkv[i * Sx:], 3 * E, ar, dst, nq, E)
            if i:
                _hip.call("vlr_rows_add", tmpq, ar, dQ, E, nq, E)
        gq, gk, gvv = gv["ap.win"][:E], gv["ap.win"][E:2 * E], gv["ap.win"][2 * E:]
        self._g(2, dQ, st["qn"], gq, E, E, nq, E, E, E, accumulate=acc)                      # dWq = dQ^T (ln_q(query) + pos_q)
        self._g(2, dQ, vw.pos_q, gq, E, E, nq, E, E, E, accumulate=1)
        _hip.call("vlr_colsum", dQ, nq, E, E, gv["ap.bin"][:E], acc, self._colsum_ws)
        d_qn = B("dqn", (nq, E))
        self._g(1, dQ, win[:E], d_qn, nq, E, E, E, E, E)
        _hip.call("vlr_layernorm_bwd", d_qn, v_["ap.query"], v_["ap.lnq_w"], 1e-6, gv["ap.query"] if not acc else tmpq, gv["ap.lnq_w"],
                  gv["ap.lnq_b"], acc, lnws, nq, E)
        if acc:
            _hip.call("vlr_rows_add", tmpq, ar, gv["ap.query"], E, nq, E)
        # ---- keys / values
        krows = self._ws[(tag, "qv_krows", n)]
        dK, dV = B("dK", (M, E)), B("dV", (M, E))
        _hip.call("vlr_rows_gather", dqkv[:, E:], 3 * E, krows, dK, M, E)
        _hip.call("vlr_rows_gather", dqkv[:, 2 * E:], 3 * E, krows, dV, M, E)
        dKs = B("dKs", (T, E))
        tmpk = B("tmpk", (T, E))
        for i in range(n):
            if i == 0:
                dKs.copy_(dK[:T])
            else:
                tmpk.copy_(dK[i * T:(i + 1) * T])
                _hip.call("vlr_rows_add", tmpk, art, dKs, E, T, E)
        self._g(2, dK, st["kvn"], gk, E, E, M, E, E, E, accumulate=acc)                      # dWk = sum_i dK_i^T (ln_kv(kv_i) + pos_k)
        self._g(2, dKs, vw.pos_k, gk, E, E, T, E, E, E, accumulate=1)
        _hip.call("vlr_colsum", dK, M, E, E, gv["ap.bin"][E:2 * E], acc, self._colsum_ws)
        self._g(2, dV, st["kvn"], gvv, E, E, M, E, E, E, accumulate=acc)
        _hip.call("vlr_colsum", dV, M, E, E, gv["ap.bin"][2 * E:], acc, self._colsum_ws)
        d_kvn = B("dkvn", (M, E))
        self._g(1, dK, win[E:2 * E], d_kvn, M, E, E, E, E, E)
        self._g(1, dV, win[2 * E:], d_kvn, M, E, E, E, E, E, accumulate=1)
        d_kv = B("dkv", (M, E))
        _hip.call("vlr_layernorm_bwd", d_kvn, st["kv"], v_["ap.lnkv_w"], 1e-6, d_kv, gv["ap.lnkv_w"], gv["ap.lnkv_b"], acc, lnws, M, E)
        self._g(2, d_kv, st["x"], gv["ap.kv"], E, W, M, E, W, W, accumulate=acc)             # kv_proj; the trunk below it is frozen

    def vision_features(self, pixel_values: torch.Tensor, key=None) -> torch.Tensor:
        """resampler output [n * n_queries, hidden] with the policy's resampler weights (no-grad helper: tests, prefetch)"""
        x = self.vit_trunk(pixel_values)
        return self.resampler_fwd(self.policy, x, pixel_values.shape[0], "vf", False)[0]

    # ------------------------------------------------------------------------------------------------ embed / merge
    def _embed_inputs(self, ws, ids, am, lab, pixel_values, image_dup, tag, image_sizes, meta):
        """modeling_qwen.py:525-537, 616-625: the slots a+1 .. b-1 of every (<img>, </img>) pair take the resampler output of the
        images in order of appearance; S = T, positions = index in the sequence, labels unchanged."""
        c = self.cfg
        Bn, T = ids.shape
        S, M, nq = T, Bn * T, self.nq
        cached = meta.get("qwen") if meta is not None else None
        if cached is None:
            idh = ids.cpu().numpy()
            st = int(c["image_start_id"])
            spans = []
            for i in range(Bn):
                bos, eos = np.nonzero(idh[i] == st)[0], np.nonzero(idh[i] == st + 1)[0]
                if len(bos) != len(eos):
                    raise ValueError(f"row {i}: {len(bos)} <img> markers but {len(eos)} </img> markers")
                for a_, b_ in zip(bos, eos):
                    if b_ - a_ - 1 != nq:
                        raise ValueError(f"row {i}: {b_ - a_ - 1} slots between <img> and </img>, the resampler yields {nq}")
                    spans.append((i, int(a_), int(b_)))
            n_img = len(spans)
            if image_dup > 1 and n_img % image_dup:
                raise ValueError(f"{n_img} image spans cannot be {image_dup} copies of one batch")
            n_uniq = n_img // max(1, image_dup) if n_img else 0
            src = np.tile(np.arange(T, dtype=np.int64), (Bn, 1))
            img_map = np.zeros((Bn, T), dtype=bool)
            inv = np.full((max(1, image_dup), max(1, n_uniq * nq)), -1, dtype=np.int32)
            for k, (i, a_, b_) in enumerate(spans):
                u = k % n_uniq
                src[i, a_ + 1:b_] = -(u * nq + np.arange(nq) + 1)
                img_map[i, a_ + 1:b_] = True
                inv[k // n_uniq, u * nq:(u + 1) * nq] = i * S + a_ + 1 + np.arange(nq)
            dv = lambda a_: torch.from_numpy(a_).to(self.dev)    # noqa: E731
            cached = dict(n_img=n_img, n_uniq=n_uniq, src=dv(src.astype(np.int32)), img_map=dv(img_map), inv=dv(inv),
                          pos=torch.arange(S, dtype=torch.int32, device=self.dev)[None].expand(Bn, S).contiguous())
            if meta is not None:
                meta["qwen"] = cached
        n_img, n_uniq = cached["n_img"], cached["n_uniq"]
        feats, res_state = None, None
        if n_img:
            if pixel_values is None or pixel_values.shape[0] != n_img:
                raise ValueError(f"The input provided to the model are wrong. The ids name {n_img} images while "
                                 f"{0 if pixel_values is None else pixel_values.shape[0]} were given to the model.")
            px = pixel_values[:n_uniq] if image_dup > 1 else pixel_values
            train_ap = self.lora is None and ws is self.policy and tag == "policy"          # full fine-tune: the resampler is trainable
            feats, res_state = self.resampler_fwd(ws, self.vit_trunk(px), px.shape[0], tag, train_ap)
        else:
            feats = self._buf((tag, "nofeat"), (8, self.H))      # never read: no negative src entry
        mask = am.to(torch.int32).contiguous()
        mlabels = lab.clone() if lab is not None else torch.full((Bn, S), -100, dtype=torch.long, device=self.dev)
        return dict(S=S, M=M, src=cached["src"], mask=mask, pos=cached["pos"], labels=mlabels, img_map=cached["img_map"],
                    inv=cached["inv"], feats=feats, vit_feat=None, proj_z=None, proj_h=None, n_rows=max(1, n_uniq * nq),
                    n_feat=n_uniq * nq, pack=None, extra=res_state)

    def _embed_backward(self, ctx, cur, acc):
        """wte and - in a full fine-tune - the resampler are trainable in front of the decoder; the ViT trunk, ln_post and proj are frozen"""
        if not acc:
            self.gv["embed"].zero_()
        st = ctx.get("extra")
        dfeats = self._buf(("dfeats", ctx["n_rows"]), (ctx["n_rows"], self.H)) if st is not None else None
        _hip.call("vlr_merge_bwd", cur, ctx["src"], ctx["inv"], ctx["ids"], dfeats, self.gv["embed"], ctx["Bn"], ctx["T"], ctx["S"], self.H,
                  ctx["n_rows"], ctx["image_dup"])
        if st is not None:
            self.resampler_bwd(ctx["ws"], st, dfeats, acc)
        elif not acc:
            for k in self.gv:
                if k.startswith("ap."):
                    self.gv[k].zero_()            # a step without images leaves the resampler untouched
