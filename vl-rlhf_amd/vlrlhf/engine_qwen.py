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
The whole tower is frozen on this path (scripts/dpo_qwenvl.sh: --freeze_vision_tower True + LoRA: peft freezes `attn_pool` too),
so only its forward exists; full fine-tuning trains the language model (the reference would also train `attn_pool` there - not built).
"""
import math
from typing import Dict, Optional

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
        # ---- resampler: everything that does not depend on the image is folded at load time (fp32 on the host)
        a = "attn_pool."
        self.E, self.nq, self.heads = E, int(sd[prefix + a + "query"].shape[0]), E // 128
        self.kv_w = dv(g(a + "kv_proj.weight")) if (prefix + a + "kv_proj.weight") in sd else None     # nn.Identity when width == output_dim
        self.lnkv_w, self.lnkv_b = dv(g(a + "ln_kv.weight")), dv(g(a + "ln_kv.bias"))
        wi, bi = g(a + "attn.in_proj_weight"), g(a + "attn.in_proj_bias")
        pos_q = g(a + "pos_embed")
        pos_k = get_abs_pos(pos_q, self.T)
        qn = F.layer_norm(g(a + "query"), (E,), g(a + "ln_q.weight"), g(a + "ln_q.bias"), 1e-6)
        self.q_proj = dv((qn + pos_q) @ wi[:E].t() + bi[:E])                               # [nq, E]: the projected queries, same for every image
        self.wk, self.wv = dv(wi[E:2 * E]), dv(wi[2 * E:])
        self.k_pos = dv(pos_k @ wi[E:2 * E].t() + bi[E:2 * E])                             # [T, E]: (pos_k Wk^T + bk), added as a residual
        self.bv = dv(bi[2 * E:])
        self.wo, self.bo = dv(g(a + "attn.out_proj.weight")), dv(g(a + "attn.out_proj.bias"))
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

    def vision_features(self, pixel_values: torch.Tensor, key=None) -> torch.Tensor:
        """[n,3,s,s] -> resampler output [n * n_queries, hidden] bf16 (visual.py:393-415), cached per pixel tensor so that the
        reference pass and the policy pass share one evaluation"""
        if key is None:
            key = (pixel_values.data_ptr(), tuple(pixel_values.shape), pixel_values._version)
        if self._vit_cache is not None and self._vit_cache[0] == key:
            return self._vit_cache[1]
        vw, v = self.vision, self.cfg["visual"]
        n, T, W, E, nq = pixel_values.shape[0], vw.T, v["width"], vw.E, vw.nq
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
        x = self._buf(("qv_x", n), (M, W))
        _hip.call("vlr_layernorm_fwd", pe, vw.pre_w, vw.pre_b, x, M, W, 1e-6)
        wsb = dict(xn=self._buf(("qv_xn", n), (M, W)), qkv=self._buf(("qv_qkv", n), (M, 3 * A)),
                   attn=self._buf(("qv_attn", n), (M, A)), h=self._buf(("qv_h", n), (M, vw.mlp)))
        ws = _hip.VitWs(*(wsb[k].data_ptr() for k in ("xn", "qkv", "attn", "h")))
        for lw in vw.layers:
            _hip.call("vlr_vit_layer_fwd", self.vit_cfg, lw, ws, x, n, T)
        # ---- resampler
        if vw.kv_w is not None:
            kv = self._buf(("qv_kv", n), (M, E))
            _hip.call("vlr_gemm_bf16", 0, x, vw.kv_w, kv, None, None, M, E, W, W, W, E, 0, 0, 0, 0)
        else:
            kv = x
        kvn = self._buf(("qv_kvn", n), (M, E))
        _hip.call("vlr_layernorm_fwd", kv, vw.lnkv_w, vw.lnkv_b, kvn, M, E, 1e-6)
        Sx = nq + T                                    # combined sequence of one image: [queries | patch tokens]
        qkv = self._ws.get(("qv_xattn", n))
        if qkv is None:                                # zero once: the K / V rows of the query positions and the Q rows of the key
            qkv = torch.zeros(n * Sx, 3 * E, dtype=BF16, device=self.dev)      # positions are never written and must stay finite
            km = torch.ones(n, Sx, dtype=torch.int32, device=self.dev)
            km[:, :nq] = 0                             # queries attend to the patch tokens only
            rows = (torch.arange(n, device=self.dev, dtype=torch.int32)[:, None] * Sx
                    + torch.arange(nq, device=self.dev, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
            for i in range(n):
                qkv[i * Sx: i * Sx + nq, :E].copy_(vw.q_proj)
            self._ws[("qv_xattn", n)] = qkv
            self._ws[("qv_xmask", n)] = km
            self._ws[("qv_xrows", n)] = rows
        km, rows = self._ws[("qv_xmask", n)], self._ws[("qv_xrows", n)]
        for i in range(n):                             # K = ln_kv(kv) Wk^T + (pos_k Wk^T + bk),  V = ln_kv(kv) Wv^T + bv, written in place
            kr = qkv[i * Sx + nq:(i + 1) * Sx]
            _hip.call("vlr_gemm_bf16", 0, kvn[i * T:(i + 1) * T], vw.wk, kr[:, E:2 * E], None, vw.k_pos, T, E, E, E, E, 3 * E, E, 0, 0, 0)
            _hip.call("vlr_gemm_bf16", 0, kvn[i * T:(i + 1) * T], vw.wv, kr[:, 2 * E:], vw.bv, None, T, E, E, E, E, 3 * E, 0, 0, 0, 0)
        o = self._buf(("qv_xo", n), (n * Sx, E))
        _hip.call("vlr_attn_fwd", qkv, qkv[:, E:], qkv[:, 2 * E:], 3 * E, o, E, None, km, n, Sx, vw.heads, 128, 0, 1.0 / math.sqrt(128.0))
        oq = self._buf(("qv_oq", n), (n * nq, E))
        _hip.call("vlr_gather_rows", o, rows, oq, n * nq, E)
        op = self._buf(("qv_op", n), (n * nq, E))
        _hip.call("vlr_gemm_bf16", 0, oq, vw.wo, op, vw.bo, None, n * nq, E, E, E, E, E, 0, 0, 0, 0)
        on = self._buf(("qv_on", n), (n * nq, E))
        _hip.call("vlr_layernorm_fwd", op, vw.post_w, vw.post_b, on, n * nq, E, 1e-6)
        feat = torch.empty(n * nq, E, dtype=BF16, device=self.dev)
        _hip.call("vlr_gemm_bf16", 0, on, vw.proj_t, feat, None, None, n * nq, E, E, E, E, E, 0, 0, 0, 0)
        self._vit_cache = (key, feat, pixel_values)
        return feat

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
        feats = None
        if n_img:
            if pixel_values is None or pixel_values.shape[0] != n_img:
                raise ValueError(f"The input provided to the model are wrong. The ids name {n_img} images while "
                                 f"{0 if pixel_values is None else pixel_values.shape[0]} were given to the model.")
            feats = self.vision_features(pixel_values[:n_uniq] if image_dup > 1 else pixel_values)
        else:
            feats = self._buf((tag, "nofeat"), (8, self.H))      # never read: no negative src entry
        mask = am.to(torch.int32).contiguous()
        mlabels = lab.clone() if lab is not None else torch.full((Bn, S), -100, dtype=torch.long, device=self.dev)
        return dict(S=S, M=M, src=cached["src"], mask=mask, pos=cached["pos"], labels=mlabels, img_map=cached["img_map"],
                    inv=cached["inv"], feats=feats, vit_feat=None, proj_z=None, proj_h=None, n_rows=max(1, n_uniq * nq),
                    n_feat=n_uniq * nq, pack=None)

    def _embed_backward(self, ctx, cur, acc):
        """only wte is trainable in front of the decoder (the vision tower incl. the resampler is frozen on this path)"""
        if not acc:
            self.gv["embed"].zero_()
        _hip.call("vlr_merge_bwd", cur, ctx["src"], ctx["inv"], ctx["ids"], None, self.gv["embed"], ctx["Bn"], ctx["T"], ctx["S"], self.H,
                  ctx["n_rows"], ctx["image_dup"])
