"""Synthetic LLaVA-shaped weights and DPO batches on the device (benchmarks / smoke runs: no checkpoints, no network).
Workload definition: SURVEY.md section 8(d)."""
import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

LLAVA_1_5_7B = dict(vit_hidden=1024, vit_mlp=4096, vit_layers=24, vit_heads=16, image_size=336, patch_size=14,
                    hidden=4096, inter=11008, layers=32, heads=32, vocab=32064, image_token=32000,
                    model_pad_token_id=32001, rms_eps=1e-5, rope_theta=10000.0)


_M32 = 0xFFFFFFFF


def _hash32(x, key):
    x = (x ^ key) & _M32
    x = (x * 0x45D9F3B) & _M32          # operands < 2^32 and < 2^27: the int64 product cannot overflow
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & _M32
    return x ^ (x >> 16)


def hashed_normal(n, seed, name, device):
    """~N(0,1) fp32 [n] as a pure integer function of (seed, name, index): sum of the four 16-bit halves of two 32-bit
    hashes (Irwin-Hall, +-3.46 sigma), then ONE fp32 multiply.  Bit-identical on any device, so full-size (7B) parity
    runs regenerate their weights where they are needed instead of shipping them."""
    import hashlib
    d = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    k1, k2 = int.from_bytes(d[:4], "little"), int.from_bytes(d[4:8], "little")
    out = torch.empty(n, dtype=torch.float32, device=device)
    step = 1 << 26
    for a in range(0, n, step):
        i = torch.arange(a, min(n, a + step), dtype=torch.int64, device=device)
        h1, h2 = _hash32(i, k1), _hash32((i + 0x9E3779B9) & _M32, k2)
        u = (h1 & 0xFFFF) + (h1 >> 16) + (h2 & 0xFFFF) + (h2 >> 16)
        out[a:a + step] = (u - 131070).to(torch.float32) * (1.0 / 37837.227)
    return out


def init_hashed_model(model, seed=0, std=0.02, policy_delta=1e-3, seed_delta=1, qk_scale=1.0, ref=None, outliers=None):
    """Weights named by their HF checkpoint keys and drawn by hashed_normal: reference = bf16(std * n) (norm gains
    1 + 0.05 n), policy = bf16(reference + policy_delta * n').  Returns the reference model."""
    from ..engine import VisionWeights
    eng = model.engine
    dev = eng.dev

    def draw(name, shape, delta):
        numel = 1
        for s_ in shape:
            numel *= s_
        n = hashed_normal(numel, seed, name, dev).view(*shape)
        gain = name.endswith(("norm.weight", "norm1.weight", "norm2.weight", "layrnorm.weight"))
        qk = qk_scale != 1.0 and name.startswith("language_model.") and name.endswith(("q_proj.weight", "k_proj.weight"))
        t = ((n * 0.05 + 1.0) if gain else n * (std * qk_scale if qk else std)).to(torch.bfloat16)      # (qk_scale: oracle.HashedWeights)
        if delta > 0 and not name.startswith(("vision_tower.", "vit.", "vision_proj.")):
            t = (t.float() + hashed_normal(numel, seed_delta, name, dev).view(*shape) * delta).to(torch.bfloat16)
        # planted outlier channels (oracle.HashedWeights(outliers=...)): rows of mlp.down_proj of the named layers x a power of two - exact in bf16
        if outliers and name.endswith("mlp.down_proj.weight") and name.startswith("language_model.model.layers.") and int(name.split(".")[3]) in outliers["layers"]:
            t[list(outliers["channels"])] *= float(outliers["scale"])
        return t

    if ref is None:
        ref = model.create_reference_model()
    for ws, delta in ((ref.weights, 0.0), (eng.policy, policy_delta)):
        for hf, name, r0, rows in eng.layout.hf_names():
            dst = ws.v[name]
            if dst.dim() == 1:
                dst.copy_(draw(hf, tuple(dst.shape), delta))
            else:
                dst[r0:r0 + rows].copy_(draw(hf, (rows, dst.shape[1]), delta))
    c = eng.cfg
    D, P, F = c["vit_hidden"], c["patch_size"], c["vit_mlp"]
    T = (c["image_size"] // P) ** 2 + 1
    vp = getattr(eng, "vision_prefix", "vision_tower.") + "vision_model."
    sh = {vp + "embeddings.class_embedding": (D,), vp + "embeddings.patch_embedding.weight": (D, 3, P, P),
          vp + "embeddings.position_embedding.weight": (T, D), vp + "pre_layrnorm.weight": (D,), vp + "pre_layrnorm.bias": (D,)}
    for i in range(c["vit_layers"] + 1 + int(c.get("vit_feature_layer", -2))):
        p = f"{vp}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sh[p + f"self_attn.{nm}.weight"], sh[p + f"self_attn.{nm}.bias"] = (D, D), (D,)
        for nm in ("layer_norm1", "layer_norm2"):
            sh[p + nm + ".weight"], sh[p + nm + ".bias"] = (D,), (D,)
        sh[p + "mlp.fc1.weight"], sh[p + "mlp.fc1.bias"] = (F, D), (F,)
        sh[p + "mlp.fc2.weight"], sh[p + "mlp.fc2.bias"] = (D, F), (D,)
    eng.vision = VisionWeights(c, {k: draw(k, s_, 0.0) for k, s_ in sh.items()}, dev, prefix=vp)
    eng._vit_cache = None
    return ref


def vision_state_dict(cfg, device, seed=0, std=0.02, prefix="vision_tower.vision_model."):
    g = torch.Generator(device=device).manual_seed(seed)
    D, P, F = cfg["vit_hidden"], cfg["patch_size"], cfg["vit_mlp"]
    T = (cfg["image_size"] // P) ** 2 + 1

    def rnd(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device) * s).to(torch.bfloat16)

    sd = {prefix + "embeddings.class_embedding": rnd(D), prefix + "embeddings.patch_embedding.weight": rnd(D, 3, P, P),
          prefix + "embeddings.position_embedding.weight": rnd(T, D)}
    for nm in ("pre_layrnorm", "post_layernorm"):
        sd[prefix + nm + ".weight"] = (1 + rnd(D, s=0.05).float()).to(torch.bfloat16)
        sd[prefix + nm + ".bias"] = rnd(D)
    for i in range(cfg["vit_layers"]):
        p = f"{prefix}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"] = rnd(D, D)
            sd[p + f"self_attn.{nm}.bias"] = rnd(D)
        for nm in ("layer_norm1", "layer_norm2"):
            sd[p + nm + ".weight"] = (1 + rnd(D, s=0.05).float()).to(torch.bfloat16)
            sd[p + nm + ".bias"] = rnd(D)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = rnd(F, D), rnd(F)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = rnd(D, F), rnd(D)
    return sd


def init_random_model(model, seed=0, std=0.02, policy_delta=1e-3):
    """N(0, std) weights (norm gains 1), written straight into the engine's flat buffers; returns a reference model
    whose LLM/projector weights differ from the policy by N(0, policy_delta) (avoids the loss == ln 2 degenerate case)."""
    from ..engine import VisionWeights
    eng = model.engine
    g = torch.Generator(device=eng.dev).manual_seed(seed)
    flat = eng.policy.flat
    step = 1 << 28
    for a in range(0, flat.numel(), step):
        b = min(flat.numel(), a + step)
        flat[a:b] = (torch.randn(b - a, generator=g, device=eng.dev) * std).to(torch.bfloat16)
    for name, shape in eng.layout.shape.items():
        if len(shape) == 1 and (name.endswith("ln1") or name.endswith("ln2") or name == "norm"):
            eng.policy.v[name].fill_(1.0)
    eng.vision = VisionWeights(eng.cfg, vision_state_dict(eng.cfg, eng.dev, seed + 1, std), eng.dev)
    ref = model.create_reference_model()
    for a in range(0, flat.numel(), step):
        b = min(flat.numel(), a + step)
        ref.weights.flat[a:b] = (flat[a:b].float() - torch.randn(b - a, generator=g, device=eng.dev) * policy_delta).to(torch.bfloat16)
    return ref


def synthetic_pixels(n, image_size, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    px = g.integers(0, 256, size=(n, 3, image_size, image_size), dtype=np.uint8).astype(np.float32) / 255.0
    mean = np.array(CLIP_MEAN, dtype=np.float32)[None, :, None, None]
    std = np.array(CLIP_STD, dtype=np.float32)[None, :, None, None]
    return torch.from_numpy((px - mean) / std)


def synthetic_batch(pairs, text_len, image_token, vocab_hi, image_size, seed, ragged=False, prompt_frac=0.5):
    """BOS at 0, one <image> at index 4, ids U{3..vocab_hi-1}; first half = prompt shared by chosen and rejected;
    labels = ids with the prompt -> -100.  ragged: response lengths U{..}, right-padded 0 / -100 / 0."""
    from ..base.collator import VLDPODataCollatorWithPadding
    g = np.random.Generator(np.random.PCG64(seed))
    lp = int(text_len * prompt_frac)
    rows = []
    for _ in range(pairs):
        prompt = g.integers(3, vocab_hi, size=lp).tolist()
        prompt[0], prompt[4] = 1, image_token
        lens = [text_len - lp, text_len - lp]
        if ragged:
            lens = [int(g.integers(max(2, (text_len - lp) // 8), text_len - lp + 1)) for _ in range(2)]
        resp = [g.integers(3, vocab_hi, size=n).tolist() for n in lens]
        rows.append(dict(prompt_input_ids=prompt, prompt_attention_mask=[1] * lp,
                         chosen_input_ids=prompt + resp[0], chosen_attention_mask=[1] * (lp + lens[0]),
                         chosen_labels=[-100] * lp + resp[0], rejected_input_ids=prompt + resp[1],
                         rejected_attention_mask=[1] * (lp + lens[1]), rejected_labels=[-100] * lp + resp[1],
                         img_path="synthetic"))
    batch = VLDPODataCollatorWithPadding()(rows)
    batch["img_input_dict"] = dict(pixel_values=synthetic_pixels(pairs, image_size, seed + 7))
    return batch


def synthetic_batch_anyres(pairs, text_len, image_token, vocab_hi, image_size, seed, image_hw=(672, 672), grid_pinpoints=None,
                           ragged=False, prompt_frac=0.5):
    """LLaVA-Next variant of synthetic_batch: the same token recipe, every image of original size `image_hw` (height, width) given
    as its anyres tiles [pairs, tiles, 3, s, s] + image_sizes [pairs, 2] (672x672 -> base + 2x2 tiles = 5, 2928 features)."""
    from ..models.LlavaNext import LLAVA_NEXT_MISTRAL_7B, anyres
    batch = synthetic_batch(pairs, text_len, image_token, vocab_hi, image_size, seed, ragged=ragged, prompt_frac=prompt_frac)
    pins = grid_pinpoints or LLAVA_NEXT_MISTRAL_7B["image_grid_pinpoints"]
    n = anyres.image_size_to_num_patches(image_hw, pins, image_size)
    tiles = torch.stack([synthetic_pixels(n, image_size, seed + 11 + i) for i in range(pairs)])
    batch["img_input_dict"] = dict(pixel_values=tiles, image_sizes=torch.tensor([list(image_hw)] * pairs, dtype=torch.long))
    return batch


# ---------------------------------------------------------------------------------------------------- Qwen-VL (BASELINE configs[2])
def qwen_vision_shapes(vcfg, prefix="transformer.visual."):
    """tensor shapes of the Qwen-VL vision tower's state_dict (reference models/QwenVL/visual.py:330-390)"""
    W, E, L, P = vcfg["width"], vcfg["output_dim"], vcfg["layers"], vcfg["patch_size"]
    F = int(W * vcfg["mlp_ratio"])
    nq = int(vcfg.get("n_queries", 256))
    sh = {prefix + "positional_embedding": (256, W), prefix + "proj": (E, E), prefix + "conv1.weight": (W, 3, P, P),
          prefix + "ln_pre.weight": (W,), prefix + "ln_pre.bias": (W,), prefix + "ln_post.weight": (E,), prefix + "ln_post.bias": (E,)}
    for i in range(L):
        p = f"{prefix}transformer.resblocks.{i}."
        for nm in ("ln_1", "ln_2"):
            sh[p + nm + ".weight"], sh[p + nm + ".bias"] = (W,), (W,)
        sh[p + "attn.in_proj.weight"], sh[p + "attn.in_proj.bias"] = (3 * W, W), (3 * W,)
        sh[p + "attn.out_proj.weight"], sh[p + "attn.out_proj.bias"] = (W, W), (W,)
        sh[p + "mlp.c_fc.weight"], sh[p + "mlp.c_fc.bias"] = (F, W), (F,)
        sh[p + "mlp.c_proj.weight"], sh[p + "mlp.c_proj.bias"] = (W, F), (W,)
    a = prefix + "attn_pool."
    sh.update({a + "pos_embed": (nq, E), a + "query": (nq, E), a + "kv_proj.weight": (E, W), a + "attn.in_proj_weight": (3 * E, E),
               a + "attn.in_proj_bias": (3 * E,), a + "attn.out_proj.weight": (E, E), a + "attn.out_proj.bias": (E,),
               a + "ln_q.weight": (E,), a + "ln_q.bias": (E,), a + "ln_kv.weight": (E,), a + "ln_kv.bias": (E,)})
    return sh


def init_hashed_qwen(model, seed=0, std=0.02, policy_delta=1e-3, seed_delta=1, with_reference=True):
    """hashed_normal weights for a QwenVLForRL (names = the reference checkpoint's keys).  Returns the reference model (or None)."""
    eng = model.engine
    dev = eng.dev

    def draw(name, shape, delta):
        numel = 1
        for s_ in shape:
            numel *= s_
        n = hashed_normal(numel, seed, name, dev).view(*shape)
        gain = name.endswith((".weight",)) and (".ln_" in name or "ln_f" in name or "ln_pre" in name or "ln_post" in name)
        t = ((n * 0.05 + 1.0) if gain else n * std).to(torch.bfloat16)
        if delta > 0 and not name.startswith("transformer.visual."):
            t = (t.float() + hashed_normal(numel, seed_delta, name, dev).view(*shape) * delta).to(torch.bfloat16)
        return t

    ref = model.create_reference_model() if with_reference else None
    for ws, delta in (((ref.weights, 0.0),) if ref is not None else ()) + ((eng.policy, policy_delta if ref is not None else 0.0),):
        for hf, name, r0, rows in eng.layout.hf_names():
            dst = ws.v[name]
            if dst.dim() == 1:
                dst.copy_(draw(hf, tuple(dst.shape), delta))
            else:
                dst[r0:r0 + rows].copy_(draw(hf, (rows, dst.shape[1]), delta))
    eng.vision = eng._load_vision({k: draw(k, s_, 0.0) for k, s_ in qwen_vision_shapes(eng.cfg["visual"]).items()})
    eng._vit_cache = None
    return ref


def synthetic_batch_qwen(pairs, text_len, cfg, seed, prompt_frac=0.5):
    """`pairs` preference pairs of `text_len` ids each (the 258 ids of one <img>...</img> slot included, as the Qwen tokenizer emits
    them); chosen / rejected share prompt and image; responses of different lengths so that right padding occurs."""
    g = torch.Generator().manual_seed(seed)
    st, nq, s = cfg["image_start_id"], cfg["visual"].get("n_queries", 256), cfg["visual"]["image_size"]
    lo_txt = 256                                          # ids below 256 are raw bytes (image paths)
    vocab_hi = min(cfg["vocab"], st) - 1
    n_prompt = max(nq + 8, int(text_len * prompt_frac))
    batch = {k: [] for k in ("chosen_input_ids", "chosen_attention_mask", "chosen_labels", "rejected_input_ids", "rejected_attention_mask",
                             "rejected_labels")}
    pad = cfg.get("pad_token_id", 0)
    for i in range(pairs):
        path = list(f"synthetic/{seed}/{i}.png".encode())
        slot = [st] + path + [st + 2] * (nq - len(path)) + [st + 1]
        pre = torch.randint(lo_txt, vocab_hi, (3,), generator=g).tolist()
        post = torch.randint(lo_txt, vocab_hi, (n_prompt - len(slot) - 3,), generator=g).tolist()
        prompt = pre + slot + post
        for side, frac in (("chosen", 1.0), ("rejected", 0.8 - 0.1 * (i % 3))):
            n_resp = max(4, int((text_len - n_prompt) * frac))
            resp = torch.randint(lo_txt, vocab_hi, (n_resp,), generator=g).tolist()
            ids = prompt + resp
            k = text_len - len(ids)
            batch[f"{side}_input_ids"].append(ids + [pad] * k)
            batch[f"{side}_attention_mask"].append([1] * len(ids) + [0] * k)
            batch[f"{side}_labels"].append([-100] * len(prompt) + resp + [-100] * k)
    out = {k: torch.tensor(v, dtype=torch.long) for k, v in batch.items()}
    out["img_input_dict"] = dict(pixel_values=torch.randn(pairs, 3, s, s, generator=g))
    out["img_path"] = [f"synthetic/{seed}/{i}.png" for i in range(pairs)]
    return out
