"""Pins the CPU oracle (oracle/llava_dpo_oracle.py) against golden vectors produced from the REFERENCE's own
functions (oracle/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import llava_dpo_oracle as O
from tests.golden_util import GOLDEN, load_case, t

CASES = ["llava_tiny", "llava_hipsmall"]


def close(a, b, tol=3e-5):
    """max-abs error relative to the tensor's scale (fp32 accumulation-order noise only)."""
    return float((a - b).abs().max()) <= tol * (float(b.abs().max()) + 1e-12)
LOSS_TYPES = ["sigmoid", "hinge", "ipo", "kto_pair", "ddpo"]


@pytest.fixture(scope="module")
def ka():
    return np.load(os.path.join(GOLDEN, "known_answers.npz"))


def test_collator_known_answer(ka):
    rows = json.loads(bytes(ka["collator_rows_json"]).decode())
    got = O.collate(rows)
    for k in ka.files:
        if k.startswith("collator.") :
            name = k[len("collator."):]
            exp = torch.from_numpy(ka[k])
            assert torch.equal(got[name].to(exp.dtype), exp) if exp.dtype != torch.float32 else torch.allclose(got[name].float(), exp), name
    assert got["img_path"] == ["a.jpg", "b.jpg"]
    # left-padded prompt, right-padded answers
    assert got["prompt_input_ids"].tolist() == [[1, 2, 3], [0, 1, 2]]
    assert got["chosen_labels"].tolist() == [[-100, -100, -100, 4, 5], [-100, -100, 4, -100, -100]]


def test_dpo_loss_known_answers(ka):
    pc, pr, rc, rr = (torch.from_numpy(ka[f"kl.{n}"]) for n in ("pc", "pr", "rc", "rr"))
    n = 0
    for lt in LOSS_TYPES:
        for ls in (0.0, 0.2):
            for rf in (False, True):
                for beta in (0.1, 0.5):
                    key = f"kl.{lt}.ls{ls}.rf{int(rf)}.b{beta}"
                    l, c, r = O.dpo_loss(pc, pr, rc, rr, beta, ls, lt, rf)
                    assert torch.allclose(l, t(ka, key + ".losses"), rtol=1e-6, atol=1e-6), key
                    assert torch.allclose(c, t(ka, key + ".cr"), rtol=1e-6, atol=1e-6)
                    assert torch.allclose(r, t(ka, key + ".rr"), rtol=1e-6, atol=1e-6)
                    n += 1
    assert n == 40
    l, _, _ = O.dpo_loss(pc, pr, pc, pr)
    assert torch.allclose(l, t(ka, "kl.ln2")) and abs(float(l[0]) - 0.6931471824645996) < 1e-7
    with pytest.raises(ValueError):
        O.dpo_loss(pc, pr, rc, rr, loss_type="nope")


def test_get_batch_logps_known_answers(ka):
    logits, labels = t(ka, "lp.logits"), t(ka, "lp.labels")
    assert torch.allclose(O.get_batch_logps(logits, labels), t(ka, "lp.sum"), rtol=1e-6, atol=1e-5)
    assert torch.allclose(O.get_batch_logps(logits, labels, average_log_prob=True), t(ka, "lp.avg"), rtol=1e-6, atol=1e-5)
    assert torch.allclose(O.get_batch_logps(logits, labels, mask_shared_tokens=True), t(ka, "lp.ddpo"), rtol=1e-6, atol=1e-5)
    with pytest.raises(ValueError):
        O.get_batch_logps(logits[:, :-1], labels)
    # SURVEY Appendix A.2 known answer for the DDPO index sets
    sh = labels[:, 1:].clone()
    sh[sh == -100] = 0
    c, r = O.get_diff_ids(sh[0].tolist(), sh[2].tolist(), 3)
    assert c == ka["lp.ddpo_c0"].tolist() and r == ka["lp.ddpo_r0"].tolist()
    assert c == [5, 6] or set([5, 6]).issubset(c)
    assert set(r) >= {5, 6, 7}


def test_merge_known_answer(ka):
    fe, fm, fl, pos, imap = O.merge_input_ids_with_image_features(
        t(ka, "mg.feats"), t(ka, "mg.emb"), t(ka, "mg.ids"), t(ka, "mg.am"), t(ka, "mg.lab"),
        image_token_index=50, pad_token_id=99)
    assert torch.equal(fe, t(ka, "mg.out_emb"))
    assert torch.equal(fm, t(ka, "mg.out_mask"))
    assert torch.equal(fl, t(ka, "mg.out_labels"))
    assert torch.equal(pos, t(ka, "mg.out_pos"))
    assert torch.equal(imap, t(ka, "mg.out_map"))
    assert fm[0].tolist() == [1, 1, 1, 1, 1, 1, 1, 1, 0, 0]
    assert pos[0].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 1, 1]
    with pytest.raises(ValueError):
        O.merge_input_ids_with_image_features(t(ka, "mg.feats")[:1], t(ka, "mg.emb"), t(ka, "mg.ids"), t(ka, "mg.am"),
                                              t(ka, "mg.lab"), 50, 99)


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_reference_composite(case):
    z, cfg, W, W_ref, batch, rows = load_case(case)
    # collator + concatenation are integer-exact
    got = O.collate(rows)
    for k in ("chosen_input_ids", "chosen_attention_mask", "chosen_labels", "rejected_input_ids",
              "rejected_labels", "prompt_input_ids", "prompt_attention_mask"):
        assert torch.equal(got[k], batch[k]), k
    cb = O.concatenated_inputs(batch)
    for k in ("concatenated_input_ids", "concatenated_attention_mask", "concatenated_labels"):
        assert torch.equal(cb[k], t(z, "cat." + k)), k
    assert cb["concatenated_img_input_dict"]["pixel_values"].shape[0] == 2 * batch["chosen_input_ids"].shape[0]
    B = batch["chosen_input_ids"].shape[0]
    with torch.no_grad():
        logits, labels, aux = O.llava_forward(W, cfg, cb["concatenated_input_ids"], cb["concatenated_attention_mask"],
                                              cb["concatenated_labels"],
                                              cb["concatenated_img_input_dict"]["pixel_values"])
    assert close(aux["vit_feat"], t(z, "vit_feat"))
    assert close(aux["image_features"][:B], t(z, "image_features"))
    assert close(aux["merged"], t(z, "merged_embeds"))
    assert torch.equal(aux["mask"], t(z, "merged_mask"))
    assert torch.equal(labels, t(z, "merged_labels"))
    assert torch.equal(aux["pos"], t(z, "merged_pos"))
    assert torch.equal(aux["img_map"], t(z, "image_position_map"))
    valid = aux["mask"].bool()
    assert close(aux["hidden"][valid], t(z, "hidden_last")[valid])
    assert close(logits[valid], t(z, "logits")[valid])
    lp = O.get_batch_logps(logits, labels)
    assert torch.allclose(lp, t(z, "policy_logps"), rtol=1e-5, atol=2e-4)
    assert torch.allclose(O.get_batch_logps(logits, labels, average_log_prob=True), t(z, "policy_logps_avg"), rtol=1e-5, atol=1e-5)
    assert torch.allclose(O.get_batch_logps(logits, labels, mask_shared_tokens=True), t(z, "policy_logps_ddpo"), rtol=1e-5, atol=2e-4)
    m = O.ddpo_shared_mask(labels)
    for b in range(B):
        assert torch.where(m[b])[0].tolist() == z[f"ddpo_chosen_ids_{b}"].tolist()
        assert torch.where(m[B + b])[0].tolist() == z[f"ddpo_rejected_ids_{b}"].tolist()


@pytest.mark.parametrize("case", CASES)
def test_losses_all_types(case):
    z, cfg, W, W_ref, batch, rows = load_case(case)
    for lt in LOSS_TYPES:
        with torch.no_grad():
            pc, pr, _, _ = O.concatenated_forward(W, cfg, batch, lt)
            rc, rr, _, _ = O.concatenated_forward(W_ref, cfg, batch, lt)
        key = "ref_logps_ddpo" if lt == "ddpo" else "ref_logps"
        assert torch.allclose(torch.cat([rc, rr]), t(z, key), rtol=1e-5, atol=2e-4)
        losses, cr, rrw = O.dpo_loss(pc, pr, rc, rr, cfg["beta"], 0.0, lt)
        assert torch.allclose(losses, t(z, f"loss_{lt}"), rtol=1e-3, atol=2e-5), lt
        assert torch.allclose(cr, t(z, f"chosen_rewards_{lt}"), rtol=1e-3, atol=2e-5)
        assert torch.allclose(rrw, t(z, f"rejected_rewards_{lt}"), rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("case", CASES)
def test_train_step_grads_and_adamw(case):
    z, cfg, W, W_ref, batch, rows = load_case(case)
    W0 = W
    W = {k: v.clone() for k, v in W.items()}
    state = {}
    loss, metrics, grads, total = O.dpo_train_step(W, W_ref, cfg, batch, cfg["optim"], state, cfg["beta"])
    assert abs(float(loss) - float(z["loss_mean_sigmoid"])) < 2e-6 + 1e-5 * abs(float(loss))
    assert abs(float(total) - float(z["grad_norm"])) < 1e-4 * float(z["grad_norm"])
    assert abs(float(total) - float(z["clip_total_norm"])) < 1e-4 * float(z["grad_norm"])
    coef = min(1.0, cfg["optim"]["max_grad_norm"] / (float(total) + 1e-6))
    ng = 0
    for k in z.files:
        if k.startswith("grad."):
            name = k[5:]
            exp = t(z, k) * coef          # oracle returns clipped grads
            g = grads[name]
            scale = float(exp.abs().max()) + 1e-12
            assert float((g - exp).abs().max()) < 2e-4 * scale + 1e-7, name
            ng += 1
    assert ng == len(grads)
    na = 0
    for k in z.files:
        if k.startswith("after_step.") :
            name = k[len("after_step."):]
            # Adam's first step is +-lr wherever |g| >> eps, so elements whose gradient is at the 1e-7 noise
            # floor (below eps=1e-6) legitimately differ by O(lr*noise/eps): compare the UPDATE in L2
            upd = t(z, k) - W0[name]
            assert float(((W[name] - W0[name]) - upd).norm()) <= 1e-3 * float(upd.norm()) + 1e-9, name
            assert float((W[name] - t(z, k)).abs().max()) <= 0.02 * cfg["optim"]["lr"], name
            na += 1
    assert na > 0
    sq = sum(float((W[n].double() ** 2).sum()) for n in O.trainable_names(W))
    assert abs(sq - float(z["after_step_sqnorm"])) < 1e-6 * sq
    exp_keys = {"rewards/chosen", "rewards/rejected", "rewards/accuracies", "rewards/margins", "logps/rejected",
                "logps/chosen", "logits/rejected", "logits/chosen"}
    assert set(metrics) == exp_keys


def test_bf16_emulation_close_to_fp32():
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    with torch.no_grad():
        l32, _ = O.compute_loss(W, W_ref, cfg, batch, cfg["beta"])
        l16, _ = O.compute_loss(W, W_ref, cfg, batch, cfg["beta"], emulate_bf16=True)
    assert abs(float(l32) - float(l16)) < 0.05


# ---------------------------------------------------------------------------------------------------- LoRA restatement
def test_lora_forward_equals_merged_base_forward():
    """peft identity: base(x) + s*B(A(x)) == (W + s*B A) x.  Pins the LoRA restatement to the golden-pinned base path."""
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    lora = O.random_lora(cfg, r=8, alpha=16, seed=3, b_std=0.05)
    with torch.no_grad():
        l_a, m_a = O.compute_loss(W, W, cfg, batch, cfg["beta"], lora=lora)
        l_m, m_m = O.compute_loss(O.lora_merged_weights(W, lora, cfg), W, cfg, batch, cfg["beta"])
        l_0, _ = O.compute_loss(W, W, cfg, batch, cfg["beta"])
    assert abs(float(l_a) - float(l_m)) < 1e-5 * max(1.0, abs(float(l_m)))
    assert abs(float(l_a) - float(l_0)) > 1e-4          # the adapters do change the loss
    for k in m_a:
        assert abs(float(m_a[k]) - float(m_m[k])) < 2e-4 * max(1.0, abs(float(m_m[k]))), k
    # B = 0 (peft init): policy == reference, loss = ln 2
    l_i, _ = O.compute_loss(W, W, cfg, batch, cfg["beta"], lora=O.random_lora(cfg, 8, 16, seed=1))
    assert abs(float(l_i) - 0.6931472) < 1e-6


def test_dropout_mask_restatement():
    m1 = O.dropout_mask(1234, 1 << 16, 0.05)
    m2 = O.dropout_mask(1234, 1 << 16, 0.05)
    m3 = O.dropout_mask(1235, 1 << 16, 0.05)
    assert torch.equal(m1, m2) and not torch.equal(m1, m3)
    assert abs(float(m1.float().mean()) - 0.95) < 5e-3
    assert abs(float((m1 & m3).float().mean()) - 0.95 * 0.95) < 8e-3           # independent streams
    assert int(O.dropout_mask(7, 64, 0.0).sum()) == 64
    # prefix property: the mask of element i does not depend on n
    assert torch.equal(O.dropout_mask(9, 4096, 0.3)[:512], O.dropout_mask(9, 512, 0.3))


def test_hashed_weights_are_machine_independent_and_match_the_product_generator():
    """oracle.hashed_normal (numpy uint32) == vlrlhf.utils.synthetic.hashed_normal (torch int64, any device), bit for bit;
    known values pin the function itself; the synthetic batch builders of both sides agree."""
    from oracle import llava_dpo_oracle as O
    from vlrlhf.utils import synthetic as S
    for n, seed, name in ((1, 0, "a"), (1000003, 5, "language_model.model.layers.3.mlp.gate_proj.weight"), ((1 << 20) + 7, 1, "x")):
        a, b = O.hashed_normal(n, seed, name), S.hashed_normal(n, seed, name, "cpu")
        assert torch.equal(a, b)
    x = O.hashed_normal(4, 5, "language_model.x")
    assert [round(v, 4) for v in x.tolist()] == [-1.3347, -0.5576, -0.144, 0.4453]
    big = O.hashed_normal(1 << 20, 0, "stats")
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1.0) < 5e-3
    W = O.HashedWeights(dict(O.LLAVA_1_5_7B, layers=1), seed=0, delta=1e-3)
    g = W["language_model.model.layers.0.input_layernorm.weight"]
    assert g.shape == (4096,) and abs(float(g.mean()) - 1.0) < 5e-3 and torch.equal(g, g.bfloat16().float())
    bo = O.synthetic_batch(2, 48, 32000, 32000, 28, 11, ragged=True)
    bp = S.synthetic_batch(2, 48, 32000, 32000, 28, 11, ragged=True)
    for k, v in bo.items():
        if isinstance(v, torch.Tensor):
            assert torch.equal(v, bp[k]), k
    assert torch.equal(bo["img_input_dict"]["pixel_values"], bp["img_input_dict"]["pixel_values"])


def test_llavanext_oracle_matches_reference_golden():
    """the LLaVA-Next restatement (anyres pack, variable-length merge, grouped-query decoder) against
    tests/golden/llavanext_small.npz = the reference's own merge / get_batch_logps / dpo_loss composed with HF CLIP, pack_image_features
    and MistralForCausalLM (oracle/make_golden_llavanext.py)."""
    from oracle import llava_dpo_oracle as O
    from tests.golden_util import load_case, t
    z, cfg, W, W_ref, batch, rows = load_case("llavanext_small")
    # merge known answers (right / left padding, two images in one row)
    for tag in ("right", "left", "nopad_two_images"):
        g = lambda k: t(z, f"mg.{tag}.{k}")   # noqa: E731
        emb, m, pos, lab, imap = O.llavanext_merge(g("feats"), g("fl"), g("emb"), g("ids"), g("am"), g("lab"), 180, "left")
        assert torch.equal(m, g("out_mask")) and torch.equal(pos, g("out_pos")) and torch.equal(lab, g("out_labels")), tag
        assert torch.equal(imap, g("out_map")) and torch.equal(emb, g("out_emb")), tag
    assert [O.anyres_num_patches(s, cfg["image_grid_pinpoints"], cfg["image_size"]) for s in cfg["image_sizes"]] == z["num_patches"].tolist()[:2]
    cb = O.concatenated_inputs(batch)
    assert torch.equal(cb["concatenated_img_input_dict"]["image_sizes"], torch.cat([batch["img_input_dict"]["image_sizes"]] * 2))
    with torch.no_grad():
        logits, labels, aux = O.llavanext_forward(W, cfg, cb["concatenated_input_ids"], cb["concatenated_attention_mask"], cb["concatenated_labels"],
                                                  cb["concatenated_img_input_dict"]["pixel_values"], cb["concatenated_img_input_dict"]["image_sizes"])
    B = batch["chosen_input_ids"].shape[0]
    assert aux["feature_lens"].tolist() == z["feature_lens"].tolist()
    assert torch.equal(labels, t(z, "merged_labels")) and torch.equal(aux["mask"], t(z, "merged_mask"))
    assert torch.equal(aux["pos"], t(z, "merged_pos")) and torch.equal(aux["img_map"], t(z, "image_position_map"))
    F = int(aux["feature_lens"][:B].sum())

    def close(a, b, tol, what):
        e = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
        assert e < tol, f"{what}: {e:.2e}"
    close(aux["vit_feat"], t(z, "vit_feat"), 3e-5, "vit")
    close(aux["packed"][:F], t(z, "packed_features"), 3e-5, "packed features")
    close(aux["merged"], t(z, "merged_embeds"), 3e-5, "merged")
    close(logits, t(z, "logits"), 5e-5, "logits")
    lp = O.get_batch_logps(logits, labels)
    lpd = O.get_batch_logps(logits, labels, mask_shared_tokens=True)
    close(lp, t(z, "policy_logps"), 2e-5, "logps")
    close(lpd, t(z, "policy_logps_ddpo"), 2e-5, "ddpo logps")
    # the DDPO training loss and its gradients (GQA backward through autograd of the restatement)
    leaves = {k: v.clone().requires_grad_(True) for k, v in W.items() if not k.startswith("vision_tower.")}
    Wp = dict(W)
    Wp.update(leaves)
    loss, _ = O.compute_loss(Wp, W_ref, cfg, batch, cfg["beta"], loss_type="ddpo")
    assert abs(float(loss) - float(z["loss_mean_ddpo"])) < 2e-5
    loss.backward()
    n = 0
    for k in z.files:
        if k.startswith("grad."):
            close(leaves[k[5:]].grad, t(z, k), 2e-3, k)
            n += 1
    assert n >= 10


# ------------------------------------------------------------------------------------------------------------ Qwen-VL
def test_qwenvl_oracle_matches_reference_golden():
    """oracle/qwenvl_oracle.py against tests/golden/qwenvl_small.npz = outputs of the reference's own QWenLMHeadModel /
    VisionTransformer / Resampler forward (oracle/make_golden_qwenvl.py): vision features, logits, image_position_map, the image
    paths decoded from the ids, log-probs / losses of three loss types, and autograd gradients."""
    from oracle import qwenvl_oracle as Q
    z, cfg, W, W_ref, batch, _ = load_case("qwenvl_small")
    px = batch["img_input_dict"]["pixel_values"]
    paths = json.loads(bytes(z["paths_json"]).decode())
    vf = Q.qwen_visual(px, W_ref, cfg["visual"])
    assert torch.allclose(vf, t(z, "visual_features"), rtol=2e-4, atol=2e-4), float((vf - t(z, "visual_features")).abs().max())
    assert torch.allclose(Q.qwen_visual(px, W, cfg["visual"]), t(z, "policy_visual_features"), rtol=2e-4, atol=2e-4)   # the policy's resampler differs
    cb = O.concatenated_inputs({k: v for k, v in batch.items() if k != "img_input_dict"}, padding_value=cfg["pad_token_id"])
    ids, am, lab = cb["concatenated_input_ids"], cb["concatenated_attention_mask"], cb["concatenated_labels"]
    assert Q.decode_image_paths(ids, cfg["image_start_id"]) == paths + paths
    Wg = {k: (v.clone().requires_grad_(True) if any(k == g[5:] or k == g[11:] for g in z.files if g.startswith("grad")) else v) for k, v in W.items()}
    logits, img_map, _ = Q.qwenvl_forward(Wg, cfg, ids, am, torch.cat([px, px], 0))
    ref_logits = t(z, "logits")
    assert torch.equal(img_map, t(z, "image_position_map"))
    assert float((logits - ref_logits).abs().max()) < 2e-3 * float(ref_logits.abs().max())
    for lt in ("sigmoid", "ipo", "ddpo"):
        pc, pr, _, _ = Q.concatenated_forward(W, cfg, dict(batch, pixel_values=px), lt)
        rc, rr, _, _ = Q.concatenated_forward(W_ref, cfg, dict(batch, pixel_values=px), lt)
        assert torch.allclose(torch.cat([pc, pr]), t(z, f"{lt}.logps"), rtol=1e-4, atol=2e-3)
        assert torch.allclose(torch.cat([rc, rr]), t(z, f"{lt}.ref_logps"), rtol=1e-4, atol=2e-3)
        losses, _, _ = O.dpo_loss(pc, pr, rc, rr, cfg["beta"], 0.0, lt, False)
        assert torch.allclose(losses, t(z, f"{lt}.losses"), rtol=2e-3, atol=2e-4), (lt, losses, t(z, f"{lt}.losses"))
    # gradients of the sigmoid loss through the oracle's own forward
    lp = O.get_batch_logps(logits, lab)
    n = batch["chosen_input_ids"].shape[0]
    rl = t(z, "sigmoid.ref_logps")
    losses, _, _ = O.dpo_loss(lp[:n], lp[n:], rl[:n], rl[n:], cfg["beta"], 0.0, "sigmoid", False)
    losses.mean().backward()
    checked = 0
    for k in z.files:
        if k.startswith("grad."):
            g, ref = Wg[k[5:]].grad, t(z, k)
            assert float((g - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-7, k
            checked += 1
        elif k.startswith("grad_probe."):
            g, ref = Wg[k[11:]].grad, t(z, k)
            assert float((g.reshape(-1)[::17] - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-7, k
            assert abs(float(g.norm()) - float(z["grad_norm." + k[11:]])) <= 2e-3 * float(z["grad_norm." + k[11:]]), k
            checked += 1
    assert checked == 20                  # ten language-model tensors + ten resampler tensors (trainable in a full fine-tune)


# ------------------------------------------------------------------------------------------------------------ InternLM-XComposer2
def test_internlm_oracle_matches_reference_golden():
    """oracle/internlm_oracle.py against tests/golden/internlmxc2_small.npz = outputs of the reference's own InternLMXC2ForRL forward
    (LLaVA-style merge, InternLM2 decoder with the fused grouped-query wqkv and PLoRA on the image rows) and autograd."""
    from oracle import internlm_oracle as IL
    z, cfg, W, W_ref, batch, _ = load_case("internlmxc2_small")
    cb = O.concatenated_inputs(batch, padding_value=cfg["model_pad_token_id"])
    ids, am, lab = cb["concatenated_input_ids"], cb["concatenated_attention_mask"], cb["concatenated_labels"]
    px2 = cb["concatenated_img_input_dict"]["pixel_values"]
    names = {k.split(".", 1)[1] for k in z.files if k.startswith(("grad.", "grad_probe."))}
    Wg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in W.items()}
    logits, labels, aux = IL.internlm_forward(Wg, cfg, ids, am, lab, px2)
    assert torch.equal(labels, t(z, "merged_labels")) and torch.equal(aux["img_map"], t(z, "image_position_map"))
    n = batch["chosen_input_ids"].shape[0]
    assert torch.allclose(aux["image_features"][:n], t(z, "image_features"), rtol=2e-4, atol=2e-4)
    ref_logits = t(z, "logits")
    assert float((logits - ref_logits).abs().max()) < 2e-3 * float(ref_logits.abs().max())
    for lt in ("sigmoid", "ddpo"):
        pc, pr, _, _ = IL.concatenated_forward(W, cfg, batch, lt)
        rc, rr, _, _ = IL.concatenated_forward(W_ref, cfg, batch, lt)
        assert torch.allclose(torch.cat([pc, pr]), t(z, f"{lt}.logps"), rtol=1e-4, atol=2e-3)
        assert torch.allclose(torch.cat([rc, rr]), t(z, f"{lt}.ref_logps"), rtol=1e-4, atol=2e-3)
        losses, _, _ = O.dpo_loss(pc, pr, rc, rr, cfg["beta"], 0.0, lt, False)
        assert torch.allclose(losses, t(z, f"{lt}.losses"), rtol=2e-3, atol=2e-4)
    # the fixture is discriminating for DDPO: chosen / rejected share >= 3-token spans, the masked log-probs lose them
    assert float((t(z, "ddpo.logps") - t(z, "sigmoid.logps")).abs().min()) > 5.0 and abs(float(z["ddpo.loss"]) - float(z["sigmoid.loss"])) > 1e-4
    lp = O.get_batch_logps(logits, labels)
    rl = t(z, "sigmoid.ref_logps")
    losses, _, _ = O.dpo_loss(lp[:n], lp[n:], rl[:n], rl[n:], cfg["beta"], 0.0, "sigmoid", False)
    losses.mean().backward()
    checked = 0
    for k in z.files:
        if k.startswith("grad."):
            g, ref = Wg[k[5:]].grad, t(z, k)
            assert float((g - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-7, k
            checked += 1
        elif k.startswith("grad_probe."):
            g, ref = Wg[k[11:]].grad, t(z, k)
            assert float((g.reshape(-1)[::17] - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-7, k
            checked += 1
    assert checked == 13


def test_internlm_gqa_row_order_matches_reference_rearrange():
    """the load-time row permutation of the fused grouped-query wqkv (oracle.qkv_row_order = engine ParamLayout.row_perm) against the
    reference's split of the projection output (modeling_internlm2.py:320-330: "b q (h gs d) -> b q h gs d", gs = 2 + group; q = first
    `group` slots of every K/V head, k = slot -2, v = slot -1), restated here with plain reshapes, for 8 query heads over 2 K/V heads."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "vl-rlhf_amd"))
    from oracle import internlm_oracle as IL
    from vlrlhf.engine import LoraLayout, ParamLayout
    nh, nkv, hd, H = 8, 2, 16, 128
    g = nh // nkv
    y = torch.randn(3, 5, (nh + 2 * nkv) * hd)
    v5 = y.view(3, 5, nkv, g + 2, hd)
    q_ref, k_ref, v_ref = v5[..., :g, :].reshape(3, 5, nh * hd), v5[..., -2, :].reshape(3, 5, nkv * hd), v5[..., -1, :].reshape(3, 5, nkv * hd)
    perm = IL.qkv_row_order(nh, nkv, hd)
    yp = y[..., perm]
    assert torch.equal(yp[..., : nh * hd], q_ref) and torch.equal(yp[..., nh * hd: (nh + nkv) * hd], k_ref) and torch.equal(yp[..., (nh + nkv) * hd:], v_ref)
    cfg = dict(family="internlm_xc2", hidden=H, inter=64, layers=1, heads=nh, kv_heads=nkv, vocab=32, vit_hidden=16)
    lay = ParamLayout(cfg)
    assert torch.equal(lay.row_perm["l0.wqkv"], perm) and torch.equal(lay.row_perm["l0.pb_qkv"], perm)
    assert torch.equal(LoraLayout(cfg, 8).row_perm["b_qkv"], perm)
    assert lay.n_opt == lay.offset["proj.w2"] < lay.numel                 # the frozen projector sits behind the optimizer's range
