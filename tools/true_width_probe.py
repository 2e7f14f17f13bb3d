#!/usr/bin/env python3
"""per-sequence log-probs of the true-width one-layer LLaVA-Next case (tests/test_hip_true_width.py) on the HIP path, the oracle's model of
the path's rounding and the fp32 oracle - where does a loss difference come from?    python tools/true_width_probe.py [text_len]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from oracle import llava_dpo_oracle as O  # noqa: E402  (checker)
import tests.test_hip_true_width as T  # noqa: E402
from tests.golden_util import load_case  # noqa: E402


def main():
    T_len = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
    from vlrlhf.models.LlavaNext import LlavaNextDPOTrainer, LlavaNextForRL
    from vlrlhf.utils.synthetic import synthetic_batch_anyres
    _, cfg0, W0, _, _, _ = load_case("llavanext_small")
    cfg = dict(cfg0, hidden=4096, inter=14336, heads=32, kv_heads=8, layers=1, vocab=32064, image_token=32000, model_pad_token_id=32001)
    W, lay = T.wide_weights(cfg, W0, seed=21)
    W_ref = T.perturbed(W, lay, seed=22)
    batch = synthetic_batch_anyres(2, T_len, cfg["image_token"], 32000, cfg["image_size"], seed=23, image_hw=(40, 75),
                                   grid_pinpoints=cfg["image_grid_pinpoints"], ragged=True)
    model = LlavaNextForRL.from_state_dict(cfg, W)
    ref = model.create_reference_model()
    ref.weights.load_state_dict(W_ref)
    tr = LlavaNextDPOTrainer(model, ref, 0.1, 0, "sigmoid", T._args(), None, -100, 0)
    with torch.no_grad():
        pc, pr, _, _ = tr.concatenated_forward(model, batch)
        rc, rr, _, _ = tr.concatenated_forward(ref, batch)
    hip = torch.cat([pc, pr, rc, rr]).cpu()
    rows = {}
    for tag, emu in (("emulation", O.HIP_ROUNDING), ("fp32", False), ("all bf16", True)):
        with torch.no_grad():
            a, b, _, _ = O.concatenated_forward(W, cfg, batch, "sigmoid", emu)
            c, d, _, _ = O.concatenated_forward(W_ref, cfg, batch, "sigmoid", emu)
        rows[tag] = torch.cat([a, b, c, d])
    n_tok = (batch["chosen_labels"] != -100).sum(-1).tolist() + (batch["rejected_labels"] != -100).sum(-1).tolist()
    print("response tokens per sequence (chosen, rejected):", n_tok)
    print("sequence order: policy chosen x2, policy rejected x2, reference chosen x2, reference rejected x2")
    print("hip      ", [f"{x:.3f}" for x in hip.tolist()])
    for k, v in rows.items():
        print(f"{k:9s}", [f"{x:.3f}" for x in v.tolist()], " max |hip - this|", f"{float((hip - v).abs().max()):.4f}")

    def loss_of(v):
        pc_, pr_, rc_, rr_ = v[0:2], v[2:4], v[4:6], v[6:8]
        return float(O.dpo_loss(pc_, pr_, rc_, rr_, 0.1)[0].mean())
    print("loss hip", loss_of(hip), {k: loss_of(v) for k, v in rows.items()})


if __name__ == "__main__":
    main()
