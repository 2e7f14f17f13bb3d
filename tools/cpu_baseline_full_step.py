#!/usr/bin/env python3
"""ONE full DPO optimizer step of LLaVA-1.5-7B on the host's cores with the fp32 oracle (oracle/llava_dpo_oracle.py), at BASELINE.json
configs[0] (4 pairs, 336x336 images, max_length 256 -> 8 sequences x 831 positions): CLIP ViT-L/14-336 (23 of 24 layers, 4 distinct images),
32 decoder layers, lm-head over all positions, policy forward + backward, reference forward, clip, AdamW on 6.76 G parameters.
bench.py's cpu_baseline leg times a bounded SAMPLE of this step (one layer x 32, ...) in ~20 s; this script is what that sample is checked
against.  ~250 GB of host memory, ~10 min on 128 threads.  Writes gpurun_out/r04_cpu_baseline_full_step.json.

    python tools/cpu_baseline_full_step.py [layers]        (layers < 32: a shorter run for trying the script)
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import llava_dpo_oracle as O       # the checker / baseline, never the product


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 2 ** 20
    return 0.0


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    need = 30 + 7.7 * layers
    avail = mem_available_gb()
    if avail < need:
        print(json.dumps(dict(error=f"needs ~{need:.0f} GB of host memory, {avail:.0f} GB available")))
        return 1
    torch.manual_seed(0)
    nthr = torch.get_num_threads()
    cfg = dict(vit_hidden=1024, vit_mlp=4096, vit_layers=24, vit_heads=16, image_size=336, patch_size=14, hidden=4096, inter=11008,
               layers=layers, heads=32, vocab=32064, image_token=32000, model_pad_token_id=32001, beta=0.1, vit_feature_layer=-2)
    optim = dict(lr=2e-8, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_grad_norm=1.0)
    t0 = time.time()
    W = {}

    class Fast:      # torch.randn of 6.7 G values is single-threaded minutes: big matrices are filled by normal_() on views instead
        pass
    g = torch.Generator().manual_seed(0)
    Wsmall = O.random_weights(dict(cfg, layers=0), seed=0, std=0.02)          # ViT + projector + embeddings + lm_head, real shapes
    W.update(Wsmall)
    lp = "language_model.model."
    H, I = cfg["hidden"], cfg["inter"]
    for i in range(layers):
        p = f"{lp}layers.{i}."
        for nm, shape in (("self_attn.q_proj", (H, H)), ("self_attn.k_proj", (H, H)), ("self_attn.v_proj", (H, H)), ("self_attn.o_proj", (H, H)),
                          ("mlp.gate_proj", (I, H)), ("mlp.up_proj", (I, H)), ("mlp.down_proj", (H, I))):
            W[p + nm + ".weight"] = torch.empty(shape).normal_(0.0, 0.02, generator=g)
        W[p + "input_layernorm.weight"] = torch.ones(H)
        W[p + "post_attention_layernorm.weight"] = torch.ones(H)
    t_init = time.time() - t0
    batch = O.synthetic_batch(4, 256, cfg["image_token"], 32000, cfg["image_size"], seed=13)
    state = {}
    # the reference model aliases the policy's tensors here (its forward runs under no_grad before the update): same arithmetic, 27 GB less
    names = O.trainable_names(W)
    t1 = time.time()
    leaves = {k: W[k].detach().clone().requires_grad_(True) for k in names}
    Wp = dict(W)
    Wp.update(leaves)
    t_clone = time.time() - t1
    t1 = time.time()
    loss, metrics = O.compute_loss(Wp, W, cfg, batch, cfg["beta"], "sigmoid")
    t_fwd = time.time() - t1                         # policy forward (graph kept) + reference forward
    t1 = time.time()
    loss.backward()
    t_bwd = time.time() - t1
    grads = {k: leaves[k].grad for k in names if leaves[k].grad is not None}
    t1 = time.time()
    total = O.clip_grad_norm_(grads, optim["max_grad_norm"])
    t_clip = time.time() - t1
    t1 = time.time()
    with torch.no_grad():
        O.adamw_step(W, grads, state, optim["lr"], optim["beta1"], optim["beta2"], optim["eps"], optim["weight_decay"])
    t_adam = time.time() - t1
    step_s = t_fwd + t_bwd + t_clip + t_adam
    n_params = sum(W[k].numel() for k in names)
    out = dict(what="one full fp32 DPO step of the CPU oracle, BASELINE.json configs[0] (4 pairs, T=256, S=831), LLaVA-1.5-7B shapes",
               layers=layers, threads=nthr, trainable_params=n_params, loss=float(loss), grad_norm=float(total),
               seconds=dict(weights_init=round(t_init, 1), leaves_clone=round(t_clone, 1), forward_policy_and_reference=round(t_fwd, 1),
                            backward=round(t_bwd, 1), clip=round(t_clip, 1), adamw=round(t_adam, 1), step=round(step_s, 1)),
               pairs_per_s=round(4.0 / step_s, 6), host_mem_available_gb_before=round(avail, 0))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_cpu_baseline_full_step.json" if layers == 32 else f"cpu_baseline_{layers}_layers.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
