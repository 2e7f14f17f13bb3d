#!/usr/bin/env python3
"""Benchmark of the north-star metric: preference-pairs/sec of one full LLaVA-1.5-7B DPO optimizer step on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1 without a launcher: re-executes itself under
                                                          torch.distributed.run, one rank per GPU, backend nccl = RCCL)

One "step" = policy forward + backward, reference forward, DPO loss, gradient all-reduce (N>1), clip + AdamW, on a
synthetic batch already resident in HBM: per rank 4 pairs, 336x336 image, 1024 text tokens each (BASELINE.json
configs[1]; S = 1599 decoder positions), random N(0,0.02) bf16 weights, policy != reference.  Prints ONE JSON line.

roofline: the dominant kernel is the 8-phase 256x256x64 bf16 MFMA GEMM (gemm256p_kernel, all three layouts: 486 launches
and ~79 % of the step); `achieved` = its algorithmic FLOPs (2*M*N*K per launch) / its summed launch durations, measured
with HIP events on the launch stream inside the timed region (in-library profiler, kernel id 5); `per_kernel` lists the
whole vlr_gemm_bf16 calls per layout (incl. peeled rows / split-K reduces) and the attention kernels.  `step_frac` = pairs/s x 174.87 TFLOP (reference forward inside the step) / 2516.6 TF/s.
cpu_baseline: the fp32 CPU oracle (oracle/llava_dpo_oracle.py, a port of the reference algorithm) timed on this host's
cores on a bounded sample of the configs[0] step - one decoder layer fwd+bwd (+ reference fwd), the lm-head + log-prob
fwd+bwd on all positions, one ViT layer, AdamW on one layer's parameters - each scaled by how often the full step runs
it; `value` is the MEASURED full 32-layer step of that oracle on the same host class (profiles/r04_cpu_baseline_full_step.json)
with the live sample beside it as the check; a reported baseline, not the target.
The timed steps rotate over four resident batches with lr = 2e-8 so that the loss stays in the non-saturated regime
(the arithmetic of every kernel, AdamW included, does not depend on lr); `loss_first_step` / `loss_last_step` are printed
and must be finite.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2516.6          # 256 CU x 4096 FLOP/clk/CU x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PF dense)
PROFILE_FILE = "profiles/r06_rocprofv3_kernel_stats_bench.csv"   # rocprofv3 --kernel-trace --stats of this command; `frac` can be recomputed from it
TFLOP_PER_PAIR = {"ref_in_step": 174.87, "ref_precomputed": 131.24}


def tflop_per_pair_qwen(cfg, S, ref_in_step=True, lora=False):
    """same counting rules for Qwen-VL: decoder as below (S = T: the 256 image slots are part of the ids), frozen ViT-bigG + resampler
    once per image; under LoRA the backward is the data gradient only (1x forward instead of 2x; adapter FLOPs not counted)"""
    H, I, V, L, nh = cfg["hidden"], cfg["inter"], cfg["vocab"], cfg["layers"], cfg["heads"]
    dense = 2 * (L * (4 * H * H + 3 * H * I) + V * H)
    attn = L * 4 * H * (S + 1) / 2
    fwd = 2 * S * (dense + attn)
    v = cfg["visual"]
    W, E, T, nq = v["width"], v["output_dim"], (v["image_size"] // v["patch_size"]) ** 2, v.get("n_queries", 256)
    F = int(W * v["mlp_ratio"])
    vit = v["layers"] * (2 * T * (4 * W * W + 2 * W * F) + 4 * T * T * W) + 2 * T * (W * E + 2 * E * E) + 4 * nq * T * E + 4 * nq * E * E
    passes = (1 if ref_in_step else 0) + 1 + (1 if lora else 2)
    return (passes * fwd + vit) / 1e12


def tflop_per_pair(cfg, S, tiles_per_image=1, ref_in_step=True):
    """BASELINE.md section 3 counting rules for any decoder of this family (2 FLOP/MAC, causal attention at half, lm-head on all
    positions, backward = 2x forward, frozen ViT once per image tile, reference forward 1x when it runs inside the step);
    reproduces 174.87 / 131.24 for LLaVA-1.5-7B at S = 1599."""
    H, I, V, L = cfg["hidden"], cfg["inter"], cfg["vocab"], cfg["layers"]
    nh = cfg["heads"]
    nkv = cfg.get("kv_heads") or nh
    hd = H // nh
    Nq, Nkv = nh * hd, nkv * hd
    dense = 2 * (L * (H * (Nq + 2 * Nkv) + Nq * H + 3 * H * I) + V * H)
    attn = L * 4 * Nq * (S + 1) / 2
    fwd = 2 * S * (dense + attn)
    D, F, T = cfg["vit_hidden"], cfg["vit_mlp"], (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
    vit = (cfg["vit_layers"] + 1 + cfg.get("vit_feature_layer", -2)) * (2 * T * (4 * D * D + 2 * D * F) + 4 * T * T * D) * tiles_per_image
    proj = 2 * (T - 1) * (D * H + H * H) * tiles_per_image
    return ((4 if ref_in_step else 3) * fwd + vit + 3 * proj) / 1e12


def cpu_baseline(budget_s=30.0):
    """fp32 oracle on a bounded sample of the configs[0] step (4 pairs, T=256 -> 8 sequences x 831 positions, LLaMA-7B
    widths): every distinct piece of the step is timed once and multiplied by its count in the full step."""
    from oracle import llava_dpo_oracle as O       # checker / baseline only
    import torch.nn.functional as F
    torch.manual_seed(0)
    n_thr = torch.get_num_threads()
    H, I, nh, V = 4096, 11008, 32, 32064
    B, S = 8, 831
    cfg = dict(hidden=H, inter=I, layers=1, heads=nh, vocab=8, rms_eps=1e-5)
    p = "language_model.model.layers.0."
    W = {p + f"self_attn.{n}_proj.weight": torch.randn(H, H) * 0.02 for n in "qkvo"}
    W.update({p + "mlp.gate_proj.weight": torch.randn(I, H) * 0.02, p + "mlp.up_proj.weight": torch.randn(I, H) * 0.02,
              p + "mlp.down_proj.weight": torch.randn(H, I) * 0.02, p + "input_layernorm.weight": torch.ones(H),
              p + "post_attention_layernorm.weight": torch.ones(H), "language_model.model.norm.weight": torch.ones(H)})
    x = torch.randn(B, S, H)
    am = torch.ones(B, S, dtype=torch.long)
    pos = torch.arange(S)[None].expand(B, S)
    t_start = time.time()
    # (1) one decoder layer: policy fwd+bwd, reference fwd
    leaves = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    t0 = time.time()
    col = []
    O.llama_hidden(x, am, pos, leaves, cfg, collect=col)
    t_fwd = time.time() - t0
    col[0].square().mean().backward()
    t_layer = time.time() - t0
    del leaves, col
    # (2) lm-head + get_batch_logps on all positions (the reference materialises [2B,S,V] fp32), fwd + bwd, and once more fwd for the reference model
    wl = (torch.randn(V, H) * 0.02).requires_grad_(True)
    hid = torch.randn(B, S, H, requires_grad=True)
    labels = torch.randint(0, V, (B, S))
    labels[:, : S - 128] = -100
    t0 = time.time()
    lp = O.get_batch_logps(hid @ wl.t(), labels)
    t_head_f = time.time() - t0
    lp.sum().backward()
    t_head = time.time() - t0
    del lp, hid
    # (3) AdamW on one decoder layer's 202 M parameters (fp32 state)
    n_layer = 4 * H * H + 3 * H * I
    st = {}
    Wp = {"w": torch.zeros(n_layer)}
    t0 = time.time()
    O.adamw_step(Wp, {"w": torch.ones(n_layer)}, st, 1e-6, step=1)
    t_adam = time.time() - t0
    # (3b) clip_grad_norm_ over one layer's gradients (the full step spends 37 s of its 673 s there: profiles/r04_cpu_baseline_full_step.json)
    gr = {"w": torch.ones(n_layer)}
    t0 = time.time()
    O.clip_grad_norm_(gr, 1.0)
    t_clip = time.time() - t0
    del Wp, st, gr
    # (4) one CLIP ViT-L/14-336 layer on the 4 distinct images (the oracle dedupes like the HIP path)
    vcfg = dict(vit_hidden=1024, vit_mlp=4096, vit_layers=2, vit_heads=16, image_size=336, patch_size=14)
    Wv = O.random_weights(dict(vcfg, hidden=8, inter=8, vocab=8, layers=0, heads=1), seed=0)
    px = torch.randn(4, 3, 336, 336)
    t0 = time.time()
    with torch.no_grad():
        O.clip_vit_features(px, Wv, vcfg)
    t_vit = time.time() - t0
    n_params = 32 * n_layer + 2 * V * H
    step_s = 32 * (t_layer + t_fwd) + (t_head + t_head_f) + 23 * t_vit + (t_adam + t_clip) * n_params / n_layer
    # the sample is checked against ONE full 32-layer step of the same oracle on the GPU box's host (tools/cpu_baseline_full_step.py, offline:
    # ~11 minutes of CPU work, profiles/r04_cpu_baseline_full_step.json)
    check = None
    value, how = 4.0 / step_s, "the live sample's extrapolation (timed in this run, on this host)"
    fp = os.path.join(ROOT, "profiles", "r04_cpu_baseline_full_step.json")
    if os.path.exists(fp):
        full = json.load(open(fp))
        check = dict(full_step_s=full["seconds"]["step"], full_step_threads=full["threads"], full_step_pairs_per_s=full["pairs_per_s"],
                     sample_pairs_per_s=round(4.0 / step_s, 6), sample_extrapolation_over_full_step=round(step_s / full["seconds"]["step"], 3),
                     file="profiles/r04_cpu_baseline_full_step.json")
        # `value` stays the LIVE sample of this run on this host; the one full 32-layer step measured offline on this host
        # class stands beside it in `checked_against` (the sample is 11 - 18 % optimistic: it does not see the cache pressure of 32 layers'
        # activations)
    return dict(value=value, unit="pairs/s", cores=n_thr, kind="port", value_is=how, checked_against=check,
                sample=f"configs[0] shape (4 pairs, T=256, S=831), fp32, {time.time() - t_start:.0f} s of CPU work: one LLaMA-7B decoder layer "
                       f"fwd+bwd {t_layer:.2f} s and reference fwd {t_fwd:.2f} s (x32); lm-head + log-probs over all 8x831 positions "
                       f"fwd+bwd {t_head:.2f} s + reference fwd {t_head_f:.2f} s (x1); one ViT layer on 4 images {t_vit:.2f} s (x23); AdamW on "
                       f"one layer's {n_layer / 1e6:.0f} M parameters {t_adam:.2f} s + gradient clipping {t_clip:.2f} s (x{n_params / n_layer:.1f}); "
                       f"extrapolated full step {step_s:.0f} s")


def rccl_debug_setup():
    """before the first communicator of a multi-rank run: RCCL's INIT log of this process goes to a file of its own, so that the line can
    quote the channel counts RCCL ACTUALLY created (not the ones it was asked for).  The user's NCCL_DEBUG settings win."""
    if "NCCL_DEBUG" in os.environ or "NCCL_DEBUG_FILE" in os.environ:
        return os.environ.get("NCCL_DEBUG_FILE")
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"vlr_rccl_init_{os.getpid()}.log")
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=path)
    return path


def rccl_debug_channels(path):
    """channel counts of the communicators this process created so far, in creation order, from RCCL's INIT log ("N coll channels" of
    init.cc; else the denominators of the "Channel i/N" ring listing); [] when the log says nothing"""
    import re
    try:
        text = open(path).read() if path else ""
    except OSError:
        return []
    got = [int(m) for m in re.findall(r"(\d+) coll channels", text)]
    if not got:
        seen = []
        for m in re.finditer(r"Channel \d+/(\d+)\s*:", text):
            if not seen or seen[-1] != int(m.group(1)):
                seen.append(int(m.group(1)))
        got = seen
    return got


def bucket_probe(world, rank, make_comm, numel, dtype, device, bounds, log_path=None, iters=5):
    """all-reduce bus bandwidth of ONE gradient bucket (a decoder layer of the 7B model = 0.4 GB of bf16) in front of the timed region, on a
    communicator of its own per channel bound in `bounds` (0 = RCCL's default) - a single scaling run then says whether the bound that
    keeps the ring kernels inside the reserved CUs starves xGMI.  make_comm(bound) -> object with all_reduce_(tensor, stream) and close().
    busbw = algbw x 2 (n - 1) / n (the per-link figure of a ring)."""
    out = []
    buf = torch.zeros(numel, dtype=dtype, device=device)
    cuda = buf.is_cuda
    for bound in bounds:
        rec = {"channel_bound": bound}
        try:
            comm = make_comm(bound)
            st = torch.cuda.current_stream() if cuda else None
            comm.all_reduce_(buf, st)                       # warm-up (connection set-up)
            if cuda:
                torch.cuda.synchronize()
            dist.barrier()
            t0 = time.time()
            for _ in range(iters):
                comm.all_reduce_(buf, st)
            if cuda:
                torch.cuda.synchronize()
            t = torch.tensor([(time.time() - t0) / iters], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec = float(t)
            gb = buf.numel() * buf.element_size() / 1e9
            rec.update(ms=round(sec * 1e3, 3), bytes=buf.numel() * buf.element_size(), algbw_gbps=round(gb / sec, 1),
                       busbw_gbps=round(gb / sec * 2 * (world - 1) / world, 1), bound_via=getattr(comm, "channel_bound", None),
                       channels_created=(rccl_debug_channels(log_path) or [None])[-1])
            comm.close()
        except Exception as e:      # noqa: BLE001 - a probe must never take the bench down (NativeComm fails on every rank or on none)
            rec["error"] = str(e)[:300]
        out.append(rec)
    return out


class _TorchComm:
    """the probe's view of torch.distributed's own communicator (gloo in the CPU dry run)"""
    channel_bound = "process group"

    def all_reduce_(self, t, stream):
        dist.all_reduce(t)

    def close(self):
        pass


def dry_run_launch(a):
    """CPU-only skeleton of the multi-rank bench (tests/test_bench_launch.py): process group from the launcher's env, the
    barrier / max-over-ranks timing, the metric all-reduce and the ONE JSON line - with a no-op step."""
    from vlrlhf.parallel import all_reduce_mean_scalars, init_distributed_from_env
    rank, local, world = init_distributed_from_env("gloo")
    if world > 1:
        dist.barrier()
    t0 = time.time()
    for _ in range(a.steps):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([time.time() - t0])
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    mean_rank = all_reduce_mean_scalars([float(rank)])[0]
    probe = bucket_probe(world, rank, lambda bound: _TorchComm(), 1 << 18, torch.float32, "cpu", (0, 16), iters=2) if world > 1 else None
    if rank == 0:
        print(json.dumps({"metric": "dry-run (launcher path only)", "value": 0.0, "unit": "pairs/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": round(float(tmax) / max(1, a.steps) * 1e3, 3), "rccl_ranks": world,
                          "mean_rank": mean_rank, "comm": {"bucket_probe": probe}, "INVALID": "dry run"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--prof_sample", type=int, default=8, help="HIP-event bracket around one kernel launch in N inside the timed region (roofline leg); 1 = every launch")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--text_len", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer decoder layers (line is then marked INVALID)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_side_stream", action="store_true")
    ap.add_argument("--ref_pipeline", action="store_true", help="A/B: reference forward of the next batch issued under the update (slower at N=1, see DESIGN.md)")
    ap.add_argument("--precomputed_ref", action="store_true", help="stream precomputed reference log-probs (SURVEY 8f rank 1)")
    ap.add_argument("--lora", action="store_true", help="variant: LoRA DPO of scripts/ddpo_llava.sh (r=128, alpha=256, dropout 0.05)")
    ap.add_argument("--lora_dropout", type=float, default=0.05)
    ap.add_argument("--model", default="llava", choices=["llava", "llava_next", "qwen_vl", "internlm_xc2"],
                    help="llava_next: variant on BASELINE.json configs[3] (LLaVA-Next-Mistral-7B, anyres 672x672 image, DDPO); not the headline line")
    ap.add_argument("--loss_type", default=None)
    ap.add_argument("--dry_run_launch", action="store_true", help="CPU test of the self-launch path: no model, gloo, no-op steps")
    ap.add_argument("--gradient_checkpointing", action="store_true", help="variant: keep only the layer inputs, re-run each layer's forward in the backward (reference scripts' --gradient_checkpointing True)")
    ap.add_argument("--fresh_batches", action="store_true", help="SURVEY 8f-3: every step takes a NEW batch from the input pipeline (JPEG files -> PIL decode + CLIP preprocess in the collator, "
                    "background prefetch, pinned H2D copy, un-memoised concatenated_inputs) instead of rotating four resident ones; reports the host ms per batch")
    ap.add_argument("--lr", type=float, default=2e-8, help="learning rate of the timed steps (kernel arithmetic does not depend on it)")
    ap.add_argument("--no_variants", action="store_true", help="default command at 1 GPU only: skip the driver-clocked variants behind the timed region "
                    "(precomputed reference log-probs in-process, LoRA as a child process of this script)")
    ap.add_argument("--variant_steps", type=int, default=5)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one (one process per GPU on 127.0.0.1, like `accelerate launch` with accelerate_config/ddp.yaml)
        from vlrlhf.parallel import relaunch_under_torchrun
        sys.exit(relaunch_under_torchrun(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    if a.dry_run_launch:
        return dry_run_launch(a)
    line = run(a)
    if line is None:
        return
    if line.pop("_want_variants", False):
        # SURVEY 8f rows 1-2 on the DRIVER's clock: the LoRA recipe of scripts/ddpo_llava.sh as a child process of this very script (the
        # full fine-tune's 177 GiB are released first; the child builds its own model, warms up and times `variant_steps` steps)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        line["variants"]["parent_allocated_gib_before_child"] = round(torch.cuda.memory_allocated() / 2 ** 30, 2)
        line["variants"]["lora"] = variant_child(["--lora"], a, line["ms_per_step"])
    if "_cpu_baseline" in line:
        line.pop("_cpu_baseline")
        line["cpu_baseline"] = cpu_baseline()
    print(json.dumps(line), flush=True)


def variant_child(flags, a, full_ms):
    """one variant of the default command as a child process (same script, same box, right behind the timed region of the parent):
    {"ms_per_step", "pairs_per_s", "ratio_to_full", "steps", "seconds"} or {"error"}"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), *flags, "--steps", str(a.variant_steps), "--warmup", "2", "--no_cpu_baseline", "--no_variants",
           "--pairs", str(a.pairs), "--text_len", str(a.text_len)]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"ms_per_step": d["ms_per_step"], "pairs_per_s": d["value"], "ratio_to_full": round(d["ms_per_step"] / full_ms, 4), "steps": d["steps"],
                "warmup": d["warmup"], "loss_first_step": d["config"].get("loss_first_step"), "workload": d["config"]["workload"],
                "peak_allocated_gib": d["config"].get("peak_allocated_gib"), "seconds": round(time.time() - t0, 1), "command": " ".join(cmd[1:])}
    except Exception as e:      # noqa: BLE001 - a failed variant must not take the headline line down
        return {"error": f"{type(e).__name__}: {e}"[:300], "seconds": round(time.time() - t0, 1)}


def run(a):
    from vlrlhf import _hip
    from vlrlhf.models.Llava import LlavaDPOTrainer, LlavaForRL
    from vlrlhf.parallel import init_distributed_from_env
    from vlrlhf.utils.synthetic import LLAVA_1_5_7B, init_random_model, synthetic_batch
    from types import SimpleNamespace

    rccl_log = rccl_debug_setup() if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None
    rank, local, world = init_distributed_from_env()
    assert world == max(1, a.gpus), f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    cfg = dict(LLAVA_1_5_7B)
    Trainer = LlavaDPOTrainer
    nxt = a.model == "llava_next"
    if nxt:
        from vlrlhf.models.LlavaNext import LLAVA_NEXT_MISTRAL_7B, LlavaNextDPOTrainer, LlavaNextForRL
        from vlrlhf.utils.synthetic import synthetic_batch_anyres
        cfg, LlavaForRL, Trainer = dict(LLAVA_NEXT_MISTRAL_7B), LlavaNextForRL, LlavaNextDPOTrainer
    qwen = a.model == "qwen_vl"
    if qwen:
        from vlrlhf.models.QwenVL import QWEN_VL_CHAT, QwenVLDPOTrainer, QwenVLForRL
        from vlrlhf.utils.synthetic import init_hashed_qwen, synthetic_batch_qwen
        cfg, LlavaForRL, Trainer = dict(QWEN_VL_CHAT), QwenVLForRL, QwenVLDPOTrainer
    ilm = a.model == "internlm_xc2"
    if ilm:
        from vlrlhf.models.InternLMXC2 import INTERNLM_XC2_VL_7B, InternLMXC2DPOTrainer, InternLMXC2ForRL
        cfg, LlavaForRL, Trainer = dict(INTERNLM_XC2_VL_7B), InternLMXC2ForRL, InternLMXC2DPOTrainer
    loss_type = a.loss_type or ("ddpo" if nxt else "sigmoid")
    if a.layers:
        cfg["layers"] = a.layers
    model = LlavaForRL(cfg)
    pad_id = cfg["pad_token_id"] if qwen else (cfg["model_pad_token_id"] if ilm else 0)
    lora_r, lora_alpha = (64, 16) if qwen else ((64, 64) if ilm else (128, 256))    # scripts/dpo_qwenvl.sh / dpo_internlmxc2vl7b.sh / ddpo_llava.sh
    ref = init_hashed_qwen(model, seed=0, std=0.02, policy_delta=1e-3, with_reference=not a.lora) if qwen else \
        init_random_model(model, seed=0, std=0.02, policy_delta=1e-3)
    eng = model.engine
    eng.gradient_checkpointing = bool(a.gradient_checkpointing)
    args = SimpleNamespace(gradient_accumulation_steps=1)
    if a.lora:
        del ref
        tr = Trainer(model, None, 0.1, 0, loss_type, args, None, -100, pad_id,
                             peft_config=dict(r=lora_r, lora_alpha=lora_alpha, lora_dropout=a.lora_dropout, target_modules="auto", bias="none", seed=rank))
        gen = torch.Generator(device=eng.dev)
        gen.manual_seed(4321 + rank)
        for k, t_ in eng.lv.items():             # peft initialises B = 0; random B (seeded) so the adapter GEMMs do real arithmetic
            if ".b_" in k:
                t_.normal_(0.0, 1e-3, generator=gen)
    else:
        tr = Trainer(model, None if a.precomputed_ref else ref, 0.1, 0, loss_type, args, None, -100, pad_id,
                     precompute_ref_log_probs=a.precomputed_ref)
    eng.init_optimizer()
    reducer, transport_fallback, probe = None, None, None
    if world > 1:
        from vlrlhf.parallel import NativeComm, comm_cus_default
        try:
            reducer = eng.make_reducer()
        except _hip.VlrError as e:      # the native transport failed on EVERY rank (its stages agree): measure on torch.distributed's RCCL, and say so
            transport_fallback = str(e)[:300]
            os.environ["VLR_COMM"] = "torch"
            reducer = eng.make_reducer()
        # one 0.4 GB bucket (a decoder layer's gradients) on communicators of their own: RCCL's default channel count against the bound
        # that matches the CU reservation, and against half of it (would 8 CUs do?)
        n_bucket = 4 * cfg["hidden"] * cfg["hidden"] + 3 * cfg["hidden"] * cfg["inter"]
        mk = (lambda bound: NativeComm(channels=bound)) if reducer.transport == "native" else (lambda bound: _TorchComm())
        probe = bucket_probe(world, rank, mk, n_bucket, torch.bfloat16, eng.dev, tuple(sorted({0, 8, comm_cus_default()})) if reducer.transport == "native" else (comm_cus_default(),),
                             log_path=rccl_log)
    tr.ref_on_side_stream = not a.no_side_stream
    if a.ref_pipeline:
        tr.ref_pipeline = True
    # four resident batches per rank (seeds 1234 + rank + 1000*i), rotated: inputs are in HBM before the timed region
    batches = []
    for i in range(4):
        if qwen:
            b_ = tr._prepare_inputs(synthetic_batch_qwen(a.pairs, a.text_len, cfg, seed=1234 + rank + 1000 * i))
        else:
            mk = synthetic_batch_anyres if nxt else synthetic_batch
            b_ = tr._prepare_inputs(mk(a.pairs, a.text_len, cfg["image_token"], 32000, cfg["image_size"], seed=1234 + rank + 1000 * i))
        if a.precomputed_ref:
            with torch.no_grad():
                rc, rr, _, _ = tr.concatenated_forward(ref, b_)
            b_["reference_chosen_logps"], b_["reference_rejected_logps"] = rc, rr
        batches.append(b_)
    hp = dict(lr=a.lr, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.0, max_grad_norm=1.0)   # scripts/dpo_llava.sh:35-41 (lr: see docstring)
    n_step = [0]
    fresh = None
    if a.fresh_batches:
        # the reference's per-step host work (its collator runs on the training thread: models/Llava/__init__.py:435-443, base/collator.py:26-68):
        # image files on disk -> PIL decode -> CLIP resize / crop / normalise -> padded tensors -> pinned -> H2D, here `dataloader_prefetch` = 2
        # batches ahead on a background thread (base/loader.py).  Tokenised rows as `dataset.map(tokenize_row)` leaves them.
        assert a.model == "llava" and not a.precomputed_ref, "--fresh_batches: the LLaVA-1.5 pipeline (the headline configuration)"
        import tempfile
        import numpy as np
        from PIL import Image
        from transformers import CLIPImageProcessor
        from vlrlhf.base.loader import PrefetchLoader
        from vlrlhf.models.Llava import LlavaDPODataCollatorWithPadding
        tmpd = tempfile.mkdtemp(prefix="vlr_fresh_")
        rng = np.random.Generator(np.random.PCG64(99 + rank))
        n_img = 64
        for i in range(n_img):         # VLFeedback-like photographs: 640 x 480 JPEG (smooth content + noise, ~100 KB each)
            base_ = rng.integers(0, 255, size=(15, 20, 3)).astype(np.uint8)
            im = Image.fromarray(base_).resize((640, 480), Image.BICUBIC)
            arr = np.clip(np.asarray(im).astype(np.int16) + rng.integers(-12, 12, size=(480, 640, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(arr).save(os.path.join(tmpd, f"{i}.jpg"), quality=90)
        ip = CLIPImageProcessor(size={"shortest_edge": cfg["image_size"]}, crop_size={"height": cfg["image_size"], "width": cfg["image_size"]})
        coll = LlavaDPODataCollatorWithPadding(pad_token_id=0, label_pad_token_id=-100, processor=SimpleNamespace(image_processor=ip))
        n_need = a.warmup + a.steps + 4

        def row_batches():
            g_ = np.random.Generator(np.random.PCG64(777 + rank))
            lp = a.text_len // 2
            for b in range(n_need):
                rows = []
                for j in range(a.pairs):
                    prompt = g_.integers(3, 32000, size=lp).tolist()
                    prompt[0], prompt[4] = 1, cfg["image_token"]
                    resp = [g_.integers(3, 32000, size=a.text_len - lp).tolist() for _ in range(2)]
                    rows.append(dict(prompt_input_ids=prompt, prompt_attention_mask=[1] * lp, chosen_input_ids=prompt + resp[0],
                                     chosen_attention_mask=[1] * a.text_len, chosen_labels=[-100] * lp + resp[0],
                                     rejected_input_ids=prompt + resp[1], rejected_attention_mask=[1] * a.text_len,
                                     rejected_labels=[-100] * lp + resp[1], img_path=os.path.join(tmpd, f"{(b * a.pairs + j) % n_img}.jpg")))
                yield rows
        # host cost of ONE batch, inline and single-threaded (what the reference pays on its training thread every step)
        first_rows = next(iter(row_batches()))
        t_h = time.time()
        for _ in range(3):
            coll(first_rows)
        host_ms = (time.time() - t_h) / 3 * 1e3
        fresh = dict(it=iter(PrefetchLoader(row_batches, coll, eng.dev, depth=2)), host_ms_per_batch=round(host_ms, 1), files=n_img, dir=tmpd)

    def step():
        if fresh is not None:
            batch_ = tr._prepare_inputs(next(fresh["it"]))
        else:
            batch_ = batches[n_step[0] % len(batches)]
        loss = tr.training_step(model, batch_)
        # as VLDPOTrainer.train does; a no-op unless the reference pipeline is switched on (--ref_pipeline / VLR_REF_PIPELINE=1)
        if fresh is None:
            tr.prefetch_reference(batches[(n_step[0] + 1) % len(batches)])
        eng.optimizer_step(grad_scale=1.0 / world, **hp)
        n_step[0] += 1
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = loss_first = step() if a.warmup > 0 else None
    for _ in range(a.warmup - 1):
        loss = step()
    barrier()
    _hip.lib_profile_start(a.prof_sample)
    t0 = time.time()
    for _ in range(a.steps):
        loss = step()
    barrier()
    dt = time.time() - t0
    prof = _hip.lib_profile_stop()
    tmax = torch.tensor([dt], device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss_last = float(loss)
    loss_first = float(loss_first) if loss_first is not None else loss_last
    import math
    assert math.isfinite(loss_first) and math.isfinite(loss_last), (loss_first, loss_last)
    assert torch.isfinite(eng.norm_out).all(), "non-finite gradient norm in the timed region"
    # exposed communication: the same steps with the gradient exchange switched off (ranks diverge afterwards - timing only); no bucket is
    # issued, so no CUs are given up either: the difference is everything the exchange costs (its exposed part + the bucket-window reservation)
    exposed_ms = None
    if reducer is not None:
        reducer.enabled = False
        step()
        barrier()
        t1 = time.time()
        k2 = max(2, a.steps // 2)
        for _ in range(k2):
            step()
        barrier()
        t2 = torch.tensor([(time.time() - t1) / k2], device="cuda")
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        exposed_ms = round((dt / a.steps - float(t2)) * 1e3, 2)
        reducer.enabled = True
    # HBM traffic of the dominant kernel: PMC passes cannot run inside the timed region; the committed rocprofv3 --pmc
    # result (tools/pmc_traffic.sh -> profiles/rNN_pmc_hbm_traffic.json, newest first) is quoted ONLY when it was taken with the library
    # built from the GEMM sources this run uses (build_hip.kernel_digest() recorded in the file) - a stale file is refused
    import build_hip as _bh
    lib_digest = _bh.kernel_digest()[:16]       # gemm256p.hip + gemm.h + common.h + flags
    traffic = None
    traffic_file = None
    for tag in ("r06", "r05", "r04"):      # the newest counter file taken with THIS library's GEMM sources
        tf = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.json")
        if not os.path.exists(tf):
            continue
        tj = json.load(open(tf))
        if tj.get("source_digest") == lib_digest:
            traffic = round(tj["gemm_hbm_bytes_per_launch"])
            traffic_file = f"profiles/{tag}_pmc_hbm_traffic.json"
            break
        traffic_file = f"profiles/{tag}_pmc_hbm_traffic.json REFUSED: taken with source digest {tj.get('source_digest')}, this library is {lib_digest}"
    # roofline of the DOMINANT kernel: the 8-phase 256x256 GEMM alone (its launches are timed under their own id; the
    # gemm_nt/nn/tn entries of per_kernel are whole vlr_gemm_bf16 calls incl. peeled rows and split-K reduces)
    g_n, g_ms, g_flop = prof["gemm256p"]
    all_ms = sum(prof[k][1] for k in ("gemm_nt", "gemm_nn", "gemm_tn"))
    g_bytes = 0.0   # algorithmic elements moved (A + B + C once) summed over the decoder GEMMs of the timed steps
    S_dec = int(tr.model._last_ctx["S"])            # decoder length of the last policy pass (1599 at configs[1])
    H_, I_, M_ = cfg["hidden"], cfg["inter"], 2 * a.pairs * S_dec
    per_shape = cfg["layers"] * a.steps * ((1 if a.precomputed_ref else 2) + 2)   # fwd (policy [+ ref]) + dgrad + wgrad
    Nqkv_ = eng.Nqkv
    for m_, n_, k_ in ((M_, Nqkv_, H_), (M_, H_, eng.Nq), (M_, 2 * I_, H_), (M_, H_, I_)):
        g_bytes += per_shape * (m_ * k_ + n_ * k_ + m_ * n_)
    if eng.resid_f32:     # o_proj / down_proj forward launches read the fp32 residual and write the fp32 stream (8 bytes per element instead of a 2-byte store)
        g_bytes += 2 * cfg["layers"] * a.steps * (1 if a.precomputed_ref else 2) * (M_ * H_ * 3)       # in 2-byte elements: + 6 bytes per element
    # every entry with its own fraction of the nominal bf16 MFMA peak (attention included: the kernels furthest below it)
    per_kernel = {k: {"launches": n, "ms": round(ms, 2), "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0.0,
                      "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if ms > 0 else 0.0}
                  for k, (n, ms, fl) in prof.items()}
    pairs_per_s = world * a.pairs * a.steps / dt
    per_pair = TFLOP_PER_PAIR["ref_precomputed" if a.precomputed_ref else "ref_in_step"]
    if nxt:
        n_tiles = int(batches[0]["img_input_dict"]["pixel_values"].shape[1])
        per_pair = tflop_per_pair(cfg, S_dec, n_tiles, not a.precomputed_ref)
    if qwen:
        per_pair = tflop_per_pair_qwen(cfg, S_dec, not a.precomputed_ref, a.lora)
    if rank == 0:
        achieved = g_flop / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        line = {
            "metric": "preference-pairs/sec (chosen+rejected) LLaVA-1.5-7B DPO step", "value": round(pairs_per_s, 4),
            "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: LLaVA-1.5-7B DPO bf16, 336x336 image, max_length {a.text_len}, "
                                   f"per-device batch {a.pairs} pairs (S=1599), full fine-tune of LLM+projector, frozen ViT, "
                                   + ("reference log-probs precomputed" if a.precomputed_ref else
                                      "one reference forward per step" + (", issued for the next batch under the update (prefetch_reference)" if tr.ref_pipeline else ", inside the step")),
                       "global_batch_pairs": world * a.pairs, "text_len": a.text_len, "parallelism": f"dp{world}",
                       "layers": cfg["layers"], "lr": a.lr, "resident_batches": 0 if fresh is not None else len(batches),
                       "fresh_batches": None if fresh is None else {"host_ms_per_batch_inline": fresh["host_ms_per_batch"], "jpeg_files": fresh["files"], "prefetch_depth": 2,
                                                                     "pipeline": "640x480 JPEG -> PIL decode -> CLIPImageProcessor(336) -> LlavaDPODataCollatorWithPadding -> pinned -> copy-stream H2D -> _prepare_inputs -> un-memoised concatenated_inputs, a NEW batch every step (SURVEY 8f-3)"},
                       "loss_first_step": loss_first,
                       "loss_last_step": loss_last, "grad_norm_last_step": float(eng.norm_out[0]),
                       "residual_stream": "fp32" if eng.resid_f32 else "bf16", "gradient_checkpointing": bool(eng.gradient_checkpointing),
                       "peak_allocated_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},      # every device buffer of the path is a torch allocation
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "comm": {"transport": reducer.transport if reducer is not None else None,
                     "library": (reducer.transport_note if reducer is not None and reducer.transport == "native" else
                                 ("torch.distributed backend " + dist.get_backend()) if world > 1 else None),
                     "transport_note": reducer.transport_note if reducer is not None else None,     # why the native transport was not used, when it was not
                     "native_transport_error": transport_fallback,       # non-null: vlr_comm_* could not be initialised and the line was measured on torch.distributed's communicator
                     "channel_bound_via": reducer.channel_bound if reducer is not None else None,     # "config" = ncclConfig_t maxCTAs of our communicator, "env" = NCCL_MAX_NCHANNELS
                     "rccl_channels_created": rccl_debug_channels(rccl_log) if world > 1 else None,     # per communicator of rank 0 in creation order (torch's, ours, the probe's), from RCCL's own INIT log
                     "bucket_probe": probe,
                     "comm_cus": reducer.comm_cus if reducer is not None else 0, "compute_cus": _hip.helper("vlr_compute_cus"),
                     "comm_cus_scope": reducer.reserve_scope if reducer is not None else None,      # "backward": the CUs are given up only while buckets are in flight (compute_cus above = outside that window)
                     "rccl_max_min_nchannels": list(reducer.rccl_channels) if reducer is not None else None,     # NCCL_MAX_NCHANNELS := comm_cus for torch's communicator (parallel.rccl_channel_env); MIN only if the user set it
                     "exposed_ms_per_step": exposed_ms, "bytes_per_step": 2 * (eng.lora_layout.numel if a.lora else eng.layout.numel)},
            "roofline": {"bound": "mfma", "kernel": "gemm256p_kernel<A_KS,B_KS> (8-phase 256x256x64 bf16 GEMM: NT/NN/TN)", "achieved": round(achieved, 1),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_file, "traffic_unit": "bytes/launch leaving L2 (2*FETCH_SIZE+WRITE_SIZE, rocprofv3 --pmc, OFFLINE pass of tools/pmc_traffic.sh with the same binary - counters cannot be collected inside the timed region; A+B of a decoder GEMM fit the 256 MB Infinity Cache, so most of the re-reads never reach HBM)",
                         "algorithmic_bytes_per_launch": round(2.0 * g_bytes / max(1, g_n)), "launches": g_n, "avg_launch_ms": round(g_ms / max(1, g_n), 4), "event_sampling": f"one launch in {max(1, a.prof_sample)} bracketed by HIP events (pseudo-random per kernel id); launches and FLOPs exact, ms = sampled mean x launches", "per_kernel": per_kernel,
                         "profile_file": PROFILE_FILE, "gemm_source_digest": lib_digest,
                         "kernel_share_of_step": round(g_ms * 1e-3 / dt, 3), "all_gemm_share_of_step": round(all_ms * 1e-3 / dt, 3),
                         "share_note": "kernel_share = the 256x256 kernel's launches alone; all_gemm_share = every vlr_gemm_* call by layout (fused launches, peeled rows and split-K reduces included), so all_gemm_share >= kernel_share",
                         "step_frac": round(pairs_per_s / world * per_pair / PEAK_BF16_TFLOPS, 4),
                         "step_frac_note": "algorithmic TFLOP per pair of SURVEY.md 8d (lm-head counted on all S positions; the kernel evaluates it on the response rows only, ~1.3 % fewer executed FLOPs) / nominal 2516.6 TF/s; this chip sustains 1828 TF/s on random bf16 operands in a register-only MFMA loop (profiles/r02_gemm_ceiling_mfma_only_and_ablation.txt)"},
        }
        if a.layers:
            line["INVALID"] = "reduced layer count (debug run)"
        if nxt:
            line["metric"] = "preference-pairs/sec (chosen+rejected) LLaVA-Next-Mistral-7B DPO step"
            line["config"]["workload"] = (f"variant on BASELINE.json configs[3]: LLaVA-Next-Mistral-7B {loss_type.upper()} bf16, anyres 672x672 image "
                                          f"({n_tiles} tiles -> 2928 image features), max_length {a.text_len}, per-device batch {a.pairs} pairs (S={S_dec}), "
                                          "full fine-tune of LLM+projector+image_newline, frozen ViT, reference forward inside the step")
            line["config"]["variant"] = "llava_next (not the headline configuration)"
            line["config"]["tflop_per_pair"] = round(per_pair, 2)
        if ilm:
            line["metric"] = "preference-pairs/sec (chosen+rejected) InternLM-XComposer2-VL-7B DPO step"
            per_pair_i = tflop_per_pair(cfg, S_dec, 1, not a.precomputed_ref)
            line["config"]["workload"] = (f"variant on BASELINE.json configs[4]: InternLM-XComposer2-VL-7B DPO bf16, 490x490 image (1225 patches), "
                                          f"max_length {a.text_len}, per-device batch {a.pairs} pairs (S={S_dec}), "
                                          + (f"LoRA r={lora_r} alpha={lora_alpha} dropout={a.lora_dropout} on wqkv / wo / w1 / w2 / w3 over the frozen PLoRA decoder "
                                             "(scripts/dpo_internlmxc2vl7b.sh), reference = adapters disabled" if a.lora else
                                             "full fine-tune of the decoder incl. its PLoRA pairs, reference forward inside the step")
                                          + ", frozen ViT + projector; PLoRA on the fused C layer calls (vlr_decoder_layer_*_lora_ex)" + ("; peft LoRA stacked on it on the two-adapter passes (vlr_decoder_layer_*_lora2)" if a.lora else ""))
            line["config"]["variant"] = "internlm_xc2" + ("+lora" if a.lora else "") + " (not the headline configuration)"
            line["config"]["tflop_per_pair"] = round(per_pair_i, 2)
            line["roofline"]["step_frac"] = None if a.lora else round(pairs_per_s / world * per_pair_i / PEAK_BF16_TFLOPS, 4)
        elif qwen:
            line["metric"] = "preference-pairs/sec (chosen+rejected) Qwen-VL-Chat DPO step"
            line["config"]["workload"] = (f"variant on BASELINE.json configs[2]: Qwen-VL-Chat DPO bf16, 448x448 image (1024 patches -> 256 resampler "
                                          f"slots inside the ids), max_length {a.text_len}, per-device batch {a.pairs} pairs (S={S_dec}), "
                                          + (f"LoRA r={lora_r} alpha={lora_alpha} dropout={a.lora_dropout} on c_attn / attn.c_proj / w1 / w2 (scripts/dpo_qwenvl.sh), frozen base, "
                                             "reference = adapters disabled, frozen vision tower incl. resampler" if a.lora else
                                             "full fine-tune of the language model + resampler (attn_pool), reference forward inside the step, frozen ViT trunk"))
            line["config"]["variant"] = "qwen_vl" + ("+lora" if a.lora else "") + " (not the headline configuration)"
            line["config"]["tflop_per_pair"] = round(per_pair, 2)
            line["roofline"]["step_frac"] = round(pairs_per_s / world * per_pair / PEAK_BF16_TFLOPS, 4)
        elif a.lora:
            line["config"]["workload"] = line["config"]["workload"].replace(
                "full fine-tune of LLM+projector", "LoRA r=128 alpha=256 dropout=0.05 on the 7 decoder linears (scripts/ddpo_llava.sh), frozen base")
            line["config"]["variant"] = "lora (not the headline configuration)"
            line["roofline"]["step_frac"] = None
        if not a.no_cpu_baseline and world == 1:
            line["_cpu_baseline"] = True
        plain = world == 1 and a.model == "llava" and not (a.lora or a.precomputed_ref or a.layers or a.gradient_checkpointing or a.fresh_batches or a.ref_pipeline)
        if plain and not a.no_variants:
            # SURVEY 8f row 1 on the driver's clock, in-process: the same engine and batches, the reference log-probs stored on the batches
            # (what trl's precompute_ref_log_probs leaves on the dataset rows) - the trainer then runs no reference forward
            for b_ in batches:
                with torch.no_grad():
                    rc, rr, _, _ = tr.concatenated_forward(ref, b_)
                b_["reference_chosen_logps"], b_["reference_rejected_logps"] = rc, rr
            step()
            barrier()
            t1 = time.time()
            for _ in range(a.variant_steps):
                lv = step()
            barrier()
            ms = (time.time() - t1) / a.variant_steps * 1e3
            line["variants"] = {"note": "timed behind the headline region of this same run (same box, same process unless `command` is given); not part of `value`",
                                "precomputed_ref": {"ms_per_step": round(ms, 2), "pairs_per_s": round(a.pairs / ms * 1e3, 4), "ratio_to_full": round(ms / line["ms_per_step"], 4),
                                                    "steps": a.variant_steps, "warmup": 1, "loss_last_step": float(lv),
                                                    "how": "reference_chosen/rejected_logps stored on the resident batches: no reference forward in the step (--precomputed_ref)"}}
            line["_want_variants"] = True
    else:
        line = None
    if fresh is not None:
        import shutil
        fresh["it"].close()                 # stops the prefetch thread
        shutil.rmtree(fresh["dir"], ignore_errors=True)
    if world > 1:
        dist.barrier()
        if reducer is not None and reducer.native is not None:
            reducer.native.close()
        dist.destroy_process_group()
        if rccl_log and os.path.basename(rccl_log).startswith("vlr_rccl_init_"):      # our own per-process INIT log: parsed above, not left in /tmp
            try:
                os.remove(rccl_log)
            except OSError:
                pass
    return line


if __name__ == "__main__":
    main()
