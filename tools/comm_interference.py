#!/usr/bin/env python3
"""Single-GPU interference bench for the data-parallel gradient exchange: what the step loses when kernels of
the communication stream run beside the backward, and what leaving CUs free (vlr_set_comm_cus) gives back.

One MI355X cannot run a multi-rank RCCL ring, and a 1-rank ncclAllReduce launches nothing, so the ring kernel is replaced by its
footprint: after every gradient bucket of the real backward (lm_head, 32 decoder layers, tail - the 34 buckets of GradReducer) the
communication stream runs `vlr_comm_probe` - `wgs` workgroups streaming 2 x bucket bytes (a ring all-reduce reads and writes every
element about twice) - exactly where vlr_allreduce_bucket would run.  Reported: ms/step for wgs in {0, 16, 32} x comm_cus in
{0, 8, 16, 32} x the window the CUs are given up for (while buckets are in flight - GradReducer's default since round 5 - or the whole step).

    python tools/comm_interference.py [--steps 4]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vl-rlhf_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from vlrlhf import _hip  # noqa: E402
from vlrlhf.parallel import GradReducer  # noqa: E402


class ProbeReducer(GradReducer):
    """GradReducer whose transport is the probe kernel (same stream, same events, same bucket order)"""

    def __init__(self, flat, buckets, wgs, comm_cus=0, scope="backward"):
        super().__init__(flat, buckets)
        self.world = 2                  # pretend: the buckets are issued
        self.wgs = wgs
        self.comm_cus, self.reserve_scope = comm_cus, scope
        if scope == "step":
            self._reserve(True)
        self.scratch = torch.empty(max(hi - lo for lo, hi in buckets.values()), dtype=flat.dtype, device=flat.device)

    def bucket_ready(self, name):
        lo, hi = self.buckets[name]
        if hi <= lo:
            return
        self._reserve(True)             # as GradReducer.bucket_ready: the launches after the first bucket leave comm_cus CUs free
        self._issued = True
        if not self.wgs:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        self._issued = True
        n = (hi - lo) * 2 // 16 * 16
        with torch.cuda.stream(self.stream):
            for _ in range(2):
                _hip.call("vlr_comm_probe", self.grads[lo:hi], self.scratch, n, self.wgs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--wgs", default="0,16,32")
    ap.add_argument("--cus", default="0,8,16,32")
    ap.add_argument("--scopes", default="backward,step", help="when the CUs are given up: while buckets are in flight (GradReducer's default) / the whole step")
    a = ap.parse_args()
    from vlrlhf.models.Llava import LlavaDPOTrainer, LlavaForRL
    from vlrlhf.utils.synthetic import LLAVA_1_5_7B, init_random_model, synthetic_batch
    cfg = dict(LLAVA_1_5_7B)
    model = LlavaForRL(cfg)
    ref = init_random_model(model, seed=0, std=0.02, policy_delta=1e-3)
    eng = model.engine
    tr = LlavaDPOTrainer(model, ref, 0.1, 0, "sigmoid", SimpleNamespace(gradient_accumulation_steps=1), None, -100, 0)
    eng.init_optimizer()
    batches = [tr._prepare_inputs(synthetic_batch(4, 1024, cfg["image_token"], 32000, cfg["image_size"], seed=1234 + 1000 * i)) for i in range(2)]
    hp = dict(lr=2e-8, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.0, max_grad_norm=1.0)
    n = [0]

    def step():
        tr.training_step(model, batches[n[0] % 2])
        eng.optimizer_step(grad_scale=0.5, **hp)
        n[0] += 1

    out = {}
    for wgs in [int(v) for v in a.wgs.split(",")]:
        for k, scope in [(int(v), sc) for v in a.cus.split(",") for sc in (a.scopes.split(",") if int(v) else ["-"])]:
            _hip.helper("vlr_set_comm_cus", 0)
            eng.reducer = ProbeReducer(eng.grads, eng.layout.bucket_after, wgs, k, scope)
            step()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            ms = (time.time() - t0) / a.steps * 1e3
            out[f"probe_wgs={wgs},comm_cus={k},scope={scope}"] = round(ms, 2)
            print(f"probe workgroups {wgs:3d}  comm_cus {k:3d}  scope {scope:8s}: {ms:8.2f} ms/step", flush=True)
    _hip.helper("vlr_set_comm_cus", -1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
