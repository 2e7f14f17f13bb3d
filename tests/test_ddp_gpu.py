"""Two-process data-parallel DPO step on ONE MI355X (both ranks on cuda:0, gloo as the transport because RCCL refuses two ranks
on one device): exercises the product's DDP path on the GPU - per-layer buckets signalled from the HIP backward, the
communication stream, 1/world folded into the optimizer's gradient scale - and checks that two ranks with one pair each
end the step with the same weights as one process with both pairs."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sub(b, i):
    out = {}
    for k, v in b.items():
        if isinstance(v, torch.Tensor):
            out[k] = v[i:i + 1]
        elif isinstance(v, dict):
            out[k] = {kk: vv[i:i + 1] for kk, vv in v.items()}
        elif isinstance(v, list):
            out[k] = v[i:i + 1]
    return out


def _step(cfg, W, W_ref, batch, world):
    from types import SimpleNamespace
    from vlrlhf.models.Llava import LlavaDPOTrainer, LlavaForRL
    model = LlavaForRL.from_state_dict(cfg, W)
    ref = model.create_reference_model()
    ref.weights.load_state_dict(W_ref)
    eng = model.engine
    eng.init_optimizer()
    if world > 1:
        eng.make_reducer()
    tr = LlavaDPOTrainer(model, ref, cfg["beta"], 0, "sigmoid", SimpleNamespace(gradient_accumulation_steps=1), None, -100, 0)
    loss = tr.training_step(model, batch)
    o = cfg["optim"]
    eng.optimizer_step(o["lr"], o["beta1"], o["beta2"], o["eps"], o["weight_decay"], o["max_grad_norm"], grad_scale=1.0 / world)
    torch.cuda.synchronize()
    return float(loss), eng.policy.flat.float().cpu(), eng.grads.float().cpu(), eng.grad_norm()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vl-rlhf_amd")]
    import torch.distributed as dist
    from tests.golden_util import load_case
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
        loss, flat, grads, norm = _step(cfg, W, W_ref, _sub(batch, rank), world)
        q.put((rank, loss, flat.numpy(), grads.numpy(), norm, None))   # numpy: pickled by value (no shared-memory handles)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # surfaced by the parent
        q.put((rank, None, None, None, None, repr(e)))


def test_two_rank_step_equals_one_rank_full_batch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vl-rlhf_amd")]
    from tests.golden_util import load_case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:                      # a rank stuck in a collective must not outlive the test
            if p.is_alive():
                p.terminate()
    for r in res:
        assert r[5] is None, r[5]
    z, cfg, W, W_ref, batch, rows = load_case("llava_hipsmall")
    loss1, flat1, grads1, norm1 = _step(cfg, W, W_ref, batch, 1)
    (_, la, fa, ga, na, _), (_, lb, fb, gb, nb, _) = res
    fa, ga, fb, gb = (torch.from_numpy(x) for x in (fa, ga, fb, gb))
    assert torch.equal(ga, gb) and torch.equal(fa, fb)               # both ranks hold the same reduced gradients / new weights
    assert abs(0.5 * (la + lb) - loss1) < 2e-3                          # mean of the per-rank losses = full-batch loss
    # sum-reduced gradients * 1/world == full-batch gradient (loss = mean over pairs), bf16 buffers
    cos = float(torch.dot(ga, grads1) / (ga.norm() * grads1.norm()))
    assert cos > 0.995 and abs(float(ga.norm()) / (2 * float(grads1.norm())) - 1) < 0.03, (cos, float(ga.norm()), float(grads1.norm()))
    assert abs(na - norm1) < 0.03 * norm1
    d2, d1 = fa - W_flat(cfg, W), flat1 - W_flat(cfg, W)
    assert float((d2 - d1).norm()) < 0.3 * float(d1.norm())           # same AdamW update (first step ~ lr*sign(g), bf16 weights)


def W_flat(cfg, W):
    from vlrlhf.engine import ParamLayout, WeightSet
    ws = WeightSet(ParamLayout(cfg), torch.device("cpu"))
    ws.load_state_dict({k: v for k, v in W.items() if not k.startswith("vision_tower.")}, strict=False)
    return ws.flat.float()
