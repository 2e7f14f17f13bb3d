"""`python bench.py --gpus N` without a launcher must start N ranks itself (one per GPU, torch.distributed.run on 127.0.0.1)
and print ONE JSON line with n_gpus = N.  CPU check of that path with 2 gloo ranks and no model (--dry_run_launch); the
real thing runs on RCCL when the driver launches the scaling bench."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    line = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry_run_launch"])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 3
    assert abs(line["mean_rank"] - 0.5) < 1e-9          # the metric all-reduce saw both ranks


def test_bench_single_rank_needs_no_launcher():
    line = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry_run_launch"])
    assert line["n_gpus"] == 1


def test_relaunch_passes_arguments_and_exit_code(tmp_path):
    sys.path[:0] = [os.path.join(ROOT, "vl-rlhf_amd")]
    from vlrlhf.parallel import relaunch_under_torchrun
    script = tmp_path / "w.py"
    script.write_text("import os, sys\nassert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "assert sys.argv[1:] == ['--x', '7']\nsys.exit(0 if os.environ['LOCAL_RANK'] in '01' else 3)\n")
    assert relaunch_under_torchrun(str(script), ["--x", "7"], 2) == 0
