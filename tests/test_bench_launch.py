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


def test_bench_self_launches_eight_ranks():
    """the driver's largest scaling point, as far as a CPU can take it: 8 gloo ranks through the same self-launch, barrier / max-over-ranks
    timing, metric all-reduce and bucket probe (bound 0 and 16: two records), ONE JSON line"""
    line = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--dry_run_launch"], env={"OMP_NUM_THREADS": "1"})
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8
    assert abs(line["mean_rank"] - 3.5) < 1e-9
    probe = line["comm"]["bucket_probe"]
    assert [p["channel_bound"] for p in probe] == [0, 16] and all("error" not in p and p["busbw_gbps"] > 0 for p in probe), probe


def test_rccl_init_log_parsing(tmp_path):
    """bench.rccl_debug_channels reads the channel counts RCCL says it created, per communicator, from its INIT log"""
    sys.path[:0] = [ROOT]
    import importlib
    bench = importlib.import_module("bench")
    log = tmp_path / "rccl.log"
    log.write_text("h:1:1 [0] NCCL INFO comm 0x1 rank 0 nRanks 8 nNodes 1 localRanks 8 localRank 0 MNNVL 0\n"
                   "h:1:1 [0] NCCL INFO 32 coll channels, 0 collnet channels, 0 nvls channels, 32 p2p channels, 4 p2p channels per peer\n"
                   "h:1:1 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer\n")
    assert bench.rccl_debug_channels(str(log)) == [32, 16]
    log.write_text("NCCL INFO Channel 00/24 : 0 1 2 3\nNCCL INFO Channel 01/24 : 0 1 2 3\nNCCL INFO Channel 00/16 : 0 1\n")
    assert bench.rccl_debug_channels(str(log)) == [24, 16]
    assert bench.rccl_debug_channels(str(tmp_path / "missing")) == [] and bench.rccl_debug_channels(None) == []


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
