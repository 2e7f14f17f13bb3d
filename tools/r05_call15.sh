#!/bin/bash
# the round's GEMM changes against their absence on ONE box: the r04 K loops (build variant) + the r04 tile map (VLR_GEMM_SCHED=0) vs the default
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call15; mkdir -p $O
for rep in 1 2 3; do
  VLR_LIB=vl-rlhf_amd/libvlr_hip_r04k.so VLR_GEMM_SCHED=0 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline 2>/dev/null | tail -1 > $O/bench_r04gemm_$rep.json
  timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline 2>/dev/null | tail -1 > $O/bench_r05_$rep.json
  python - <<PY
import json
for t in ("r04gemm", "r05"):
    d=json.load(open("$O/bench_%s_$rep.json" % t)); print(t, $rep, d["ms_per_step"], d["roofline"]["frac"])
PY
done
timeout 600 python -m pytest tests/test_hip_depth.py -q -x -s -k "seed_sweep" 2>&1 | grep "seed sweep\] n" > $O/seed_sweep.txt
cat $O/seed_sweep.txt
