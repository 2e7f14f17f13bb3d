#!/bin/bash
# same-box A/B of library builds on the default command: tools/ab_bench.sh <outdir> <lib-or-"default"> ... (each twice, interleaved)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$1; shift; mkdir -p $O
for rep in 1 2; do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    if [ "$lib" = default ]; then unset VLR_LIB; else export VLR_LIB=$lib; fi
    timeout 300 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_variants $AB_ARGS 2>/dev/null | tail -1 > $O/bench_${tag}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${tag}_$rep.json")); print("$tag", $rep, d["ms_per_step"], d["roofline"]["frac"], d["config"].get("loss_first_step"))
except Exception as e: print("$tag", $rep, "FAILED", e)
PY
  done
done
