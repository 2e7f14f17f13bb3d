#!/bin/bash
# same-box A/B of environment switches on the default command: tools/ab_env.sh <outdir> "VAR=val [VAR2=val2]" ... ("-" = no switch); each twice, interleaved
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$1; shift; mkdir -p $O
for rep in 1 2; do
  i=0
  for sw in "$@"; do
    i=$((i+1))
    if [ "$sw" = "-" ]; then pre=""; else pre="$sw"; fi
    env $pre timeout 300 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_variants $AB_ARGS 2>/dev/null | tail -1 > $O/bench_${i}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$O/bench_${i}_$rep.json")); print("[$sw]", $rep, d["ms_per_step"], d["roofline"]["frac"], d["config"].get("loss_first_step"))
except Exception as e: print("[$sw]", $rep, "FAILED", e)
PY
  done
done
