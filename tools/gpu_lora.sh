#!/bin/bash
# LoRA adapter kernels on the GPU box: kernel tests, the per-call microbenchmark against the HBM floor and a same-box in-step A/B of
# environment switches.    tools/gpu_lora.sh <outdir> [tests|micro|step|all] ["ENV=.. ENV=.." ...]   (each env set: bench.py --lora, twice, interleaved)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$1; what=${2:-all}; shift; shift; mkdir -p $O
if [ $what = tests ] || [ $what = all ]; then
  timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -k "lora or adapter or dropout" > $O/pytest_lora.txt 2>&1; tail -3 $O/pytest_lora.txt
fi
if [ $what = micro ] || [ $what = all ]; then
  timeout 600 python tools/lora_gemm_bench.py --shape llava > $O/micro_llava.txt 2>&1; cat $O/micro_llava.txt | grep -v amdgpu.ids
fi
if [ $what = step ] || [ $what = all ]; then
  [ $# -eq 0 ] && set -- "VLR_NOP=1"
  for rep in 1 2; do
    i=0
    for envs in "$@"; do
      i=$((i+1))
      env $envs timeout 400 python bench.py --lora --steps 6 --warmup 2 --no_cpu_baseline 2>$O/err_${i}_$rep.txt | tail -1 > $O/lora_${i}_$rep.json
      python - <<PY
import json
try:
    d=json.load(open("$O/lora_${i}_$rep.json")); print("$envs", $rep, d["ms_per_step"], d["config"].get("loss_first_step"))
except Exception as e: print("$envs", $rep, "FAILED", e)
PY
    done
  done
fi
