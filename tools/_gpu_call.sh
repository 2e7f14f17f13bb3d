#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
out=gpurun_out/r04_bench_1gpu_internlm_variants_final.json
: > $out
for v in "--model internlm_xc2" "--model internlm_xc2 --lora" "--model internlm_xc2 --text_len 512" "--model internlm_xc2 --text_len 512 --lora"; do
  timeout 300 python bench.py $v --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 >> $out
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_1gpu_internlm_variants_final.json"):
    d=json.loads(l); print(d["config"].get("variant"), d["config"].get("text_len"), d["ms_per_step"], d["value"])
PY
