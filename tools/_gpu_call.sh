#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_kernels.py -q -k "row_set or packed or bits" -x 2>&1 | tail -15
timeout 300 python -m pytest tests/test_hip_internlm.py -q -x 2>&1 | tail -5
for m in "--lora" ""; do
  timeout 300 python bench.py --model internlm_xc2 $m --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('internlm', '$m', d['ms_per_step'])"
done
