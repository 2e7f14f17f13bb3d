#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for s in 32 0 32 0; do
VLR_GEMM_SCHED=$s timeout 300 python bench.py --model internlm_xc2 --lora --steps 5 --warmup 2 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('internlm lora sched $s', d['ms_per_step'])"
done
for s in 32 0; do
VLR_GEMM_SCHED=$s timeout 300 python bench.py --lora --steps 8 --warmup 3 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('llava lora sched $s', d['ms_per_step'])"
done
