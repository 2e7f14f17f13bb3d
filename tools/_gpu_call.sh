#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_hip_kernels.py -q -k "k_tile_list or row_set" -x 2>&1 | tail -15
timeout 300 python -m pytest tests/test_hip_internlm.py -q -x 2>&1 | tail -3
for v in 0 1; do
VLR_ROW_TILES=$v timeout 300 python bench.py --model internlm_xc2 --steps 5 --warmup 2 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('internlm full row_tiles=$v', d['ms_per_step'])"
done
