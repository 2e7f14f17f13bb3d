#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call12; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -k "segment or row_tile_skip or lora" > $O/pytest_seg.txt 2>&1; tail -3 $O/pytest_seg.txt
timeout 600 python -m pytest tests/test_hip_internlm.py -q -x > $O/pytest_internlm.txt 2>&1; tail -2 $O/pytest_internlm.txt
AB_ARGS="--model internlm_xc2 --lora" bash tools/ab_env.sh $O/lora VLR_SEG_SKIP=0 -
AB_ARGS="--model internlm_xc2" bash tools/ab_env.sh $O/full VLR_SEG_SKIP=0 -
