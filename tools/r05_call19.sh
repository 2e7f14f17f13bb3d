#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call19; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -k "segment or row_tile_skip" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
AB_ARGS="--model internlm_xc2 --lora" bash tools/ab_env.sh $O/lora VLR_SEG_SKIP=0 -
AB_ARGS="--model internlm_xc2" bash tools/ab_env.sh $O/full VLR_SEG_SKIP=0 -
