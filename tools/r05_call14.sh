#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call14; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_e2e.py -q -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
bash tools/ab_env.sh $O/ab VLR_GEMM_SCHED=96 -
