#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call11; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -k "gemm or swiglu or rope or lmhead" > $O/pytest_gemm.txt 2>&1
tail -2 $O/pytest_gemm.txt
bash tools/ab_bench.sh $O vl-rlhf_amd/libvlr_hip_nt0.so default
