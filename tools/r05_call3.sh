#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call3; mkdir -p $O
T=vl-rlhf_amd/libvlr_hip_trace.so
for s in 0 32; do
  echo "=== VLR_GEMM_SCHED=$s" >> $O/trace.txt
  VLR_GEMM_SCHED=$s VLR_LIB=$T timeout 200 python tools/gemm_tile_trace.py >> $O/trace.txt 2>&1
done
timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "gemm" > $O/pytest_gemm_sched0.txt 2>&1
VLR_GEMM_SCHED=32 timeout 300 python -m pytest tests/test_hip_kernels.py -q -x -k "gemm" > $O/pytest_gemm_sched32.txt 2>&1
for s in 0 32 0 32; do
  VLR_GEMM_SCHED=$s timeout 300 python bench.py --steps 8 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 > $O/bench_sched${s}_$RANDOM.json
done
echo call3 done
