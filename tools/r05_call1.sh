#!/bin/bash
# round 5, GPU call 1: box baseline + tile timeline of the persistent GEMM with / without de-phased workgroups + the peel shapes of the 128x128 kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call1; mkdir -p $O
timeout 300 python bench.py --steps 8 --warmup 2 --no_cpu_baseline > $O/bench_base.json 2> $O/bench_base.err
T=vl-rlhf_amd/libvlr_hip_trace.so
VLR_LIB=$T timeout 200 python tools/gemm_tile_trace.py > $O/trace_plain.txt 2>&1
for d in "8,16" "8,32" "4,24" "32,32" "2,16"; do
  echo "=== VLR_GEMM_DEPHASE=$d" >> $O/trace_dephase.txt
  VLR_GEMM_DEPHASE=$d VLR_LIB=$T timeout 200 python tools/gemm_tile_trace.py >> $O/trace_dephase.txt 2>&1
done
for e in 1 2 3; do
  echo "=== VLR_EPI_ABLATE=$e" >> $O/trace_epi_ablate.txt
  VLR_EPI_ABLATE=$e VLR_LIB=$T timeout 200 python tools/gemm_tile_trace.py >> $O/trace_epi_ablate.txt 2>&1
done
for s in 128 256 512 1024; do
  echo "=== VLR_SPLITK_TARGET=$s" >> $O/gemm128_splitk.txt
  VLR_SPLITK_TARGET=$s timeout 200 python tools/gemm128_bench.py >> $O/gemm128_splitk.txt 2>&1
done
echo call1 done
