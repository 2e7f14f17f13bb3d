#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call6; mkdir -p $O
T=vl-rlhf_amd/libvlr_hip_trace.so
for w in 0 3 50 300; do
 for c in 0 1; do
  echo "=== PROBE_WARM=$w VLR_GEMM_TRACE_CLK=$c" >> $O/probe_warm.txt
  PROBE_WARM=$w VLR_GEMM_TRACE_CLK=$c VLR_GEMM_SPLIT=0 VLR_LIB=$T timeout 300 python tools/gemm_ktile_probe.py "4096,4096]" >> $O/probe_warm.txt 2>&1
 done
done
echo call6 done
