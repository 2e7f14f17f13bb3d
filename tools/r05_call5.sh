#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call5; mkdir -p $O
T=vl-rlhf_amd/libvlr_hip_trace.so
for c in 0 1; do
  echo "=== VLR_GEMM_TRACE_CLK=$c (1: shader cycles / 100)" >> $O/probe_clk.txt
  VLR_GEMM_TRACE_CLK=$c VLR_GEMM_SPLIT=0 VLR_LIB=$T timeout 300 python tools/gemm_ktile_probe.py >> $O/probe_clk.txt 2>&1
  echo "=== VLR_GEMM_TRACE_CLK=$c (1: shader cycles / 100)" >> $O/trace_clk.txt
  VLR_GEMM_TRACE_CLK=$c VLR_LIB=$T timeout 300 python tools/gemm_tile_trace.py >> $O/trace_clk.txt 2>&1
done
echo call5 done
