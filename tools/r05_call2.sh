#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call2; mkdir -p $O
VLR_GEMM_SPLIT=0 VLR_LIB=vl-rlhf_amd/libvlr_hip_trace.so timeout 300 python tools/gemm_ktile_probe.py > $O/ktile_probe.txt 2>&1
echo call2 done
