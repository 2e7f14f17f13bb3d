#!/bin/bash
# parity + timing of the 64-query forward kernel (VLR_ATTN_FWD3=1) against the 32-query one on one MI355X -> gpurun_out/r04_attn_fwd3_*
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
o=gpurun_out/r04_attn_fwd3_check.txt
: > $o
echo "== parity: tests/test_hip_kernels.py -k attention with VLR_ATTN_FWD3=1" >> $o
VLR_ATTN_FWD3=1 timeout 600 python -m pytest tests/test_hip_kernels.py -k "attention" -x -q 2>&1 | tail -15 >> $o
echo "== timing at the step's shape (8 x 1599, 32 heads), 20 launches" >> $o
for f in 0 1; do VLR_ATTN_FWD3=$f timeout 120 python tools/attn_time.py 20 2>&1 | grep attn | sed "s/^/FWD3=$f /" >> $o; done
echo "== timing at the LLaVA-Next shape (4 x 4975, 32 / 8 heads)" >> $o
for f in 0 1; do VLR_ATTN_FWD3=$f timeout 120 python tools/attn_time.py 10 4 4975 32 8 2>&1 | grep attn | sed "s/^/FWD3=$f /" >> $o; done
cat $o
