#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 1; do VLR_NORM_FWD_REG=$v timeout 120 python tools/norm_time.py 2>/dev/null; done
timeout 200 python -m pytest tests/test_hip_kernels.py -q -k "rmsnorm" 2>&1 | tail -2
