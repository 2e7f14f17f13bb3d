#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 1; do VLR_NORM_BWD_EARLY=$v timeout 100 python tools/norm_time.py 2>/dev/null | tail -1; done
timeout 100 python -m pytest tests/test_hip_kernels.py -q -k "rmsnorm" 2>&1 | tail -1
