#!/bin/bash
# the whole -m gpu suite + smoke, as the driver runs them
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_suite; mkdir -p $O
timeout 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" >> $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo suite done
