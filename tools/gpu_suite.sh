#!/bin/bash
# the whole -m gpu suite + smoke, as the driver runs them:  tools/gpu_suite.sh <tag>   -> gpurun_out/<tag>_suite/{pytest_gpu,smoke}.txt
tag=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${tag}_suite; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" >> $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt
