#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call7; mkdir -p $O
timeout 300 python -m pytest tests/test_comm_gpu.py tests/test_ddp_gpu.py -x -q -m gpu -s > $O/pytest_comm.txt 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no_cpu_baseline > $O/bench_resident.json 2> $O/bench_resident.err
timeout 300 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --fresh_batches > $O/bench_fresh.json 2> $O/bench_fresh.err
echo call7 done
