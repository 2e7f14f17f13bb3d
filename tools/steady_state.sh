#!/bin/bash
# steady-state per-kernel breakdown of one bench configuration:  tools/steady_state.sh <tag> <name> [bench.py args ...]
#   -> gpurun_out/<tag>_steady_state_kernel_breakdown_<name>.txt  (rocprofv3 --kernel-trace of 5 timed steps behind 2 warm-up steps)
tag=$1; name=$2; shift; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
D=/tmp/prof_${tag}_${name}
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_variants "$@" > gpurun_out/${tag}_bench_under_rocprof_${name}.json 2>/dev/null
f=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py $f 2 5 gpurun_out/${tag}_steady_state_kernel_breakdown_${name}.txt > /dev/null
rm -rf $D
head -3 gpurun_out/${tag}_steady_state_kernel_breakdown_${name}.txt
