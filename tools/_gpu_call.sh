#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for m in "full:" "lora:--lora"; do
  tag=${m%%:*}; flag=${m#*:}
  rm -rf /tmp/prof_ilm
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ilm -o r -- python bench.py --model internlm_xc2 $flag --steps 3 --warmup 2 --no_cpu_baseline > gpurun_out/ilm_${tag}_under_rocprof.json 2>/dev/null
  f=$(find /tmp/prof_ilm -name "*kernel_trace.csv" | head -1)
  python tools/step_trace.py $f 2 3 gpurun_out/r04_steady_state_kernel_breakdown_internlm_${tag}_final.txt > /dev/null
  head -3 gpurun_out/r04_steady_state_kernel_breakdown_internlm_${tag}_final.txt | cut -c1-120
done
