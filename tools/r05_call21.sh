#!/bin/bash
# per-kernel steady-state breakdown with 16 CUs left to RCCL for the whole step (env) next to the whole chip, one box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call21; mkdir -p $O
for k in 0 16; do
  VLR_COMM_CUS=$k rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$k -o r -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline > $O/bench_cus$k.json 2>/dev/null
  f=$(find /tmp/prof_c$k -name "*kernel_trace.csv" | head -1)
  python tools/step_trace.py $f 2 5 $O/breakdown_cus$k.txt > /dev/null
  rm -rf /tmp/prof_c$k
done
echo done
