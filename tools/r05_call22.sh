#!/bin/bash
# per-kernel steady-state breakdown of the InternLM-XComposer2 full fine-tune and LoRA steps
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call22; mkdir -p $O
for v in full lora; do
  extra=""; [ $v = lora ] && extra="--lora"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_i$v -o r -- python bench.py --model internlm_xc2 $extra --steps 3 --warmup 2 --no_cpu_baseline > $O/bench_$v.json 2>/dev/null
  f=$(find /tmp/prof_i$v -name "*kernel_trace.csv" | head -1)
  python tools/step_trace.py $f 2 3 $O/breakdown_$v.txt > /dev/null
  rm -rf /tmp/prof_i$v
done
echo done
