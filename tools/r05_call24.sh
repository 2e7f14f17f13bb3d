#!/bin/bash
# two-adapter decoder (LoRA over PLoRA): adapter terms of down_proj as the addend of the fused SwiGLU-backward dgrad GEMM - tests, then A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call24; mkdir -p $O
[ -n "$NOTEST" ] || timeout 300 python -m pytest tests/test_hip_internlm.py tests/test_hip_fullsize_qwen_internlm.py -x -q -k "lora" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for sw in VLR_LORA_FUSE_DOWN=0 VLR_LORA_FUSE_DOWN=1 VLR_LORA_FUSE_DOWN=0 VLR_LORA_FUSE_DOWN=1; do
  env $sw timeout 180 python bench.py --model internlm_xc2 --lora --steps ${STEPS:-5} --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; d=json.load(open('$O/b.json')); print('$sw', d['ms_per_step'], d['config'].get('loss_first_step'))" | tee -a $O/ab.txt
done
