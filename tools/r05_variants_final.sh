#!/bin/bash
# the variants the round's last adapter changes touch, re-measured with the final binary on ONE box (tools/round_variants.sh ran before them)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() { name=$1; shift; python bench.py --steps 6 --warmup 2 --no_cpu_baseline "$@" 2>/dev/null | tail -1 > gpurun_out/r05_variantF_${name}.json; }
run full
run lora --lora
run qwen_vl_lora --model qwen_vl --lora
run internlm_xc2 --model internlm_xc2
run internlm_xc2_lora --model internlm_xc2 --lora
run internlm_xc2_t512_lora --model internlm_xc2 --text_len 512 --lora
echo done
