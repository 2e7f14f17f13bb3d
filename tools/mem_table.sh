#!/bin/bash
# peak device memory and step time per configuration: bash tools/mem_table.sh
cd "$(dirname "$0")/.."
run() { name=$1; shift; python bench.py --steps 2 --warmup 1 --no_cpu_baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['config']['peak_allocated_gib'])"; }
run default
run ckpt --gradient_checkpointing
run lora --lora
run lora_ckpt --lora --gradient_checkpointing
run precomputed_ref --precomputed_ref
run next_2x2048 --model llava_next --pairs 2 --text_len 2048
run next_4x2048_ckpt --model llava_next --pairs 4 --text_len 2048 --gradient_checkpointing
run qwen --model qwen_vl
run qwen_lora --model qwen_vl --lora
run ilm_1024 --model internlm_xc2
run ilm_1024_ckpt --model internlm_xc2 --gradient_checkpointing
run ilm_lora --model internlm_xc2 --lora
