#!/bin/bash
# the variant bench lines quoted in DESIGN.md section 6 (one JSON line each): tools/round_variants.sh r06 [name ...]   (names: a subset to run)
tag=${1:-r06}; shift
only=" $* "
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() { name=$1; shift; [ "$only" != "  " ] && [[ "$only" != *" $name "* ]] && return; python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_variants "$@" 2>/dev/null | tail -1 > gpurun_out/${tag}_variant_${name}.json; }
run full
run lora --lora
run precomputed_ref --precomputed_ref
run ckpt --gradient_checkpointing
run llava_next_4x2048_ckpt --model llava_next --pairs 4 --text_len 2048 --gradient_checkpointing
run llava_next_2x1024 --model llava_next --pairs 2
run qwen_vl --model qwen_vl
run qwen_vl_lora --model qwen_vl --lora
run internlm_xc2 --model internlm_xc2
run internlm_xc2_lora --model internlm_xc2 --lora
run internlm_xc2_t512 --model internlm_xc2 --text_len 512
run internlm_xc2_t512_lora --model internlm_xc2 --text_len 512 --lora
echo variants done
