#!/bin/bash
# the bench variants quoted in DESIGN.md section 6, one JSON line each under gpurun_out/variants/: tools/bench_variants.sh
cd /root/repo
mkdir -p gpurun_out/variants
run() { name=$1; shift; python bench.py "$@" > gpurun_out/variants/$name.json 2> gpurun_out/variants/$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/variants/$name.json")); print("$name", d["ms_per_step"], "ms", d["value"], d["unit"])
except Exception as e:
    print("$name FAILED", e)
PY
}
run llava_lora --lora
run llava_next --model llava_next --pairs 2 --text_len 2048
run qwen_vl_lora --model qwen_vl --lora
run qwen_vl_full_finetune_lm --model qwen_vl
run internlm_xc2_full --model internlm_xc2 --text_len 512
run internlm_xc2_lora --model internlm_xc2 --lora --text_len 512
