#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call8; mkdir -p $O
rm -f $O/margins.txt
VLR_MARGINS=$O/margins.txt timeout 900 python -m pytest tests/test_hip_e2e.py tests/test_hip_llavanext.py tests/test_hip_qwenvl.py tests/test_hip_internlm.py -x -q -m gpu > $O/pytest_margins.txt 2>&1
timeout 600 python -m pytest tests/test_hip_fullsize_qwen_internlm.py -x -q -m gpu -k "lora_properties" > $O/pytest_fullsize_lora.txt 2>&1
echo call8 done
