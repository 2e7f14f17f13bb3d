#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call18; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -k "dropout_acc or row_set or lora" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for lib in vl-rlhf_amd/libvlr_hip_dx1.so vl-rlhf_amd/libvlr_hip.so; do
  echo "== $lib" >> $O/microbench.txt
  VLR_LIB=$lib timeout 300 python tools/lora_gemm_bench.py --iters 20 2>/dev/null | grep "multi" >> $O/microbench.txt
  VLR_LIB=$lib timeout 300 python tools/lora_gemm_bench.py --shape internlm --iters 20 2>/dev/null | grep "multi" >> $O/microbench.txt
done
cat $O/microbench.txt
AB_ARGS="--lora" bash tools/ab_bench.sh $O/lora vl-rlhf_amd/libvlr_hip_dx1.so default
