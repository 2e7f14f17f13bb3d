#!/bin/bash
# GEMM sweep over the decoder shapes of the 7B step (M = 12792 token rows): correctness tests, then TF/s per layout/shape.
# Environment switches of the dispatcher: VLR_GEMM_8PHASE (bit mask NT|NN|TN, 0 = 128x128 kernel only), VLR_GEMM_CONT=0,
# VLR_GEMM_PERSIST=0, VLR_GEMM_SPLIT=0 (no peeling), VLR_GEMM_ABLATE (NT timing ablations: 1 no DMA, 4 no barriers, 8 no epilogue).
cd "$(dirname "$0")/.."
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
for args in "0 8192 8192 8192" "0 12792 12288 4096" "0 12792 22016 4096" "0 12792 4096 11008" "1 12792 4096 12288" "1 12792 11008 4096" "1 12792 4096 22016" "2 12288 4096 12792" "2 4096 11008 12792" "2 22016 4096 12792"; do
  python tools/gemm_time.py $args 2>&1 | tail -1
done
