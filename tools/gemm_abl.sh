cd /root/repo
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
for f in 0 1 3 4 7; do
  VLR_GEMM_ABLATE=$f python tools/gemm_time.py 0 8192 8192 8192 2>&1 | tail -1
done
for args in "0 4096 4096 4096" "0 12792 12288 4096" "0 12792 22016 4096" "0 12792 4096 11008" "1 12792 4096 12288" "1 12792 11008 4096" "1 12792 4096 22016" "2 12288 4096 12792" "2 4096 11008 12792" "2 22016 4096 12792"; do
  python tools/gemm_time.py $args 2>&1 | tail -1
done
