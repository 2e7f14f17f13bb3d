cd /root/repo
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
for C in 0 1; do
for args in "0 12288 12288 1024" "0 12288 12288 4096" "0 12792 12288 4096" "0 12792 22016 4096" "1 12792 4096 12288" "1 12792 11008 4096" "1 12792 4096 22016" "2 12288 4096 12792" "2 4096 11008 12792" "2 22016 4096 12792"; do
  VLR_GEMM_CONT=$C python tools/gemm_time.py $args 2>&1 | tail -1 | sed "s/^/CONT=$C /"
done; done
