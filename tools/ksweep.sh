cd /root/repo
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
for P in 0 1; do
for K in 1024 4096; do VLR_GEMM_PERSIST=$P python tools/gemm_time.py 0 12288 12288 $K 2>&1 | tail -1 | sed "s/^/PERSIST=$P /"; done
for args in "0 12792 12288 4096" "0 12792 22016 4096" "0 12792 4096 11008" "1 12792 4096 12288" "1 12792 11008 4096" "1 12792 4096 22016" "2 12288 4096 12792" "2 4096 11008 12792" "2 22016 4096 12792"; do
  VLR_GEMM_PERSIST=$P python tools/gemm_time.py $args 2>&1 | tail -1 | sed "s/^/PERSIST=$P /"
done; done
