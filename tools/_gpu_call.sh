timeout 300 python -m pytest tests/test_hip_kernels.py -k "attention" -x -q 2>&1 | tail -2
for i in 1 2 3; do
  for sh in "20" "10 4 4975 32 8"; do
    timeout 120 python tools/attn_time.py $sh 2>&1 | grep "attn bwd"
  done
done
