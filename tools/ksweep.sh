cd /root/repo
for F in 0 256 512 1024 2048; do
for args in "0 12288 12288 1024" "0 12792 12288 4096" "1 12792 4096 12288"; do
  VLR_GEMM_FLAGS=$F python tools/gemm_time.py $args 2>&1 | tail -1 | sed "s/^/FLAGS=$F /"
done; done
