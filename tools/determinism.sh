cd /root/repo
run() { python bench.py --steps 3 --warmup 0 --no_cpu_baseline $@ 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['loss'])"; }
echo "default:        $(run) $(run)"
echo "no_side_stream: $(run --no_side_stream) $(run --no_side_stream)"
export VLR_GEMM_CONT=0
echo "CONT=0 noside:  $(run --no_side_stream) $(run --no_side_stream)"
export VLR_GEMM_PERSIST=0
echo "+PERSIST=0:     $(run --no_side_stream) $(run --no_side_stream)"
export VLR_ASYNC_OPT=0
echo "+ASYNC_OPT=0:   $(run --no_side_stream) $(run --no_side_stream)"
