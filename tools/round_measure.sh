#!/bin/bash
# the measurements quoted in DESIGN.md section 6 for one round: tools/round_measure.sh r04   (writes gpurun_out/<tag>_*; copy to profiles/)
# Order: the PMC passes FIRST (they record the source digest of the library they ran with), their JSON is put where bench.py looks for it,
# so the default-command line quotes roofline.traffic of the SAME binary - bench.py refuses a file taken with another digest.
tag=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
# 1. HBM / fabric traffic (PMC, separate passes)
bash tools/pmc_traffic.sh > /dev/null 2>&1
cp gpurun_out/pmc_hbm_traffic.txt gpurun_out/${tag}_pmc_hbm_traffic.txt
cp gpurun_out/pmc_hbm_traffic.json gpurun_out/${tag}_pmc_hbm_traffic.json
cp gpurun_out/pmc_hbm_traffic.json profiles/${tag}_pmc_hbm_traffic.json
# 2. the default command, as the driver runs it
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_1gpu_default_cmd.json 2> gpurun_out/${tag}_bench.err
# 3. rocprofv3 kernel stats + steady-state breakdown of the same command (shorter run)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag} -o r -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_variants > gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof_${tag} -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py $f 2 5 gpurun_out/${tag}_steady_state_kernel_breakdown_final.txt > /dev/null
cp $(find /tmp/prof_${tag} -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_rocprofv3_kernel_stats_bench.csv
rm -rf /tmp/prof_${tag}
echo measured
