#!/bin/bash
# PMC profile of one GEMM shape: tools/gemm_pmc.sh <layout> <M> <N> <K>   (results under gpurun_out/pmc_*)
# layout is the C-ABI's integer: 0 NT, 1 NN, 2 TN (tools/gemm_time.py)
case "$1" in 0|1|2) ;; *) echo "gemm_pmc.sh: layout must be 0 (NT), 1 (NN) or 2 (TN), got '$1'" >&2; exit 2;; esac
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_$tag -o r -f csv -- python tools/gemm_time.py "$@" > gpurun_out/pmc_$tag.log 2>&1 || { echo "gemm_pmc.sh: profiled run failed, see gpurun_out/pmc_$tag.log" >&2; tail -5 gpurun_out/pmc_$tag.log >&2; exit 1; }
  python - <<PY
import csv,glob,collections
fs=glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,d in acc.items():
    if "gemm" not in k: continue
    print(k)
    for c,v in d.items(): print("   %-34s %.4g per launch"%(c, v/max(1,n[(k,c)])))
PY
done
