#!/bin/bash
# PMC profile of the decoder attention kernels at the C2 shape: tools/attn_pmc.sh   (results under gpurun_out/attn_pmc_*)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/attn_pmc_$tag -o r -f csv -- python tools/attn_time.py 3 > gpurun_out/attn_pmc_$tag.log 2>&1
  python - <<PY
import csv,glob,collections
fs=glob.glob("gpurun_out/attn_pmc_$tag/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:48]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,d in acc.items():
    if "attn" not in k: continue
    print(k)
    for c,v in d.items(): print("   %-34s %.5g per launch"%(c, v/max(1,n[(k,c)])))
PY
done
