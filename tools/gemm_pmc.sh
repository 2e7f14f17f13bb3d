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
# third pass: effective clock and matrix-pipe occupancy from the graphics-engine cycle counter and the kernel trace's own timestamps
# (GRBM_GUI_ACTIVE is summed over the 8 XCDs: profiles/r02_gemm_ceiling_mfma_only_and_ablation.txt, 1.3017e7 per 994.4 us launch = 8 x 1.64 GHz)
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d gpurun_out/pmc_GRBM -o r -f csv -- python tools/gemm_time.py "$@" > gpurun_out/pmc_GRBM.log 2>&1 || { echo "gemm_pmc.sh: profiled run failed, see gpurun_out/pmc_GRBM.log" >&2; tail -5 gpurun_out/pmc_GRBM.log >&2; exit 1; }
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_GRBM/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256p" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for f in glob.glob("gpurun_out/pmc_GRBM/**/*kernel_trace.csv", recursive=True)
       for r in csv.DictReader(open(f)) if "gemm256p" in r["Kernel_Name"]]
if not dur or not n["GRBM_GUI_ACTIVE"]:
    raise SystemExit("gemm_pmc.sh: no gemm256p rows in the GRBM pass")
g = acc["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"]; b = acc["SQ_BUSY_CYCLES"] / n["SQ_BUSY_CYCLES"]; m = acc["SQ_VALU_MFMA_BUSY_CYCLES"] / n["SQ_VALU_MFMA_BUSY_CYCLES"]
d = sum(dur) / len(dur)
print("clock pass (%d launches, avg %.1f us between the trace's start and end timestamps, profiled)" % (len(dur), d / 1e3))
print("   GRBM_GUI_ACTIVE %.5g  SQ_BUSY_CYCLES %.5g  SQ_VALU_MFMA_BUSY_CYCLES %.5g per launch" % (g, b, m))
print("   effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration        = %.3f GHz" % (g / 8 / d))
print("   SQ_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8)                       = %.2f  (32 if it is summed over 32 always-busy units)" % (b / (g / 8)))
print("   matrix pipe busy = MFMA_BUSY / (1024 SIMDs x GRBM / 8)       = %.3f" % (m / (1024 * g / 8)))
print("   matrix pipe busy = MFMA_BUSY / (32 x SQ_BUSY_CYCLES)         = %.3f" % (m / (32 * b)))
PY
