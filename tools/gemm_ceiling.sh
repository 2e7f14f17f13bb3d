#!/bin/bash
# Evidence for the clock/power ceiling of bf16 MFMA work on this box:
#   (1) an MFMA-only register loop (tools/mfma_peak.hip), zero vs random operands;
#   (2) the production 8-phase GEMM at 8192^3 and at the decoder shape, plain and with parts ablated (VLR_GEMM_ABLATE:
#       1 no LDS-DMA in the K loop, 4 no barriers, 5 both, 8 no epilogue - timing only, results wrong);
#   (3) one PMC pass of the production kernel: busy cycles, MFMA-busy cycles, GRBM_GUI_ACTIVE (effective clock = cycles / wall).
# usage: bash tools/gemm_ceiling.sh > gpurun_out/r02_gemm_ceiling.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && /tmp/mfma_peak
for shape in "0 8192 8192 8192" "0 12288 12288 4096" "1 12288 4096 22016" "2 4096 11008 12792"; do
  for abl in 0 1 4 5 8; do
    [ "$abl" != 0 ] && [ "${shape:0:1}" != 0 ] && continue
    VLR_GEMM_ABLATE=$abl python tools/gemm_time.py $shape
  done
done
mkdir -p gpurun_out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/gemm_ceiling_pmc -o r -f csv -- python tools/gemm_time.py 0 12288 12288 4096 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/gemm_ceiling_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256p" not in r["Kernel_Name"]: continue
        acc["gemm256p NT 12288x12288x4096"][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for f in glob.glob("gpurun_out/gemm_ceiling_pmc/**/*kernel_trace.csv", recursive=True)
       for r in csv.DictReader(open(f)) if "gemm256p" in r["Kernel_Name"]]
for k, d in acc.items():
    print("PMC", k, "(profiled pass: avg launch %.1f us)" % (sum(dur) / len(dur) / 1e3))
    for c, v in d.items(): print("   %-28s %.5g per launch" % (c, v / n[c]))
    if "GRBM_GUI_ACTIVE" in d and dur:
        print("   effective clock = GRBM_GUI_ACTIVE / wall = %.2f GHz" % (d["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"] / (sum(dur) / len(dur))))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        print("   MFMA busy / (busy cycles x 1024 SIMDs / #SE-units) see DESIGN.md; raw ratio MFMA_BUSY / BUSY = %.1f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]))
PY
