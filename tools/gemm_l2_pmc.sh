#!/bin/bash
# L2 / fabric counters of the persistent 256x256 GEMM on the twelve GEMM shapes of the 7B layer (tools/gemm_sched_bench.py, one call per
# shape after one warm-up call), two rocprofv3 --pmc passes (TCC: 4 slots; SQ / GRBM in their own pass):
#   pass 1  TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum   -> L2 hit rate = HIT / (HIT + MISS), fabric requests per launch
#   pass 2  GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
# The ideal hit rate of the 8 x 4 tile footprint of an XCD (32 workgroups share 8 A panels and 4 B panels in its 4 MB L2) is
# 1 - 12 / 64 = 81 %: every panel line is fetched from the fabric once and hit by the other 3 (A) / 7 (B) workgroups.
# MODES=0 bash tools/gemm_l2_pmc.sh: the tile map of rounds 1-4 (default 32: the shared-panel map).
# Writes gpurun_out/gemm_l2_pmc.txt; copy to profiles/ to have it judged.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/pmc_l2_1 /tmp/pmc_l2_2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -d /tmp/pmc_l2_1 -o r -f csv -- python tools/gemm_sched_bench.py --rounds 1 --iters 1 --modes ${MODES:-32} > /tmp/pmc_l2_1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pmc_l2_2 -o r -f csv -- python tools/gemm_sched_bench.py --rounds 1 --iters 1 --modes ${MODES:-32} > /tmp/pmc_l2_2.log 2>&1
python - <<'PY'
import csv, glob, collections
NAMES = ["qkv+rope NT [M,12288,4096]", "o_proj f32res NT [M,4096,4096]", "swiglu NT [M,22016,4096]", "down f32res NT [M,4096,11008]", "dgrad qkv NN [M,4096,12288]",
         "dgrad gu NN [M,4096,22016]", "swiglu-bwd NN [M,11008,4096]", "dattn NN [M,4096,4096]", "wgrad qkv TN [12288,4096,M]", "wgrad o TN [4096,4096,M]",
         "wgrad gu TN [22016,4096,M]", "wgrad down TN [4096,11008,M]", "wgrad gu + down, one launch"]
def load(d):
    fs = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    rows = collections.OrderedDict()
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "gemm256p_kernel" not in r["Kernel_Name"]:
                continue
            did = int(r["Dispatch_Id"])
            e = rows.setdefault(did, {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "grid": int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # full-grid launches only (the fused tail launches of peeled rows run fewer workgroups), in dispatch order; the LAST 13 = the timed calls
    full = [v for k, v in sorted(rows.items()) if v["grid"] >= 240 * 512]
    return full[-len(NAMES):]
a, b = load("/tmp/pmc_l2_1"), load("/tmp/pmc_l2_2")
out = ["tools/gemm_l2_pmc.sh: rocprofv3 --pmc, one launch of the persistent 256x256 kernel per GEMM shape of the 7B layer (M = 12792)",
       "L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS); EA rd / wr = requests the XCD L2s sent to the fabric (Infinity Cache / HBM side) per launch;",
       "clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time is not derived here (no duration in the counter file): MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)", "",
       "%-34s %-44s %8s %12s %12s %10s %10s %10s" % ("shape", "kernel", "L2 hit", "EA rd req", "EA wr req", "MFMA busy", "wait any", "wait inst")]
for i, nm in enumerate(NAMES):
    if i >= len(a) or i >= len(b):
        break
    x, y = a[i], b[i]
    hit = x.get("TCC_HIT_sum", 0) / max(1.0, x.get("TCC_HIT_sum", 0) + x.get("TCC_MISS_sum", 0))
    busy = y.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, 4 * y.get("SQ_BUSY_CU_CYCLES", 0))
    wc = max(1.0, y.get("SQ_WAVE_CYCLES", 0))
    out.append("%-34s %-44s %7.1f%% %12.3g %12.3g %9.1f%% %9.1f%% %9.1f%%" % (nm, x["name"][:44], 100 * hit, x.get("TCC_EA0_RDREQ_sum", 0), x.get("TCC_EA0_WRREQ_sum", 0),
               100 * busy, 100 * y.get("SQ_WAIT_ANY", 0) / wc, 100 * y.get("SQ_WAIT_INST_ANY", 0) / wc))
open("gpurun_out/gemm_l2_pmc.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
