#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call16; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lora -o r -- python bench.py --steps 4 --warmup 2 --no_cpu_baseline --lora > $O/bench_lora_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof_lora -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py $f 2 4 $O/r05_steady_state_kernel_breakdown_lora.txt > /dev/null
python - "$f" > $O/lora_small_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel")]
sel = rows[adam[-2] + 1: adam[-1] + 1]          # the last step
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r["Kernel_Name"]
    if "gemm128p" in n or "splitk" in n or "lora_dx" in n or "dropout" in n or "gemm_bf16_kernel" in n:
        key = (n[:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-42s grid %-8s %-4s %-3s  x%4d  %9.1f us total  %7.1f us each" % (k[0], k[1], k[2], k[3], n, us, us / n))
PY
rm -rf /tmp/prof_lora
echo done
