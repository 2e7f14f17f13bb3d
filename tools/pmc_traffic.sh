#!/bin/bash
# HBM traffic of every kernel of one DPO step: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (guide: TCC slots),
# bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB; gfx950 FETCH_SIZE counts half of a wide coalesced read - MI355X_MICROARCH.md, HBM).
# Writes gpurun_out/pmc_hbm_traffic.{txt,json}; copy them to profiles/ to have them judged.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o r -f csv -- python bench.py --steps 1 --warmup 0 --no_side_stream --no_cpu_baseline > /tmp/pmc_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, "vl-rlhf_amd")
import build_hip
def load(c):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"]); n[k] += 1
    return acc, n
fa, fn = load("FETCH_SIZE"); wa, wn = load("WRITE_SIZE")
rows = []
for k in fa:
    f = fa[k] / fn[k]; w = wa.get(k, 0.0) / max(1, wn.get(k, 1))
    rows.append((k, fn[k], f, w, (2 * f + w) * 1024 / 1e6))
rows.sort(key=lambda r: -r[1] * r[4])
out = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no_side_stream",
       "units: KiB per dispatch (averaged over the dispatches of a kernel); HBM bytes = 2*FETCH_SIZE + WRITE_SIZE", "",
       "%-62s %8s %19s %19s %15s" % ("kernel", "launches", "FETCH_SIZE avg KiB", "WRITE_SIZE avg KiB", "HBM MB/launch")]
for k, n, f, w, mb in rows[:40]:
    out.append("%-62s %8d %19.0f %19.0f %15.1f" % (k[:62], n, f, w, mb))
open("gpurun_out/pmc_hbm_traffic.txt", "w").write("\n".join(out) + "\n")
g = [r for r in rows if r[0].startswith("gemm256p_kernel")]
nl = sum(r[1] for r in g)
json.dump({"source_digest": build_hip.kernel_digest()[:16], "gemm_launches": nl, "gemm_hbm_bytes_per_launch": sum(r[1] * r[4] * 1e6 for r in g) / max(1, nl),
           "note": "2*FETCH_SIZE+WRITE_SIZE, KiB->bytes, averaged over the gemm256p (8-phase 256x256 tile) launches of one DPO step"},
          open("gpurun_out/pmc_hbm_traffic.json", "w"))
print("\n".join(out[:16]))
PY
