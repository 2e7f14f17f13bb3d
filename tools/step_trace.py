#!/usr/bin/env python3
"""Steady-state kernel time breakdown of bench.py from a rocprofv3 --kernel-trace CSV: only kernels launched after the
last warm-up optimizer step are counted (adamw_kernel marks the end of a step: 2 launches per step).
usage: step_trace.py <kernel_trace.csv> <warmup_steps> <timed_steps> [out.txt]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
warm, steps = int(sys.argv[2]), int(sys.argv[3])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel")]
per = len(adam) // (warm + steps)
lo = adam[warm * per - 1] + 1 if warm else 0
hi = adam[-1] + 1
sel = rows[lo:hi]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[r["Kernel_Name"][:72]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(v[1] for v in agg.values())
lines = [f"steady state: {steps} steps, wall {(t1 - t0) / 1e6 / steps:.1f} ms/step, kernel-busy {busy / 1e6 / steps:.1f} ms/step"]
for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    lines.append("%-72s %7.1f launches/step %8.2f ms/step %5.1f%%" % (k, n / steps, ns / 1e6 / steps, 100.0 * ns / busy))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 4:
    open(sys.argv[4], "w").write(out + "\n")
