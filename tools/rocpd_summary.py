#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.2, rocpd sqlite output) results .db into the `--stats`-style per-kernel text table that is
committed under profiles/.   usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    if len(name) > 90:
        name = name[:87] + "..."
    return name


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"{'kernel':<92} {'calls':>7} {'total_ms':>11} {'avg_us':>10} {'%':>7}"]
    for n, c, tot, avg, pct in rows:
        lines.append(f"{short(n):<92} {c:>7} {tot / 1e3:>11.3f} {avg:>10.2f} {pct:>7.2f}")
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(txt)
    print(txt)


if __name__ == "__main__":
    main()
