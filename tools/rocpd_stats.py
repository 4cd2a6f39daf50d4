#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (markdown/CSV).

    python tools/rocpd_stats.py gpurun_out/prof1/r1_results.db > profiles/r01_bench_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, (end - start) from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0, 1e30, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# kernel stats from {path}\n\ntotal kernel time: {total/1e6:.2f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{n}` | {a[0]} | {a[1]/1e6:.2f} | {a[1]/a[0]/1e3:.1f} | {a[2]/1e3:.1f} | {a[3]/1e3:.1f} | {100*a[1]/total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
