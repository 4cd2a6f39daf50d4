#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd SQLite): total, and the largest gaps with their neighbours."""
import re
import sqlite3
import sys


def short(n):
    return re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n))[:60]


c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy_end = rows[0][2]
gaps = []
idle = 0
for (n, s, e), (pn, ps, pe) in zip(rows[1:], rows[:-1]):
    if s > busy_end:
        g = s - busy_end
        idle += g
        gaps.append((g, short(pn), short(n), s - t0))
    busy_end = max(busy_end, e)
print(f"span {(t1 - t0) / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms ({100 * idle / (t1 - t0):.1f} %), {len(rows)} kernels")
import collections
hist = collections.Counter()
for g, *_ in gaps:
    hist[("<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms")] += g
print({k: round(v / 1e6, 2) for k, v in hist.items()}, "ms by gap size")
pair = collections.Counter()
for g, a, b, _ in gaps:
    if g >= 2e4:
        pair[(a, b)] += g
for (a, b), v in pair.most_common(15):
    print(f"{v / 1e6:8.2f} ms  {a}  ->  {b}")
