#!/usr/bin/env python
"""Per-kernel mean PMC counter values from a rocprofv3 rocpd database (counters summed over SEs/XCDs per dispatch)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name)[:70]


def main(path, filt=""):
    c = sqlite3.connect(path)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection").fetchall()
    per = defaultdict(lambda: defaultdict(float))
    dur = {}
    names = {}
    for did, kn, cn, v, d in rows:
        per[did][cn] += v
        dur[did] = d
        names[did] = short(kn)
    agg = defaultdict(lambda: defaultdict(list))
    for did, cs in per.items():
        if filt and filt not in names[did]:
            continue
        for cn, v in cs.items():
            agg[names[did]][cn].append(v)
        agg[names[did]]["_dur_us"].append(dur[did] / 1e3)
    for kn, cs in agg.items():
        print(f"## {kn}  (n={len(cs['_dur_us'])})")
        for cn in sorted(cs):
            v = cs[cn]
            print(f"   {cn:28s} {sum(v)/len(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
