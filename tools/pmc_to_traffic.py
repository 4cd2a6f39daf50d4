#!/usr/bin/env python
"""Turn the text tools/prof_round3.sh writes (gpurun_out/r03_gemm_pmc.txt: one rocpd_pmc.py table per shape and counter group) into
profiles/r03_gemm_traffic.json, the file bench.py reads `roofline.traffic` from (the newest profiles/rNN_gemm_traffic.json).

    python tools/pmc_to_traffic.py gpurun_out/r03_gemm_pmc.txt > profiles/r03_gemm_traffic.json
"""
import json
import re
import sys

LABEL = {"fc1": "fc1 (+bias+GELU)", "fc1_ln": "fc1 (+bias+GELU)", "qkv": "qkv (+bias)", "qkv_ln": "qkv (+bias)", "fc2": "fc2 (+bias+residual)",
         "fc2_st": "fc2 (+bias+residual)", "proj": "proj (+bias+residual)", "proj_st": "proj (+bias+residual)"}
SHAPES = {"fc1": (6144, 1408), "qkv": (4224, 1408), "fc2": (1408, 6144), "proj": (1408, 1408)}
M = 279616
cur, kern, vals = None, {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+) ::", line)
    if m:
        cur = m.group(1)
        vals.setdefault(cur, {})
        continue
    m = re.match(r"## (.*?)\s+\(n=(\d+)\)", line)
    if m and cur:
        kern[cur] = (m.group(1), int(m.group(2)))
        continue
    m = re.match(r"\s+(\S+)\s+([-0-9.e+]+)\s*$", line)
    if m and cur and not (m.group(1) == "_dur_us" and "_dur_us" in vals[cur]):  # duration of the first (SQ / GRBM) pass
        vals[cur][m.group(1)] = float(m.group(2))
out = {"_comment": "HBM/fabric traffic and MFMA-busy counters of the ViT GEMM kernels the bench runs (the LayerNorm-folded blocks: fc1 / qkv = "
                   "consumer kernels gemm_pp4_kernel<EPI,false,1>, proj / fc2 = statistics producers gemm_pp4_kernel<0,false,2>), round 3: "
                   "rocprofv3 --pmc, one counter group per pass with --kernel-trace only; PROBE_M=279616 tools/gemm_probe.py 0 <shape> 1 = the "
                   "bench launch shape, random bf16 operands; tools/prof_round3.sh; raw: profiles/r03_gemm_pmc.txt.  bytes = (2*FETCH_SIZE + "
                   "WRITE_SIZE)*1024: FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced stream).  These are "
                   "L2 <-> fabric bytes: re-reads of an A / W panel by another XCD or a later tile batch that the 256-MB Infinity Cache serves "
                   "are counted, so the figure is an UPPER bound of the HBM bytes (DESIGN 3c).  fc2 is one launch of all rows (per-tile A descriptors, round 3).  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                   "1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); eff_clock_ghz = GRBM_GUI_ACTIVE / 8 / duration."}
for name, v in vals.items():
    base = name.split("_")[0]
    n, k = SHAPES[base]
    rows = M
    e = {"kernel": kern.get(name, ("?", 0))[0],
         "fetch_kb": v.get("FETCH_SIZE"), "write_kb": v.get("WRITE_SIZE"), "algorithmic_bytes": 2 * (rows * k + n * k + rows * n * (2 if base in ("fc2", "proj") else 1)),
         "tcc_hit": v.get("TCC_HIT_sum"), "tcc_miss": v.get("TCC_MISS_sum"), "mfma_busy_cycles": v.get("SQ_VALU_MFMA_BUSY_CYCLES"),
         "grbm_gui_active": v.get("GRBM_GUI_ACTIVE"), "dur_us": v.get("_dur_us")}
    if e["mfma_busy_cycles"] and e["grbm_gui_active"]:
        e["mfma_busy_frac"] = round(e["mfma_busy_cycles"] / 1024 / (e["grbm_gui_active"] / 8), 4)
        e["eff_clock_ghz"] = round(e["grbm_gui_active"] / 8 / (e["dur_us"] * 1e3), 3)
    out[LABEL[name]] = e
print(json.dumps(out, indent=1))
