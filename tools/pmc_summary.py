#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one row per dispatch and counter) per kernel: mean counter value per launch.
usage: tools/pmc_summary.py <counter_collection.csv> [...]  -> JSON on stdout"""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    if not any(s in k for s in ("encoder", "gemm_split", "pc_deconv", "maze_step", "gemm_kernel")):
        continue
    # drop warm-up outliers: use the median launch
    out[k] = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    out[k]["launches"] = max(len(v) for v in cs.values())
    o = out[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "GRBM_GUI_ACTIVE" in o and o["GRBM_GUI_ACTIVE"] > 0:
        # MFMA-busy SIMD-cycles over all SIMD-cycles of the dispatch (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs)
        o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
print(json.dumps(out, indent=1, sort_keys=True))
