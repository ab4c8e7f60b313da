#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes tools/pmc_bench.sh took of bench.py itself: per kernel, over the dispatches of
the TIMED region, mean counter value per launch.
  The timed region is found in the trace itself, not from constants: every learner update ends with exactly one
  `rmsprop_kernel` launch, bench.py makes `warmup` untimed and `steps` timed process() calls of `groups` updates each
  (bench_run.json, written by bench.py under UNREAL_BENCH_SIDECAR), so the timed dispatches are those after update number
  warmup*groups and up to the last update -- whatever the schedule (batch_aux, groups, fuse_bptt) launches inside.
  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM), WRITE_SIZE as is, both
  in KB -> hbm_bytes_per_launch.
usage: tools/pmc_insitu.py <outdir of pmc_bench.sh> > profiles/rNN_pmc_bench.json"""
import argparse
import collections
import csv
import glob
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument("outdir")
args = ap.parse_args()
run = json.load(open(os.path.join(args.outdir, "bench_run.json")))
DELIM = "rmsprop_kernel"


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return n if n.startswith("gemm_split_nt_kernel") else n.split("<")[0]


out = {"_how": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu-baseline --steps %d --warmup %d "
               "(tools/pmc_bench.sh; one pass per counter group); mean per launch over the launches of the timed "
               "process() calls (found by their rmsprop_kernel delimiters); FETCH_SIZE doubled per MI355X_MICROARCH.md, "
               "WRITE_SIZE as is (KB)" % (run["steps"], run["warmup"]),
       "profiled_run": run, "kernels": {}}
for p in sorted(glob.glob(os.path.join(args.outdir, "*", "*", "*counter_collection.csv"))):
    rows = list(csv.DictReader(open(p)))
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if key:
        rows.sort(key=lambda r: int(r[key]))
    # dispatch ids of the update delimiters (one row per counter per dispatch: de-duplicate)
    seen, delims = set(), []
    for i, r in enumerate(rows):
        if short(r["Kernel_Name"]) == DELIM:
            d = r[key] if key else i
            if d not in seen:
                seen.add(d)
                delims.append(int(d) if key else i)
    n_timed = run["steps"] * run["groups"]
    if len(delims) < n_timed + 1:
        raise SystemExit("%s: %d updates in the trace, need more than %d" % (p, len(delims), n_timed))
    lo, hi = delims[-n_timed - 1], delims[-1]                  # (after the last untimed update, last timed update]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for i, r in enumerate(rows):
        d = int(r[key]) if key else i
        if lo < d <= hi:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        o = out["kernels"].setdefault(k, {})
        for c, v in cs.items():
            o[c] = sum(v) / len(v)
            o["launches_averaged"] = len(v)
            o["launches_per_call"] = len(v) / float(run["steps"])
for k, o in out["kernels"].items():
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_bytes_per_launch"] = (2.0 * o["FETCH_SIZE"] + o["WRITE_SIZE"]) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and o.get("GRBM_GUI_ACTIVE", 0) > 0:
        o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if o.get("SQ_WAVE_CYCLES", 0) > 0:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in o:
                o[c.lower() + "_share"] = o[c] / o["SQ_WAVE_CYCLES"]
print(json.dumps(out, indent=1, sort_keys=True))
