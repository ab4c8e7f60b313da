#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes tools/pmc_bench.sh took of bench.py itself: per kernel, over the dispatches of
the TIMED region (the last `steps` process() calls), mean counter value per launch.
  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM), WRITE_SIZE as is, both
  in KB -> hbm_bytes_per_launch.
usage: tools/pmc_insitu.py <outdir of pmc_bench.sh> --steps K [--calls-per-step kernel=count ...] > profiles/rNN_pmc_bench.json"""
import argparse
import collections
import csv
import glob
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument("outdir")
ap.add_argument("--steps", type=int, required=True)
args = ap.parse_args()

PER_STEP = {"encoder_bwd_kernel": 3, "encoder_fwd_kernel": 24, "pc_deconv_fwd_kernel": 2, "pc_deconv_bwd_kernel": 1,
            "maze_step_kernel": 20, "gemm_split_tn_kernel": 8, "rmsprop_kernel": 1,
            # the NT kernel's instantiations are different kernels: big forward / dgrad products, the whole-kernel LSTM
            # step (4096-row rollout / bootstrap steps; 8192-row steps of the batched replay pass), the 4096-row fc, the
            # fused BPTT steps (4096 rows: base; 8192 rows: replay pass)
            "gemm_split_nt_kernel<128, 128, true, false, 0, 1, false>": 8,
            "gemm_split_nt_kernel<128, 128, true, false, 1, 2, true>": 22,
            "gemm_split_nt_kernel<128, 128, true, false, 1, 1, true>": 20,
            "gemm_split_nt_kernel<64, 64, true, true, 0, 4, false>": 22,
            "gemm_split_nt_kernel<64, 64, true, true, 2, 4, false>": 19,
            "gemm_split_nt_kernel<64, 64, true, true, 2, 2, false>": 19}


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return n if n.startswith("gemm_split_nt_kernel") else n.split("<")[0]


out = {"_how": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu-baseline --steps %d (tools/pmc_bench.sh; one "
               "pass per counter group); mean per launch over the launches of the timed process() calls; FETCH_SIZE "
               "doubled per MI355X_MICROARCH.md, WRITE_SIZE as is (KB)" % args.steps, "kernels": {}}
for p in sorted(glob.glob(os.path.join(args.outdir, "*", "*", "*counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if k not in PER_STEP:
            continue
        o = out["kernels"].setdefault(k, {})
        for c, v in cs.items():
            tail = v[-PER_STEP[k] * args.steps:]
            o[c] = sum(tail) / len(tail)
            o["launches_averaged"] = len(tail)
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
