#!/usr/bin/env python3
"""Markdown tables of the capped-return evaluations (tools/return_eval.py lines) of the frontier / long runs.
usage: python tools/return_table.py eval_a.jsonl [eval_b.jsonl ...] [--logs DIR ...]
Rows are grouped by run family (name without the _s<seed> suffix) and snapshot; `--logs` directories hold
<run>/log.jsonl (training throughput: the last steps_per_s of each seed)."""
import glob
import json
import os
import re
import statistics as st
import sys

evals, logdirs = [], []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--logs":
        logdirs.append(args.pop(0))
    else:
        evals.append(a)
rows = []
for f in evals:
    for l in open(f):
        l = l.strip()
        if l:
            rows.append(json.loads(l))
fam = {}
for r in rows:
    m = re.match(r"(.*)_s(\d+)$", r["run"])
    name, seed = (m.group(1), int(m.group(2))) if m else (r["run"], 0)
    fam.setdefault(name, {}).setdefault(r["t"], {})[seed] = r
thr = {}
for d in logdirs:
    for lf in glob.glob(os.path.join(d, "*", "log.jsonl")):
        run = os.path.basename(os.path.dirname(lf))
        m = re.match(r"(.*)_s(\d+)$", run)
        name = m.group(1) if m else run
        sp = [json.loads(l).get("steps_per_s") for l in open(lf) if '"steps_per_s"' in l]
        sp = [x for x in sp if x]
        if sp:
            thr.setdefault(name, []).append(sp[-1])
print("| run | snapshot | capped return per seed | mean +- sigma | success rate | mean length | training env-steps/s per process |")
print("|---|---|---|---|---|---|---|")
for name in sorted(fam):
    for t in sorted(fam[name]):
        rs = [fam[name][t][s] for s in sorted(fam[name][t])]
        vals = [r["mean_return"] for r in rs]
        mean = sum(vals) / len(vals)
        sd = st.stdev(vals) if len(vals) > 1 else float("nan")
        print("| %s | %.2f M | %s | %.3f +- %.3f | %s | %s | %s |" % (
            name, t / 1e6, ", ".join("%.3g" % v for v in vals), mean, sd,
            " / ".join("%.2f" % r["success_rate"] for r in rs), " / ".join("%.0f" % r["mean_length"] for r in rs),
            ", ".join("%.0f" % x for x in thr.get(name, [])) or "-"))
