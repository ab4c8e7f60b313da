#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV over the timed Trainer.process() calls of bench.py
(segments are delimited by the rmsprop kernel, one launch per process()); replay fill excluded.
usage: tools/trace_breakdown.py <kernel_trace.csv> [title]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rms = [i for i, r in enumerate(rows) if 'rmsprop_kernel' in r['Kernel_Name']]
seg = rows[rms[0] + 1:rms[-1] + 1]
n = len(rms) - 1
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
tot = collections.defaultdict(lambda: [0, 0])
durs = collections.defaultdict(list)
busy = 0
for r in seg:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
    tot[k][0] += d
    tot[k][1] += 1
    durs[k].append(d)
    busy += d
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("# %d timed Trainer.process() calls (replay fill and first warm-up excluded)" % n)
print("# wall %.2f ms per process(), GPU busy %.2f ms per process()" % ((t1 - t0) / n / 1e6, busy / n / 1e6))
print("%-72s %12s %12s %10s %10s %10s" % ("kernel", "ms/process", "calls/proc", "avg_us", "median_us", "max_us"))
for k, (d, c) in sorted(tot.items(), key=lambda x: -x[1][0]):
    ds = sorted(durs[k])
    print("%-72s %12.3f %12.1f %10.1f %10.1f %10.1f" % (k, d / n / 1e6, c / n, d / c / 1e3, ds[len(ds) // 2] / 1e3, ds[-1] / 1e3))
