#!/bin/bash
# Counters of ONE kernel family from the micro-benchmarks (GPU box): tools/pmc_kernel.sh <outdir> <bench_kernels filter> <kernel substring>
# Two passes (SQ slots: 8): (a) matrix-pipe busy + wave state, (b) instruction mix + LDS.
set -o pipefail
OUT=$(realpath -m "$1"); FILT=$2; KSUB=$3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/a" -- python3 "$ROOT/tools/bench_kernels.py" "$FILT" > "$OUT/a.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$OUT/b" -- python3 "$ROOT/tools/bench_kernels.py" "$FILT" > "$OUT/b.log" 2>&1 || exit 1
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, json, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for p in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
res["launches"] = max(len(v) for v in acc.values()) if acc else 0
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
rm -rf "$OUT/a" "$OUT/b"
