#!/bin/bash
# In-situ counters of bench.py (GPU box): one rocprofv3 pass per counter group, as MI355X_MICROARCH.md prescribes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/hip/hsa tracing).
# usage: tools/pmc_bench.sh <outdir> [bench.py args...]       e.g. tools/pmc_bench.sh gpurun_out/pmc --steps 2 --warmup 1
set -o pipefail
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export UNREAL_BENCH_SIDECAR="$OUT/bench_run.json"      # bench.py records what it ran (actors, groups, frames per launch)
run() {   # name, counters...
  local name=$1; shift
  export UNREAL_SAVE_MAPS="$OUT/$name.maps.txt"        # the process's DSO map: a crash under the profiler can then be symbolised
  timeout -k 10 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- \
      python3 "$ROOT/bench.py" --no-cpu-baseline "${ARGS[@]}" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "pass $name rc=$rc" | tee -a "$OUT/passes.log"
  return $rc
}
ARGS=("$@")
run fetch FETCH_SIZE && run write WRITE_SIZE && run mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE && \
run waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
rc=$?
# raw per-dispatch CSVs are tens of MB per pass: keep the per-kernel summary only
python3 "$ROOT/tools/pmc_insitu.py" "$OUT" > "$OUT/pmc_bench.json" && rm -rf "$OUT"/fetch "$OUT"/write "$OUT"/mfma "$OUT"/waves
exit $rc
