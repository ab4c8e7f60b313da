#!/bin/bash
# The round's profile set in one call on the GPU box: usage tools/round_profiles.sh rNN
#   1. rocprofv3 --kernel-trace --stats of the bench command   -> gpurun_out/<r>/bench_kernel_stats.csv + process breakdown
#   2. in-situ PMC passes of bench.py (tools/pmc_bench.sh)      -> gpurun_out/<r>/pmc/pmc_bench.json
#   3. kernel trace at --groups 8: wall vs GPU-busy per call (is a grouped call launch-bound?)
set -o pipefail
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/trace.log" 2>&1 || exit 1
cd "$ROOT"
f=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_breakdown.py "$f" "MI355X, bench.py --steps 5 --warmup 2 under rocprofv3 --kernel-trace --stats ($R)" > "$OUT/bench_process_breakdown.txt"
cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/bench_kernel_stats.csv"
rm -rf "$OUT/trace"
head -n 16 "$OUT/bench_process_breakdown.txt"
tools/pmc_bench.sh "$OUT/pmc" --steps 3 --warmup 1 || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_g8" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --groups 8 --no-cpu-baseline > "$OUT/trace_g8.log" 2>&1 || exit 1
cd "$ROOT"
f=$(find "$OUT/trace_g8" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_breakdown.py "$f" "MI355X, bench.py --groups 8 --steps 3 --warmup 1 under rocprofv3 --kernel-trace ($R)" > "$OUT/bench_g8_breakdown.txt"
rm -rf "$OUT/trace_g8"
head -n 5 "$OUT/bench_g8_breakdown.txt"
