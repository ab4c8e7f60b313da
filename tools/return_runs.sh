#!/bin/bash
# Return runs of one gpurun call: N device trainers side by side on the one GPU (each is launch-latency-bound at 8 actors,
# so they overlap), then the capped evaluation of every snapshot present.
# usage: tools/return_runs.sh <steps> <budget_s> <actors> <groups> <tag> <seed> [<seed> ...]
set -o pipefail
STEPS=$1; BUDGET=$2; ACTORS=$3; GROUPS_=$4; TAG=$5; shift 5
mkdir -p gpurun_out/return
pids=()
for s in "$@"; do
  d=gpurun_out/return/${TAG}_s$s
  mkdir -p "$d"
  [ -d "runs/return/${TAG}_s$s" ] && cp -n runs/return/${TAG}_s$s/* "$d"/ 2>/dev/null     # resume state of an earlier call
  python3 tools/return_device.py --seed "$s" --actors "$ACTORS" --groups "$GROUPS_" --steps "$STEPS" --budget-s "$BUDGET" \
      --checkpoints "${CKPTS:-250000,500000,1000000}" --resume --out "$d" > "$d/stdout.log" 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
tail -n 2 gpurun_out/return/${TAG}_s*/stdout.log
# evaluate the snapshots here (gpurun_out travels back with <= 64 MiB: the 7.6 MB snapshots do not), then drop them
dirs=(); for s in "$@"; do dirs+=(gpurun_out/return/${TAG}_s$s); done
python3 tools/return_eval.py --cap "${CAP:-2000}" "${dirs[@]}" >> gpurun_out/return/eval_${TAG}.jsonl 2> gpurun_out/return/eval_${TAG}.err || { rc=$?; tail -n 20 gpurun_out/return/eval_${TAG}.err; }
rm -f gpurun_out/return/${TAG}_s*/ckpt-*.npz
exit $rc
