#!/bin/bash
# Throughput-vs-equivalence frontier on the device: for each configuration "B:G:steps:checkpoints" the
# seeds are trained side by side on the one GPU (a run at <= 512 actors is launch-latency-bound; 4096-actor runs -- 173 GB of
# ring -- one after the other), every snapshot is evaluated by the capped-return evaluator (tools/return_eval.py: sampled
# policy, 256 episodes, 2000-step cap), snapshots are dropped (gpurun_out travels back with <= 64 MiB).
# usage: SEEDS="0 1 2" tools/return_frontier.sh <tag> <budget_s per configuration> B:G:steps:ckpt,ckpt ...
set -o pipefail
TAG=$1; BUDGET=$2; shift 2
SEEDS=${SEEDS:-"0 1 2"}
mkdir -p gpurun_out/return
STAMP=$(date +%m%d_%H%M%S)        # one evaluation file per call: gpurun merges a call's gpurun_out/ OVER the local one
rc=0
for cfg in "$@"; do
  IFS=: read -r B G STEPS CKPTS <<< "$cfg"
  dirs=()
  pids=()
  for s in $SEEDS; do
    d=gpurun_out/return/${TAG}_b${B}_g${G}_s$s
    mkdir -p "$d"; dirs+=("$d")
    [ -d "runs/return/${TAG}_b${B}_g${G}_s$s" ] && cp -n runs/return/${TAG}_b${B}_g${G}_s$s/* "$d"/ 2>/dev/null   # resume state of an earlier call
    cmd=(python3 tools/return_device.py --seed "$s" --actors "$B" --groups "$G" --steps "$STEPS" --budget-s "$BUDGET"
         --checkpoints "$CKPTS" --resume --log-every "${LOG_EVERY:-20}" --out "$d")
    if [ "$B" -ge 2048 ]; then                      # one ring at a time
      "${cmd[@]}" > "$d/stdout.log" 2>&1 || rc=$?
    else
      "${cmd[@]}" > "$d/stdout.log" 2>&1 &
      pids+=($!)
    fi
  done
  for p in "${pids[@]}"; do wait "$p" || rc=$?; done
  tail -q -n 1 gpurun_out/return/${TAG}_b${B}_g${G}_s*/stdout.log
  python3 tools/return_eval.py --cap "${CAP:-2000}" "${dirs[@]}" >> gpurun_out/return/eval_${TAG}_${STAMP}.jsonl 2>> gpurun_out/return/eval_${TAG}_${STAMP}.err || { rc=$?; tail -n 5 gpurun_out/return/eval_${TAG}_${STAMP}.err; }
  rm -f gpurun_out/return/${TAG}_b${B}_g${G}_s*/ckpt-*.npz
  # resume states are 15 MB each and gpurun refuses to copy back more than 64 MiB in total: keep them only when asked to
  [ "${KEEP_STATE:-0}" = "1" ] || rm -f gpurun_out/return/${TAG}_b${B}_g${G}_s*/state.pt
  echo "[frontier] $cfg done rc=$rc at $(date +%T)"
done
exit $rc
