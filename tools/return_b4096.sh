#!/bin/bash
# (c) of the return table: the batched learner at 4096 actors, one update per call (G = 1) and eight (G = 8), to 10 M env
# steps, one after the other (each needs the 173 GB ring), snapshots at 1 / 5 / 10 M evaluated in the same call.
set -o pipefail
mkdir -p gpurun_out/return
rc=0
for G in 1 8; do
  d=gpurun_out/return/dev_b4096_g$G
  mkdir -p "$d"
  python3 tools/return_device.py --seed 0 --actors 4096 --groups $G --steps 1e7 --checkpoints 1000000,5000000,10000000 \
      --log-every 10 --out "$d" > "$d/stdout.log" 2>&1 || rc=$?
  tail -n 1 "$d/stdout.log"
done
python3 tools/return_eval.py --cap "${CAP:-2000}" gpurun_out/return/dev_b4096_g1 gpurun_out/return/dev_b4096_g8 \
    >> gpurun_out/return/eval_dev_b4096.jsonl 2> gpurun_out/return/eval_dev_b4096.err || { rc=$?; tail -n 20 gpurun_out/return/eval_dev_b4096.err; }
rm -f gpurun_out/return/dev_b4096_g*/ckpt-*.npz gpurun_out/return/dev_b4096_g*/state.pt
exit $rc
