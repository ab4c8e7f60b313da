#!/usr/bin/env python3
"""Train the maze agent with the batched Trainer and log episode returns (GPU box).

usage: python tools/train_maze.py [--actors 4096] [--groups 1] [--history 2000] [--steps 2e7] [--log-every 10]
                                  [--lr-scale 1.0] [--max-time-step 0] [--out curve.jsonl]
`--groups G`: G sequential updates per process() call (update density x G, see Trainer).  One JSON line per
`--log-every` calls: global_t, episodes finished since the last line, their mean return, losses, entropy, steps/s."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--actors", type=int, default=4096)
ap.add_argument("--groups", type=int, default=1)
ap.add_argument("--history", type=int, default=2000)
ap.add_argument("--steps", type=float, default=2e7)
ap.add_argument("--log-every", type=int, default=10)
ap.add_argument("--lr-scale", type=float, default=1.0)
ap.add_argument("--max-time-step", type=float, default=0)
ap.add_argument("--entropy-beta", type=float, default=None, help="override options_lab's 0.001 (a stated deviation)")
ap.add_argument("--out", default="")
args = ap.parse_args()
device = torch.device("cuda", 0)
flags, net, tr = build_trainer(args, 0, 1, device)
tr.initial_learning_rate *= args.lr_scale
if args.entropy_beta is not None:
    tr.entropy_beta = args.entropy_beta
if args.max_time_step:
    tr.max_global_time_step = int(args.max_time_step)
updates_total = tr.max_global_time_step / float(tr.Bg * flags.n_step_TD)
if updates_total < 5000:
    print("# WARNING: %.0f optimiser steps in the whole learning-rate schedule (max_time_step %d / (%d actors per update x "
          "%d steps)); the reference makes ~%d.  Use --groups to raise the update density." % (
              updates_total, tr.max_global_time_step, tr.Bg, flags.n_step_TD, tr.max_global_time_step // flags.n_step_TD),
          file=sys.stderr)
t_fill = time.time()
while not tr._full:
    tr.process(None, 0)
torch.cuda.synchronize()
t_fill = time.time() - t_fill
out = open(args.out, "w") if args.out else None
head = {"actors": args.actors, "groups": args.groups, "entropy_beta": tr.entropy_beta, "history": args.history, "lr0": tr.initial_learning_rate,
        "max_time_step": tr.max_global_time_step, "replay_fill_s": round(t_fill, 1)}
print(json.dumps(head), flush=True)
if out:
    out.write(json.dumps(head) + "\n")
global_t, t0, k = 0, time.time(), 0
while global_t < args.steps:
    tr.process(None, global_t + k % args.log_every * args.actors * flags.n_step_TD, sync_stats=False)
    k += 1
    if k % args.log_every == 0:
        steps, episodes, score_sum = tr.read_stats()
        global_t += steps
        l = tr._publish_losses()
        rw = tr.rewards[:tr.Bg * flags.n_step_TD]          # the last group's rollout: what the policy currently does
        live = tr.active_log[:tr.Bg * flags.n_step_TD].float()
        n_live = float(live.sum().clamp(min=1))
        bump = float(((rw < 0).float() * live).sum()) / n_live
        goal = float(((rw > 0).float() * live).sum()) / n_live
        line = json.dumps({"global_t": global_t, "episodes": episodes, "bump_rate": round(bump, 5), "goal_rate": round(goal, 6),
                           "mean_return": (score_sum / episodes) if episodes else None,
                           "total_loss": round(l["total_loss"], 4), "entropy": round(l["entropy"], 4),
                           "grad_norm": round(l["grad_norm"], 3), "steps_per_s": round(global_t / (time.time() - t0))})
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()
