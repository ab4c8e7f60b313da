#!/usr/bin/env python3
"""Train the maze agent with the batched Trainer and log episode returns (GPU box).
usage: python tools/train_maze.py [--actors 4096] [--history 2000] [--steps 10000000] [--log-every 10] [--lr-scale 1.0]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--actors", type=int, default=4096)
ap.add_argument("--history", type=int, default=2000)
ap.add_argument("--steps", type=float, default=1e7)
ap.add_argument("--log-every", type=int, default=10)
ap.add_argument("--lr-scale", type=float, default=1.0)
ap.add_argument("--max-time-step", type=float, default=0)
args = ap.parse_args()
device = torch.device("cuda", 0)
flags, net, tr = build_trainer(args, 0, 1, device)
tr.initial_learning_rate *= args.lr_scale
if args.max_time_step:
    tr.max_global_time_step = int(args.max_time_step)
while not tr._full:
    tr.process(None, 0)
global_t, t0, k = 0, time.time(), 0
ep_n, ep_sum = 0, 0.0
while global_t < args.steps:
    tr.process(None, global_t, sync_stats=False)
    k += 1
    if k % args.log_every == 0:
        steps, episodes, score_sum = tr.read_stats()
        global_t += steps
        l = tr._publish_losses()
        print(json.dumps({"global_t": global_t, "episodes": episodes, "mean_return": (score_sum / episodes) if episodes else None,
                          "total_loss": round(l["total_loss"], 4), "entropy": round(l["entropy"], 4), "grad_norm": round(l["grad_norm"], 3),
                          "steps_per_s": round(global_t / (time.time() - t0))}), flush=True)
