#!/usr/bin/env python3
"""Return baseline on the CPU: the oracle's restatement of the REFERENCE algorithm
(oracle/trainer.py `process_async`: parallel_size = 8 actors, each computes its own gradient from the shared
parameters, clips it at 40 and steps the shared RMSProp; /root/reference/train/trainer.py:438-636, main.py:72-162)
run for --steps env steps from one seed, with parameter snapshots at --checkpoints.

The reference's threads race (hogwild order = whatever the scheduler does, main.py:455); here the 8 actors take turns
round-robin in ONE thread, which is one legal interleaving and makes a seed reproducible.  Draws come from one shared
RandomState like main.py:213.  Five seeds are run as five processes (tools/return_runs.sh); the capped-return /
success-rate numbers are then measured by the evaluator on the snapshots (tools/return_eval.py), the same instrument for
both implementations.

TEST / MEASUREMENT INFRASTRUCTURE: imports oracle/, never part of the product path."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.trainer import OracleTrainer, RefDraws  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--steps", type=float, default=1e6)
    ap.add_argument("--checkpoints", default="250000,500000,1000000")
    ap.add_argument("--actors", type=int, default=8, help="parallel_size (options.py:37)")
    ap.add_argument("--history", type=int, default=2000)
    ap.add_argument("--max-time-step", type=float, default=13.2e6)
    ap.add_argument("--out", required=True, help="directory: log.jsonl, episodes.jsonl, ckpt-<t>.npz")
    args = ap.parse_args()
    torch.set_num_threads(1)
    os.makedirs(args.out, exist_ok=True)
    cfg = dict(action_size=4, use_lstm=True, use_pixel_change=True, use_value_replay=True,
               use_reward_prediction=True, pixel_change_lambda=0.05, entropy_beta=0.001, local_t_max=20,
               n_step_TD=20, gamma=0.99, gamma_pc=0.9, experience_history_size=args.history,
               max_time_step=int(args.max_time_step), rmsp_alpha=0.99, rmsp_epsilon=0.1, grad_norm_clip=40.0,
               initial_alpha_low=1e-4, initial_alpha_high=5e-3, initial_alpha_log_rate=0.5)
    shared = RefDraws(np.random.RandomState(0xA3C + args.seed))
    tr = OracleTrainer(cfg, n_actors=args.actors, draws=[shared] * args.actors, seed=args.seed)
    marks = sorted(int(float(x)) for x in args.checkpoints.split(",") if x)
    t0 = time.time()
    tr.fill()
    log = open(os.path.join(args.out, "log.jsonl"), "a")
    eps = open(os.path.join(args.out, "episodes.jsonl"), "a")
    log.write(json.dumps({"seed": args.seed, "actors": args.actors, "history": args.history, "fill_s": time.time() - t0,
                          "lr0": tr.initial_lr, "max_time_step": cfg["max_time_step"]}) + "\n")
    log.flush()
    global_t, k, n_eps, last = 0, 0, 0, {}
    recent = []
    t0 = time.time()
    next_log = 0
    while global_t < args.steps:
        i = k % args.actors
        k += 1
        d, score, losses = tr.process_async(i, global_t)
        global_t += d
        if losses is not None:
            last = losses
        if score is not None:
            n_eps += 1
            recent = (recent + [score])[-100:]
            eps.write(json.dumps({"t": global_t, "return": score}) + "\n")
            eps.flush()
        while marks and global_t >= marks[0]:
            m = marks.pop(0)
            np.savez(os.path.join(args.out, "ckpt-%d.npz" % m), **{n: v.numpy() for n, v in tr.params.items()})
        if global_t >= next_log:
            next_log += 20000
            log.write(json.dumps({"global_t": global_t, "elapsed_s": round(time.time() - t0, 1),
                                  "steps_per_s": round(global_t / max(time.time() - t0, 1e-9), 1), "episodes": n_eps,
                                  "mean_return_last100": (sum(recent) / len(recent)) if recent else None,
                                  "entropy": float(np.sum(last.get("entropy", 0.0))) if last else None,
                                  "total_loss": last.get("total_loss"), "grad_norm": last.get("grad_norm")}) + "\n")
            log.flush()
    log.write(json.dumps({"done": True, "global_t": global_t, "elapsed_s": round(time.time() - t0, 1), "episodes": n_eps}) + "\n")
    log.close()
    eps.close()


if __name__ == "__main__":
    main()
