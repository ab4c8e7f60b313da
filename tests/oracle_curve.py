#!/usr/bin/env python3
"""Reference-algorithm learning curve on CPU: the oracle's threaded restatement of main.py:72-162 +
trainer.py:438-636 (hogwild RMSProp, per-thread replay), logged per finished episode.  Lives under tests/ because it drives the oracle
(test infrastructure); it is not a test and pytest does not collect it.
usage: python tests/oracle_curve.py --threads 8 --steps 1000000 --out tests/golden/oracle_curve_maze.json"""
import argparse, json, os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.trainer import OracleTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--steps", type=float, default=1e6)
ap.add_argument("--history", type=int, default=2000)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--out", default="gpurun_out/oracle_curve.json")
args = ap.parse_args()
torch.set_num_threads(1)
cfg = dict(action_size=4, use_lstm=True, use_pixel_change=True, use_value_replay=True, use_reward_prediction=True,
           pixel_change_lambda=0.05, entropy_beta=0.001, local_t_max=20, n_step_TD=20, gamma=0.99, gamma_pc=0.9,
           experience_history_size=args.history, max_time_step=int(13.2e6), rmsp_alpha=0.99, rmsp_epsilon=0.1,
           grad_norm_clip=40.0, initial_alpha_low=1e-4, initial_alpha_high=5e-3, initial_alpha_log_rate=0.5)
tr = OracleTrainer(cfg, n_actors=args.threads, seed=args.seed)
tr.fill()
state = {"t": 0}
episodes, stats = [], []
t0 = time.time()
lock = threading.Lock()

def work(i):
    while state["t"] < args.steps:
        d, score, losses = tr.process_async(i, state["t"])
        with lock:
            state["t"] += d
            if score is not None:
                episodes.append((state["t"], float(score)))
            if losses is not None and len(stats) < 200000:
                stats.append((state["t"], d, losses["total_loss"], float(np.sum(losses["entropy"]))))

ths = [threading.Thread(target=work, args=(i,)) for i in range(args.threads)]
for t in ths: t.start()
while any(t.is_alive() for t in ths):
    time.sleep(60)
    with lock:
        json.dump({"cfg": cfg, "threads": args.threads, "elapsed_s": time.time() - t0, "global_t": state["t"],
                   "episodes": episodes, "updates": stats[::50]}, open(args.out, "w"))
    print("t=%d  %.0f steps/s  episodes=%d" % (state["t"], state["t"] / (time.time() - t0), len(episodes)), flush=True)
for t in ths: t.join()
json.dump({"cfg": cfg, "threads": args.threads, "elapsed_s": time.time() - t0, "global_t": state["t"],
           "episodes": episodes, "updates": stats[::50]}, open(args.out, "w"))
