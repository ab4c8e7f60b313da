#!/usr/bin/env python3
"""Return runs on the device (GPU box): the batched Trainer trained for --steps env steps from one seed, parameter
snapshots at --checkpoints (same file format as tools/return_oracle.py: ckpt-<t>.npz with the variables by name), one
log line every --log-every calls.  `--actors 8 --groups 8` is the reference algorithm actor after actor
(Trainer(groups = B): each actor's own clipped gradient applied in turn, = oracle process_async in call order);
`--actors 4096 --groups 1|8` the batched learner.  Resumable across gpurun calls (--resume: parameters + RMSProp slots +
global_t from the run directory; replay refilled like the reference does after a restore, main.py:382-427).

usage: python tools/return_device.py --seed S --actors 8 --groups 8 --steps 1e6 --out runs/return/dev_b8_sS"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--actors", type=int, default=8)
    ap.add_argument("--groups", type=int, default=8)
    ap.add_argument("--history", type=int, default=2000)
    ap.add_argument("--steps", type=float, default=1e6)
    ap.add_argument("--checkpoints", default="250000,500000,1000000")
    ap.add_argument("--max-time-step", type=float, default=13.2e6)
    ap.add_argument("--log-every", type=int, default=50)
    ap.add_argument("--budget-s", type=float, default=0, help="stop (resumably) after this many seconds")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    from unreal_amd.environment.environment import Environment
    from unreal_amd.model.model import UnrealModel
    from unreal_amd.options import get_options
    from unreal_amd.train.rmsprop_applier import RMSPropApplier
    from unreal_amd.train.trainer import Trainer, log_uniform
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    flags = get_options("training", preset="lab", argv=["--env_type", "maze", "--env_name", ""])
    Environment.action_size = -1
    A = Environment.get_action_size("maze", "")
    net = UnrealModel(A, 0, -1, True, True, True, True, flags.pixel_change_lambda, flags.entropy_beta, dev, seed=args.seed)
    lr0 = log_uniform(flags.initial_alpha_low, flags.initial_alpha_high, flags.initial_alpha_log_rate)
    applier = RMSPropApplier(None, decay=flags.rmsp_alpha, momentum=0.0, epsilon=flags.rmsp_epsilon,
                             clip_norm=flags.grad_norm_clip, device=dev)
    tr = Trainer(0, net, lr0, None, applier, "maze", "", True, True, True, True, flags.pixel_change_lambda,
                 flags.entropy_beta, flags.local_t_max, flags.n_step_TD, flags.gamma, flags.gamma_pc, args.history,
                 int(args.max_time_step), dev, batch_size=args.actors, seed=0xA3C + args.seed, groups=args.groups)
    tr.prepare()
    state_path = os.path.join(args.out, "state.pt")
    global_t, n_eps, calls = 0, 0, 0
    marks = sorted(int(float(x)) for x in args.checkpoints.split(",") if x)
    if args.resume and os.path.exists(state_path):
        st = torch.load(state_path, map_location="cpu", weights_only=True)
        net.params.flat.copy_(st["params"])
        net.mark_params_changed()
        applier._create_slots(net.params.flat)
        applier.ms.copy_(st["rms"])
        applier.mom.zero_()                            # momentum = 0.0 (main.py:240): the slot stays zero, not saved
        global_t, n_eps, calls = int(st["global_t"]), int(st["episodes"]), int(st["calls"])
        tr.draws.counter = int(st["draw_counter"])
        marks = [m for m in marks if m > global_t]
    t_fill = time.time()
    while not tr._full:
        tr.process(None, global_t)
    torch.cuda.synchronize()
    log = open(os.path.join(args.out, "log.jsonl"), "a")
    log.write(json.dumps({"seed": args.seed, "actors": args.actors, "groups": args.groups, "history": args.history,
                          "lr0": lr0, "max_time_step": tr.max_global_time_step, "resumed_at": global_t,
                          "fill_s": round(time.time() - t_fill, 1)}) + "\n")
    t0, g0 = time.time(), global_t
    ep_sum, ep_n, recent = 0.0, 0, []
    done = False
    while global_t < args.steps and not done:
        steps, score = tr.process(None, global_t)
        global_t += steps
        calls += 1
        # (score = mean return of the episodes that finished inside the call, None if none did)
        if score is not None:
            recent = (recent + [score])[-100:]
            n_eps += 1
        while marks and global_t >= marks[0]:
            m = marks.pop(0)
            np.savez(os.path.join(args.out, "ckpt-%d.npz" % m), **net.export_named())
        if calls % args.log_every == 0:
            l = tr.last_losses
            log.write(json.dumps({"global_t": global_t, "elapsed_s": round(time.time() - t0, 1),
                                  "steps_per_s": round((global_t - g0) / max(time.time() - t0, 1e-9), 1),
                                  "calls_with_finished_episodes": n_eps,
                                  "mean_return_last100_calls": (sum(recent) / len(recent)) if recent else None,
                                  "entropy": l.get("entropy"), "total_loss": l.get("total_loss"), "grad_norm": l.get("grad_norm")}) + "\n")
            log.flush()
        if args.budget_s and time.time() - t0 > args.budget_s:
            done = True
    torch.save({"params": net.params.flat.detach().cpu(), "rms": applier.ms.detach().cpu(),
                "global_t": global_t, "episodes": n_eps, "calls": calls, "draw_counter": tr.draws.counter}, state_path)
    if global_t >= args.steps:             # finished: nothing to resume (the state is 15 MB; gpurun_out travels back with <= 64 MiB)
        os.remove(state_path)
    log.write(json.dumps({"stopped_at": global_t, "finished": global_t >= args.steps, "elapsed_s": round(time.time() - t0, 1),
                          "steps_per_s": round((global_t - g0) / max(time.time() - t0, 1e-9), 1)}) + "\n")
    log.close()
    print(json.dumps({"out": args.out, "global_t": global_t, "steps_per_s": round((global_t - g0) / max(time.time() - t0, 1e-9), 1)}), flush=True)


if __name__ == "__main__":
    main()
