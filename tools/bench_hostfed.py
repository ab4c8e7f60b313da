#!/usr/bin/env python3
"""Throughput of the HOST-FED path (BASELINE configs 4 / 5) on one GPU: simulators on the host -> pinned staging ->
PCIe -> HBM ring -> the same batched learner.  DeepMind Lab / MINOS are not in the image, so the simulator here is a
replay of pre-generated uint8 frames (a fixed pool cycled per step; rewards / terminals / objectives as in
unreal_amd.environment.synthetic_sim): the number is the rate of everything EXCEPT the simulator itself.

  python tools/bench_hostfed.py [--env lab|indoor] [--actors 1024] [--history 200] [--steps 5] [--objective 5]

Prints one JSON line: env-steps/s of (a) ingest only (stage + H2D copy + hostfed_step kernel, no learner) and
(b) Trainer.process() end to end (PCIe-inclusive), plus the bytes moved over PCIe per env step."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class PoolSimulator(object):
    """Batched host simulator that costs (almost) nothing: frames come from a pre-generated pool."""

    def __init__(self, batch, objective_size=0, episode_len=900, pool=4, seed=4, ms_per_step=0.0):
        rs = np.random.RandomState(seed)
        self.B, self.objective_size, self.episode_len = batch, objective_size, episode_len
        self.pool = [rs.randint(0, 256, size=(batch, 84, 84, 3)).astype(np.uint8) for _ in range(pool)]
        self.obj = [rs.uniform(-1, 1, size=(batch, max(objective_size, 1))).astype(np.float32) for _ in range(pool)]
        self.rew = [(rs.random_sample(batch) < 0.01).astype(np.float32) for _ in range(pool)]
        self.t, self.k = np.zeros(batch, np.int64), 0
        self.ms_per_step = ms_per_step          # modelled cost of stepping ALL actors once (worker processes: the host
                                                # thread just waits), scaled by the share of actors a call steps

    def _out(self, frames, *rest):
        return (frames,) + rest + ((self.obj[self.k % len(self.obj)][:, :self.objective_size],) if self.objective_size else ())

    def reset(self, mask=None):
        self.k += 1
        out = self._out(self.pool[self.k % len(self.pool)])
        return out if self.objective_size else out[0]

    def step(self, actions, active=None):
        self.k += 1
        live = np.ones(self.B, bool) if active is None else (np.asarray(active) != 0)
        if self.ms_per_step > 0:
            time.sleep(1e-3 * self.ms_per_step * float(live.sum()) / self.B)
        self.t[live] += 1
        term = ((self.t >= self.episode_len) & live).astype(np.int32)
        self.t[term != 0] = 0
        return self._out(self.pool[self.k % len(self.pool)], self.rew[self.k % len(self.rew)], term)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="lab", choices=["lab", "indoor"])
    ap.add_argument("--actors", type=int, default=1024)
    ap.add_argument("--history", type=int, default=200)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--objective", type=int, default=5)
    ap.add_argument("--no-overlap", action="store_true", help="lock-step host / device phases (round-1 behaviour)")
    ap.add_argument("--overlap", action="store_true", help="force the two-half-batch schedule (default: from 2048 actors)")
    ap.add_argument("--sim-ms", type=float, default=0.0, help="modelled simulator time per step of the whole batch (ms)")
    args = ap.parse_args()
    from unreal_amd.environment.environment import Environment
    from unreal_amd.model.model import UnrealModel
    from unreal_amd.options import get_options
    from unreal_amd.train.rmsprop_applier import RMSPropApplier
    from unreal_amd.train.trainer import Trainer, log_uniform
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    obj = args.objective if args.env == "indoor" else 0
    name = "pool_rooms" if args.env == "indoor" else "pool_lab"
    if obj:
        Environment.register_indoor_config(name, obj)
    flags = get_options("training", preset="lab", argv=["--env_type", args.env, "--env_name", name])
    Environment.action_size = -1
    A = Environment.get_action_size(args.env, name)
    net = UnrealModel(A, obj, -1, True, True, True, True, flags.pixel_change_lambda, flags.entropy_beta, dev, seed=1,
                      frame_scale=1.0 / 255.0)
    applier = RMSPropApplier(None, decay=flags.rmsp_alpha, momentum=0.0, epsilon=flags.rmsp_epsilon,
                             clip_norm=flags.grad_norm_clip, device=dev)
    sim = PoolSimulator(args.actors, objective_size=obj, ms_per_step=args.sim_ms)
    tr = Trainer(0, net, log_uniform(flags.initial_alpha_low, flags.initial_alpha_high, flags.initial_alpha_log_rate),
                 None, applier, args.env, name, True, True, True, True, flags.pixel_change_lambda, flags.entropy_beta,
                 flags.local_t_max, flags.n_step_TD, flags.gamma, flags.gamma_pc, args.history, flags.max_time_step, dev,
                 batch_size=args.actors, simulator=sim, overlap_host=False if args.no_overlap else (True if args.overlap else None))
    tr.prepare()
    t0 = time.time()
    while not tr._full:
        tr.process(None, 0)
    fill_s = time.time() - t0
    # (a) ingest only
    env, B = tr.environment, args.actors
    acts = torch.zeros(B, dtype=torch.int32, device=dev)
    for _ in range(3):
        env.process(acts, None, tr.rewards[:B], tr.terminals[:B])
    torch.cuda.synchronize()
    n_ing = 40
    t0 = time.time()
    for _ in range(n_ing):
        env.process(acts, None, tr.rewards[:B], tr.terminals[:B])
    torch.cuda.synchronize()
    ingest = n_ing * B / (time.time() - t0)
    # (b) whole path
    tr.process(None, 0)
    torch.cuda.synchronize()
    t0, total = time.time(), 0
    for _ in range(args.steps):
        steps, _ = tr.process(None, total)
        total += steps
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(json.dumps({"env": args.env, "actors": B, "overlap_host": bool(tr.overlap_host), "sim_ms_per_step": args.sim_ms, "action_size": A, "objective_size": obj, "history": args.history,
                      "replay_fill_s": round(fill_s, 2), "ingest_only_env_steps_per_s": round(ingest),
                      "process_env_steps_per_s": round(total / dt), "ms_per_process": round(1e3 * dt / args.steps, 2),
                      "pcie_bytes_per_env_step": 21168 + 8 + 4 * obj + 4,
                      "note": "simulator = replay of a pre-generated frame pool (cost of the real simulator excluded)"}))


if __name__ == "__main__":
    main()
