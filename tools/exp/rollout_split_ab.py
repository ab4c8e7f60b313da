#!/usr/bin/env python3
"""the 20 lock-step rollout steps as two half-batches on two HIP streams (Trainer._rollout_steps_split)
against the single-stream loop, SAME trainer, same process, interleaved rounds.  GPU box.
usage: python tools/exp/rollout_split_ab.py [--actors 4096] [--history 2000] [--calls 20] [--rounds 3] [--parts 2]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import build_trainer
from unreal_amd.train.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--actors", type=int, default=4096)
ap.add_argument("--history", type=int, default=2000)
ap.add_argument("--groups", type=int, default=1)
ap.add_argument("--calls", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--parts", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda", 0)
Trainer.rollout_parts_default = args.parts
Trainer.ROLLOUT_SPLIT_MIN_ACTORS = 2
flags, net, tr = build_trainer(args, 0, 1, dev)
split = tr._split
assert split is not None
while not tr._full:
    tr.process(None, 0)
gt = 0
for _ in range(3):
    tr.process(None, gt, sync_stats=False); gt += args.actors * 20
res = {"lockstep": [], "split": []}
for r in range(args.rounds):
    for name, mode in (("lockstep", None), ("split", split)):
        tr._split = mode
        tr.process(None, gt, sync_stats=False); gt += args.actors * 20
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.calls):
            tr.process(None, gt, sync_stats=False); gt += args.actors * 20
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / args.calls * 1e3)
        print("round %d %-9s %.3f ms per process()" % (r, name, res[name][-1]), flush=True)
tr.read_stats()
l = tr._publish_losses()
med = lambda v: sorted(v)[len(v) // 2]
print(json.dumps({"actors": args.actors, "groups": args.groups, "parts": args.parts, "ms_lockstep": med(res["lockstep"]),
                  "ms_split": med(res["split"]), "delta_ms": med(res["split"]) - med(res["lockstep"]),
                  "total_loss": l["total_loss"]}))
