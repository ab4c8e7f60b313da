#!/usr/bin/env python3
"""Capped-return evaluation of parameter snapshots (GPU box): every ckpt-<t>.npz under the given run directories is
loaded into an UnrealModel and rolled by unreal_amd.evaluate.Evaluate -- sampled policy, the first episode of each of `--batch` lock-step actors,
`--cap` steps per episode (this fork's maze has no step limit, maze_environment.py:98-128; an episode that does not reach
the goal within the cap counts as a failure and contributes the return it has collected).  The same instrument for the
oracle's (CPU, reference algorithm) and the device's snapshots.  One JSON line per snapshot.

usage: python tools/return_eval.py [--batch 256] [--cap 2000] runs/return/oracle_s0 runs/return/dev_b8_s0 ... > out.jsonl"""
import argparse
import glob
import json
import os
import re
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--cap", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    from unreal_amd.evaluate import Evaluate
    from unreal_amd.model.model import UnrealModel
    dev = torch.device("cuda", 0)
    net = UnrealModel(4, 0, -1, True, True, True, True, 0.05, 0.001, dev, seed=0)
    ev = Evaluate(net, batch_size=args.batch, device=dev, seed=0xE7A1)
    for d in args.dirs:
        for p in sorted(glob.glob(os.path.join(d, "ckpt-*.npz")), key=lambda x: int(re.findall(r"ckpt-(\d+)", x)[-1])):
            z = np.load(p, allow_pickle=False)
            net.load_named({k: z[k] for k in z.files})
            ev.draws.counter = 1                       # every snapshot sees the same evaluation draws
            r = ev.process(args.batch, max_episode_steps=args.cap, one_episode_per_actor=True)
            r.update(run=os.path.basename(os.path.normpath(d)), t=int(re.findall(r"ckpt-(\d+)", p)[-1]), cap=args.cap)
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
