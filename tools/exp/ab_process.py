#!/usr/bin/env python3
"""Same-process A/B of Trainer.process() schedules on the bench workload (GPU box): toggles class attributes of UnrealModel
between rounds of K calls and reports the median ms per call of each setting.
  python tools/exp/ab_process.py fuse_bptt hoist_lstm_x            (each named attribute: default vs flipped)"""
import os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import build_trainer
from unreal_amd.model.model import UnrealModel
from unreal_amd.train.trainer import Trainer

names = [a for a in sys.argv[1:] if not a.startswith("-")]
args = argparse.Namespace(actors=4096, history=int(os.environ.get("AB_HISTORY", 300)), groups=1)
flags, net, tr = build_trainer(args, 0, 1, torch.device("cuda", 0))
while not tr._full:
    tr.process(None, 0)
torch.cuda.synchronize()


def timed(k=6):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(k):
        tr.process(None, 0)
    torch.cuda.synchronize()
    return (time.time() - t0) / k * 1e3


timed(3)
for name in names:
    UnrealModel_ = UnrealModel
    if name.startswith("Trainer."):
        UnrealModel, name = Trainer, name[8:]
    base = getattr(UnrealModel, name)
    res = {base: [], (not base): []}
    for rnd in range(5):
        for v in (base, not base):
            setattr(UnrealModel, name, v)
            res[v].append(timed())
    setattr(UnrealModel, name, base)
    med = {v: sorted(r)[len(r) // 2] for v, r in res.items()}
    print("%s: default (%s) %.2f ms   flipped (%s) %.2f ms   [rounds %s | %s]" % (
        name, base, med[base], not base, med[not base], " ".join("%.2f" % x for x in res[base]),
        " ".join("%.2f" % x for x in res[not base])), flush=True)
    UnrealModel = UnrealModel_
