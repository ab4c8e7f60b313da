#!/usr/bin/env python3
"""Host-side cost of Trainer.process() in the launch-bound settings (GPU box): cProfile of a few calls at --groups G."""
import cProfile, pstats, os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import build_trainer
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
args = argparse.Namespace(actors=4096, history=200, groups=G)
flags, net, tr = build_trainer(args, 0, 1, torch.device("cuda", 0))
while not tr._full:
    tr.process(None, 0)
for _ in range(2):
    tr.process(None, 0)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    tr.process(None, 0)
torch.cuda.synchronize()
print("G=%d: %.1f ms per process()" % (G, (time.time() - t0) / 3 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.process(None, 0)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
