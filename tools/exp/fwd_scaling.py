#!/usr/bin/env python3
"""encoder_fwd launch time against the number of frames (GPU box): with 512 workgroups a launch of 512 k frames is
prologue + k frames per workgroup -- the intercept is the per-launch overhead the 4096-frame rollout steps pay 21 times."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unreal_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
NMAX = 16384
pool = torch.randint(0, 256, (NMAX * 21168,), dtype=torch.uint8, device=dev)
idx = torch.randperm(NMAX, device=dev).to(torch.int32)
W1 = torch.randn(192 * 16, device=dev) * 0.07; b1 = torch.randn(16, device=dev) * 0.1
W2 = torch.randn(256 * 32, device=dev) * 0.06; b2 = torch.randn(32, device=dev) * 0.1
f2 = torch.zeros(NMAX * 2592, device=dev); c1 = torch.zeros(NMAX * 6400, device=dev)
bits = torch.zeros(NMAX * 162, dtype=torch.int16, device=dev)
s = torch.zeros(2, device=dev)
for save in (True, False):
    for N in (64, 512, 1024, 2048, 4096, 8192, 16384):
        run = lambda: ops.encoder_fwd(pool, idx[:N], 1.0 / 255, W1, b1, W2, b2, f2, c1 if save else None,
                                      relu_bits=bits if save else None, f2_max=s[0:1], c1_max=s[1:2])
        for _ in range(3):
            run()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        ts.sort()
        print("save_c1=%-5s N=%6d  %7.1f us  (%.1f frames per workgroup)" % (save, N, ts[3], N / min(N, 512)))
