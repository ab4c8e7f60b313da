#!/usr/bin/env python3
"""unreal_encoder_fwd with and without the prepared weight block (unreal_encoder_prepare), interleaved in one process (GPU box):
launch time at the row counts of a rollout step (8 ... 4096 frames) and of a replay pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from unreal_amd import ops  # noqa: E402

DEV = "cuda:0"
torch.manual_seed(0)
rnd = lambda n: torch.randn(n, device=DEV)
W1, b1, W2, b2 = rnd(3072) * .07, rnd(16) * .07, rnd(8192) * .06, rnd(32) * .06
prep = ops.encoder_prepare(W1, b1, W2, 1.0 / 255.0)
for N, save in ((8, False), (64, False), (512, False), (4096, False), (160, True), (10240, True), (81920, True)):
    pool = torch.randint(0, 256, (N * ops.FRAME_BYTES,), dtype=torch.uint8, device=DEV)
    idx = torch.randperm(N, device=DEV).to(torch.int32)
    f2 = torch.zeros(N * 2592, device=DEV)
    c1 = torch.zeros(N * 6400, device=DEV) if save else None
    s2 = torch.zeros(1, device=DEV)
    res = {"per workgroup": [], "prepared": []}

    def run(v):
        ops.encoder_fwd(pool, idx, 1.0 / 255.0, W1, b1, W2, b2, f2, c1, f2_max=s2, prepared=prep if v == "prepared" else None)

    for v in res:
        run(v)
    torch.cuda.synchronize()
    reps = 20 if N <= 4096 else 5
    for rnd_ in range(7):
        for v in res:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / reps * 1e3)
    print("N=%6d save_c1=%-5s  " % (N, save) + "   ".join("%s %8.1f us" % (v, sorted(r)[3]) for v, r in res.items()))
    del pool, f2, c1
    torch.cuda.empty_cache()
