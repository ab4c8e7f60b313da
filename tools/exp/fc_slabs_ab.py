#!/usr/bin/env python3
"""fc 2592 -> 256 at few rows (GPU box): the one-launch split GEMM against unreal_gemm_f32_split_nt_slabs (K slabs in separate
workgroups + ordered sum), interleaved in one process, bias + ReLU + max |C| in both."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from unreal_amd import ops  # noqa: E402

DEV = "cuda:0"
torch.manual_seed(0)
N, K = 256, 2592
Wsrc = torch.randn(N * K, device=DEV) * 0.02
W = ops.SplitWeights(Wsrc, N, K, K, False)
bias = torch.randn(N, device=DEV) * 0.1
for M in (8, 512, 1024, 2048, 4096, 8192):
    A = torch.relu(torch.randn(M * K, device=DEV))
    amax, cmax = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    ops.absmax(M, K, A, K, amax)
    C = torch.zeros(M * N, device=DEV)
    cands = {"one launch": None}
    for S in sorted({ops.slab_count(M, N, K) or 4, 2, 3, 4, 8}):
        cands["slabs S=%d" % S] = (S, torch.empty(S * M * N, device=DEV))

    def run(v):
        if cands[v] is None:
            ops.gemm_split_nt(M, N, K, A, K, W, C, N, bias=bias, flags=ops.GEMM_RELU, a_max=amax, c_max=cmax)
        else:
            ops.gemm_split_nt_slabs(M, N, K, A, K, W, C, N, cands[v][1], cands[v][0], bias=bias, flags=ops.GEMM_RELU, a_max=amax, c_max=cmax)

    res = {v: [] for v in cands}
    for v in cands:
        run(v)
    torch.cuda.synchronize()
    for rnd in range(7):
        for v in cands:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 20 * 1e3)
    pick = ops.slab_count(M, N, K)
    print("M=%5d  " % M + "   ".join("%s %6.1f us" % (v, sorted(r)[3]) for v, r in res.items()) + "   (slab_count: %d)" % pick)
