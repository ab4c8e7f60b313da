#!/usr/bin/env python3
"""Split-K sweep of the TN (wgrad) split GEMM at the trainer's shapes (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unreal_amd import ops
from unreal_amd.model.model import _splitk
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
R = 81920
for name, M, N, K in (("dW_fc1", 2592, 256, R), ("dW_lstm_x", 256, 1024, R), ("dW_pc_fc1", 256, 2592, R)):
    A = torch.randn(K * M, device="cuda"); B = torch.randn(K * N, device="cuda"); C = torch.zeros(M * N, device="cuda")
    line = "%-10s default sk=%d:" % (name, _splitk(M, N, K))
    for sk in (8, 12, 16, 24, 32, 48, 64, 96, 128):
        ms = timeit(lambda: ops.gemm_split_tn(M, N, K, A, M, B, N, C, N, splitk=sk))
        line += "  %d:%.3f" % (sk, ms)
    print(line, flush=True)
    del A, B, C
