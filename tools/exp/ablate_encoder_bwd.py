#!/usr/bin/env python3
"""Ablation timing of encoder_bwd's three phases (builds an experimental .so with -DUNREAL_ABLATE; GPU box only)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "gpurun_out", "libexp_enc.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUNREAL_ABLATE",
                       os.path.join(ROOT, "unreal_amd/csrc/encoder.hip"), "-o", so])
lib = ctypes.CDLL(so)
N = 81920
dev = "cuda:0"
pool = torch.randint(0, 256, (N * 21168,), dtype=torch.uint8, device=dev)
idx = torch.randperm(N, device=dev).to(torch.int32)
W2 = torch.randn(8192, device=dev) * .06
c1 = torch.relu(torch.randn(N * 6400, device=dev))
d2 = torch.randn(N * 2592, device=dev) * (torch.rand(N * 2592, device=dev) > 0.5)
g = [torch.zeros(n, device=dev) for n in (3072, 16, 8192, 32)]
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
for ph in (7, 0, 1, 2, 4):
    f = lambda: lib.exp_encoder_bwd_phases(ph, N, P(pool.data_ptr()), P(idx.data_ptr()), ctypes.c_float(1.0), P(W2.data_ptr()),
                                           P(c1.data_ptr()), P(d2.data_ptr()), P(g[0].data_ptr()), P(g[1].data_ptr()),
                                           P(g[2].data_ptr()), P(g[3].data_ptr()), P(st))
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print("phases=%d  %.3f ms" % (ph, e0.elapsed_time(e1) / 5))
