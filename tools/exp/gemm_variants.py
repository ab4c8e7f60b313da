#!/usr/bin/env python3
"""Timing-only ablation of the k-major (KM) operand path of the GEMM (results are garbage for EXP != 0)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = 4096
dev = "cuda:0"
A = torch.randn(n * n, device=dev); B = torch.randn(n * n, device=dev); C = torch.zeros(n * n, device=dev)
P = ctypes.c_void_p
st = torch.cuda.current_stream().cuda_stream
for exp in (0, 2, 3, 4):
    so = "/tmp/libexp_gemm%d.so" % exp
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DEXP=%d" % exp,
                           os.path.join(ROOT, "tools/exp/gemm_exp.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    for ta, tb, name in ((0, 1, "NT rk/rk"), (0, 0, "NN rk/km"), (1, 0, "TN km/km")):
        f = lambda: lib.exp_gemm_f32(ta, tb, n, n, n, P(A.data_ptr()), n, P(B.data_ptr()), n, P(C.data_ptr()), n, None, None, 0, 0, 1, P(st))
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("EXP=%d %-10s %.3f ms  %.1f TF" % (exp, name, ms, 2.0 * n ** 3 / ms / 1e9), flush=True)
