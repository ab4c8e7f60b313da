#!/usr/bin/env python3
"""Ablation timing of the split-operand GEMM (run on the GPU box): compiles csrc/gemm_split.hip with
-DSPLIT_ABLATE=<bits> (1: no global loads in the K loop, 2: no split/LDS store after the first tile, 4: no LDS read/MFMA)
and times each variant at the trainer's shapes.  Results are wrong by construction; only the times matter."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "ablate")
os.makedirs(OUT, exist_ok=True)
SHAPES = (("fc_fwd", 81920, 256, 2592), ("pc_fc1", 81920, 2592, 256), ("d_fc", 81920, 256, 1024), ("fc_roll", 4096, 256, 2592), ("dh_rec", 4096, 256, 1024))


def build(bits):
    so = os.path.join(OUT, "split_%d.so" % bits)
    extra = []
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DSPLIT_ABLATE=%d" % (bits % 100),
                           os.path.join(ROOT, "unreal_amd/csrc/gemm_split.hip"), "-o", so] + extra + sys.argv[1:])
    return ctypes.CDLL(so)


def main():
    for bits in (0, 8, 1, 3, 4):
        lib = build(bits)
        f = lib.unreal_gemm_f32_split_nt
        f.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_long,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        line = "ablate=%d" % bits
        for name, M, N, K in SHAPES:
            A = torch.randn(M * K, device="cuda"); C = torch.zeros(M * N, device="cuda")
            B = torch.randn(3 * N * K, device="cuda").to(torch.bfloat16)
            st = torch.cuda.current_stream().cuda_stream
            run = lambda: f(M, N, K, A.data_ptr(), K, B.data_ptr(), K, N * K, C.data_ptr(), N, None, None, 0, 0, 1, st)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            line += "  %s %.3f ms" % (name, e0.elapsed_time(e1) / 5)
            del A, B, C
        g = lib.unreal_gemm_f32_split_tn
        g.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        for name, M, N, K, sk in (("tn_fc1", 2592, 256, 81920, 25), ("tn_lstm", 256, 1024, 81920, 64)):
            A = torch.randn(K * M, device="cuda"); B = torch.randn(K * N, device="cuda"); C = torch.zeros(M * N, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            run = lambda: g(M, N, K, A.data_ptr(), M, B.data_ptr(), N, C.data_ptr(), N, None, sk, st)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            line += "  %s %.3f ms" % (name, e0.elapsed_time(e1) / 5)
            del A, B, C
        print(line, flush=True)


if __name__ == "__main__":
    main()
