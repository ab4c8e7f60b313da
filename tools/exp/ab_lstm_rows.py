#!/usr/bin/env python3
"""Whole-kernel LSTM step at 4096 / 8192 / 16384 rows: 128x128 tiles with 2 wave groups (1 workgroup per CU) vs plain
128x128 workgroups (2 per CU).  `--build` in the container, run on the GPU box."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "build")
if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    ps = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLSTM_BIG_KW=%d" % v,
                            os.path.join(ROOT, "unreal_amd/csrc/gemm_split.hip"), "-o", os.path.join(OUT, "libsplit_bigkw%d.so" % v)]) for v in (1, 2)]
    sys.exit(max(p.wait() for p in ps))
import torch
sys.path.insert(0, ROOT)
from unreal_amd import ops
libs = {v: ctypes.CDLL(os.path.join(OUT, "libsplit_bigkw%d.so" % v)) for v in (1, 2)}
dev = "cuda:0"
K_x, xld = 261, 264
Wk = torch.randn((K_x + 256) * 1024, device=dev) * .05
sh = ops.LstmKernelShadow(Wk, K_x)
P = ctypes.c_void_p
st = torch.cuda.current_stream().cuda_stream
for B in (4096, 8192, 16384):
    x, h, c, b = (torch.randn(n, device=dev) for n in (B * xld, B * 256, B * 256, 1024))
    g, c2, h2 = torch.zeros(B * 1024, device=dev), torch.zeros(B * 256, device=dev), torch.zeros(B * 256, device=dev)
    res = {1: [], 2: []}
    for rnd in range(5):
        for v, lib in libs.items():
            run = lambda: lib.unreal_lstm_step_fwd(B, P(x.data_ptr()), xld, K_x, P(h.data_ptr()), 256, P(sh.planes.data_ptr()), sh.ldw,
                                                   ctypes.c_long(sh.plane), P(g.data_ptr()), P(b.data_ptr()), P(c.data_ptr()),
                                                   P(c2.data_ptr()), P(h2.data_ptr()), 256, P(st))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 100)
    m = {v: sorted(r)[2] for v, r in res.items()}
    print("rows %5d: plain 128x128 (when > 256 tiles) %.1f us   2 wave groups %.1f us" % (B, m[1], m[2]), flush=True)
