#!/usr/bin/env python3
"""Fused BPTT step at 4096 / 8192 rows with 1 / 2 / 4 wave groups per 64x64 tile.  `--build` in the container."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "build")
if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    ps = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DBPTT_FORCE_KW=%d" % v,
                            os.path.join(ROOT, "unreal_amd/csrc/gemm_split.hip"), "-o", os.path.join(OUT, "libsplit_bptt%d.so" % v)]) for v in (1, 2, 4)]
    sys.exit(max(p.wait() for p in ps))
import torch
sys.path.insert(0, ROOT)
from unreal_amd import ops
libs = {v: ctypes.CDLL(os.path.join(OUT, "libsplit_bptt%d.so" % v)) for v in (1, 2, 4)}
dev = "cuda:0"
Wh = torch.randn(256 * 1024, device=dev) * .05
sh = ops.SplitWeights(Wh, 256, 1024, 1024, False)
P = ctypes.c_void_p
st = torch.cuda.current_stream().cuda_stream
for B in (4096, 8192):
    dg, ga, dpre = (torch.randn(B * 1024, device=dev) for _ in range(3))
    dh, dc, cp, cn = (torch.randn(B * 256, device=dev) for _ in range(4))
    res = {v: [] for v in libs}
    for rnd in range(5):
        for v, lib in libs.items():
            run = lambda: lib.unreal_lstm_bptt_step(B, P(dg.data_ptr()), P(sh.planes.data_ptr()), sh.ldw, ctypes.c_long(sh.plane), P(dh.data_ptr()),
                                                    P(dc.data_ptr()), P(ga.data_ptr()), P(cp.data_ptr()), P(cn.data_ptr()), P(dpre.data_ptr()), P(st))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 100)
    print("rows %5d: " % B + "   ".join("%d wave group(s) %.1f us" % (v, sorted(r)[2]) for v, r in res.items()), flush=True)
