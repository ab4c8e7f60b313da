#!/usr/bin/env python3
"""Where the time of one whole-kernel LSTM step (4096 rows) goes: in-kernel s_memtime stamps of one workgroup against the
launch time seen by HIP events.  Build the diagnostic libraries in the container first (`--build`), run on the GPU box."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "build")
VARIANTS = {"128x128_kw2": [], "64x64_kw1": ["-DLSTM_FORCE_64"]}
if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    ps = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSPLIT_STAMPS"] + f +
                           [os.path.join(ROOT, "unreal_amd/csrc/gemm_split.hip"), "-o", os.path.join(OUT, "libsplit_%s.so" % v)])
          for v, f in VARIANTS.items()]
    sys.exit(max(p.wait() for p in ps))
import torch
sys.path.insert(0, ROOT)
from unreal_amd import ops
dev = "cuda:0"
B, K_x, xld = 4096, 261, 264
Wk = torch.randn((K_x + 256) * 1024, device=dev) * .05
sh = ops.LstmKernelShadow(Wk, K_x)
x, h, c, b = (torch.randn(n, device=dev) for n in (B * xld, B * 256, B * 256, 1024))
g, c2, h2 = torch.zeros(B * 1024, device=dev), torch.zeros(B * 256, device=dev), torch.zeros(B * 256, device=dev)
P = ctypes.c_void_p
st = torch.cuda.current_stream().cuda_stream
for v in VARIANTS:
    lib = ctypes.CDLL(os.path.join(OUT, "libsplit_%s.so" % v))

    def run():
        rc = lib.unreal_lstm_step_fwd(B, P(x.data_ptr()), xld, K_x, P(h.data_ptr()), 256, P(sh.planes.data_ptr()), sh.ldw,
                                      ctypes.c_long(sh.plane), P(g.data_ptr()), P(b.data_ptr()), P(c.data_ptr()),
                                      P(c2.data_ptr()), P(h2.data_ptr()), 256, P(st))
        assert rc == 0, rc
    for _ in range(3):
        run()
    buf = (ctypes.c_ulonglong * 8)()
    lib.exp_split_read_stamps(buf, 1)
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    lib.exp_split_read_stamps(buf, 0)
    us = e0.elapsed_time(e1) * 1e3 / reps
    seg = [buf[k] / reps / 100.0 for k in range(4)]          # s_memtime ticks at 100 MHz
    print("%-12s launch %.1f us | workgroup 9: prologue %.1f  K loop %.1f  park partials %.1f  gate epilogue %.1f  (sum %.1f us)"
          % (v, us, seg[0], seg[1], seg[2], seg[3], sum(seg)), flush=True)
