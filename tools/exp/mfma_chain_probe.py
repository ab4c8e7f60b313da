#!/usr/bin/env python3
"""Cycles per bf16 MFMA as a function of the number of independent accumulator chains (GPU box)."""
import ctypes, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "build", "libmfma_probe.so"))
out = torch.zeros(256 * 8, dtype=torch.int64, device="cuda:0")
sink = torch.zeros(256 * 512, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
iters = 200
for kind, per in ((16, 48), (32, 24)):
    for waves in (4, 8):
        for nch in ((1, 2, 3, 4, 8) if kind == 16 and waves == 4 else (1, 2, 4, 8) if kind == 16 else (1, 2, 4) if waves == 4 else (1, 2)):
            for rep in range(2):
                lib.probe(kind, nch, waves, P(out.data_ptr()), P(sink.data_ptr()), iters, P(st))
            torch.cuda.synchronize()
            c = out[:256 * waves].double().cpu()
            big = 20000
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.probe(kind, nch, waves, P(out.data_ptr()), P(sink.data_ptr()), big, P(st))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            c2 = out[:256 * waves].double().cpu()
            flops = 256.0 * waves * big * per * (kind * kind * (512 // kind) * 2)
            print("mfma %dx%d  %d wave(s)/SIMD  %d chain(s): %.1f ticks per MFMA per wave; %.0f ms wall -> %.0f TFLOP/s, tick rate %.2f GHz" % (
                kind, kind, waves // 4, nch, float(c.median()) / (iters * per), ms, flops / ms / 1e9, float(c2.median()) / ms / 1e6))
