#!/usr/bin/env python3
"""encoder_fwd: A/B timing of build variants + per-phase stamps (GPU box).  The variant libraries are built in the
container first (`python tools/exp/fwd_ab.py --build`; they travel with the snapshot) from csrc/encoder.hip alone with
one knob flipped each; every variant is timed in interleaved rounds in ONE process on the same operands (random uint8
frames), at the two launch sizes of the trainer (163,840-row replay pass / 4096-row rollout step), and its outputs are
compared with variant `base` (bit-for-bit where the arithmetic is the same)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "build")
VARIANTS = {                       # name -> -D flags
    "base": ["-DENC_FWD_MAGIC=0", "-DENC_FWD_FR2=0"],          # table-driven conv2 + symmetric epilogue + prefetch (commit 1)
    "fr2": ["-DENC_FWD_MAGIC=0", "-DENC_FWD_FR2=1"],
    "magic": ["-DENC_FWD_MAGIC=1", "-DENC_FWD_FR2=0"],
    "fr2+magic": ["-DENC_FWD_MAGIC=1", "-DENC_FWD_FR2=1"],
}
for a in sys.argv[1:]:             # extra variants: name=-DX=1,-DY=2
    if "=" in a and not a.startswith("--"):
        k, v = a.split("=", 1)
        VARIANTS[k] = v.split(",")


def so_of(v):
    return os.path.join(OUT, "libfwd_%s.so" % v.replace("+", "_"))


if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-DENC_FWD_STAMPS"] + fl + [os.path.join(ROOT, "unreal_amd/csrc/encoder.hip"), "-o", so_of(v)])
             for v, fl in VARIANTS.items()]
    sys.exit(max(p.wait() for p in procs))

import torch  # noqa: E402

N = int(os.environ.get("FWD_N", 81920))
dev = "cuda:0"
torch.manual_seed(0)
pool = torch.randint(0, 256, (N * 21168,), dtype=torch.uint8, device=dev)
idx = torch.randperm(N, device=dev).to(torch.int32)
W1 = torch.randn(192 * 16, device=dev) * 0.07
b1 = torch.randn(16, device=dev) * 0.1
W2 = torch.randn(256 * 32, device=dev) * 0.06
b2 = torch.randn(32, device=dev) * 0.1
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
libs = {v: ctypes.CDLL(so_of(v)) for v in VARIANTS if os.path.exists(so_of(v))}
outs = {v: dict(c1=torch.zeros(N * 6400, device=dev), f2=torch.zeros(N * 2592, device=dev),
                bits=torch.zeros(N * 162, dtype=torch.int16, device=dev), s=torch.zeros(2, device=dev))
        for v in (("base", "x") if "base" in libs else ("x",))}


def run(v, n, save=True, o=None):
    o = o or outs["x"]
    rc = libs[v].unreal_encoder_fwd(n, P(pool.data_ptr()), P(idx.data_ptr()), ctypes.c_float(1.0 / 255), P(W1.data_ptr()),
                                    P(b1.data_ptr()), P(W2.data_ptr()), P(b2.data_ptr()),
                                    P(o["c1"].data_ptr()) if save else None, P(o["f2"].data_ptr()),
                                    P(o["bits"].data_ptr()) if save else None, P(o["s"].data_ptr()),
                                    P(o["s"].data_ptr() + 4), P(st))
    assert rc == 0, rc


def timed(v, n, save, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(v, n, save)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# correctness against `base`
if "base" in libs:
    run("base", N, True, outs["base"])
    for v in libs:
        if v == "base":
            continue
        for k in ("c1", "f2", "s"):
            outs["x"][k].zero_()
        outs["x"]["bits"].zero_()
        run(v, N, True, outs["x"])
        torch.cuda.synchronize()
        d = {k: float((outs["x"][k] - outs["base"][k]).abs().max()) for k in ("c1", "f2", "s")}
        same_bits = bool((outs["x"]["bits"] == outs["base"]["bits"]).all())
        print("variant %-10s vs base: max |d c1| %.3e  max |d f2| %.3e  max |d absmax| %.3e  relu bits equal %s   (max c1 %.3f f2 %.3f)" % (
            v, d["c1"], d["f2"], d["s"], same_bits, float(outs["base"]["c1"].max()), float(outs["base"]["f2"].max())))

for n, save, reps, what in ((N, True, 3, "%d frames, c1 + bits saved" % N), (4096, True, 20, "4096 frames, c1 + bits saved"),
                            (4096, False, 20, "4096 frames, inference")):
    for v in libs:
        run(v, n, save)
    torch.cuda.synchronize()
    res = {v: [] for v in libs}
    for rnd in range(5):
        for v in libs:
            res[v].append(timed(v, n, save, reps))
    for v in libs:
        r = sorted(res[v])
        print("%-34s %-10s median %8.1f us   min %8.1f us" % (what, v, r[len(r) // 2], r[0]))

names = ["loop top", "conv1", "wait F1", "dma issue + conv2", "wait own DMA", "wait F2", "epilogue"]
iters = (N + 511) // 512
for v in libs:
    buf = (ctypes.c_ulonglong * 32)()
    libs[v].exp_read_fstamps(buf, 1)
    run(v, N, True)
    libs[v].exp_read_fstamps(buf, 0)
    print("stamps of %s (ticks of wave lifetime per frame, workgroup 3):" % v)
    for w in range(4):
        print("  wave %d: " % w + " | ".join("%s %d" % (names[k], buf[w * 8 + k] // iters) for k in range(7))
              + " | total %d" % (sum(buf[w * 8 + k] for k in range(7)) // iters))
