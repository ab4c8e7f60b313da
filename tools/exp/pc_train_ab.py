#!/usr/bin/env python3
"""unreal_pc_deconv_train A/B (GPU box): the product library against whole-library variants built with extra -D flags
(`python tools/exp/pc_train_ab.py --build name=-DFLAG=V ...` in the container; they travel with the snapshot), timed in
interleaved rounds in ONE process at the trainer's 81,920 frames; outputs compared with the product's."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "exp", "build")
VARIANTS = {"base": []}
for a in sys.argv[1:]:
    if "=" in a and not a.startswith("--"):
        k, v = a.split("=", 1)
        VARIANTS[k] = v.split(",")


def so_of(v):
    return os.path.join(OUT, "libunreal_pc_%s.so" % v)


if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(ROOT, "unreal_amd", "csrc", "*.hip")))
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + fl +
                              srcs + ["-o", so_of(v)]) for v, fl in VARIANTS.items()]
    sys.exit(max(p.wait() for p in procs))

import torch  # noqa: E402
from unreal_amd import _lib, ops  # noqa: E402

libs = {}
for v in sorted(os.path.basename(f)[len("libunreal_pc_"):-3] for f in glob.glob(so_of("*"))):
    _lib.LIB_PATH = so_of(v)
    libs[v] = _lib._Lib()


def use(v):
    _lib._LIB = libs[v]


DEV = "cuda:0"
torch.manual_seed(0)
N, A = 81920, 4
rnd = lambda n: torch.randn(n, device=DEV)
hp = torch.relu(rnd(N * 2592))
Wv, bv, Wa, ba = rnd(512) * .04, rnd(1), rnd(2048) * .04, rnd(4)
act = torch.randint(0, 4, (N,), dtype=torch.int32, device=DEV)
tgt = rnd(N * 400)
mask = torch.ones(N, dtype=torch.int32, device=DEV)
s_hp = torch.zeros(1, device=DEV)
use("base")
ops.absmax(N, 2592, hp, 2592, s_hp)
outs, res = {}, {v: [] for v in libs}
for v in libs:
    use(v)
    d_hp = torch.zeros(N * 2592, device=DEV)
    g = [torch.zeros(n, device=DEV) for n in (512, 1, 2048, 4)]
    loss = torch.zeros(1, device=DEV)
    ops.pc_deconv_train(N, A, hp, Wv, bv, Wa, ba, act, tgt, mask, 0.05, 1.0, loss, d_hp, *g, hp_max=s_hp)
    torch.cuda.synchronize()
    outs[v] = (d_hp, g, loss)
for v in libs:
    if v != "base":
        print("%s vs base: d_hp identical %s, max |d dWa| %.3e (of %.3e), loss %.9g vs %.9g" % (
            v, bool(torch.equal(outs[v][0], outs["base"][0])), float((outs[v][1][2] - outs["base"][1][2]).abs().max()),
            float(outs["base"][1][2].abs().max()), float(outs[v][2]), float(outs["base"][2])))
for rnd_ in range(7):
    for v in libs:
        use(v)
        d_hp, g, loss = outs[v]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.pc_deconv_train(N, A, hp, Wv, bv, Wa, ba, act, tgt, mask, 0.05, 1.0, loss, d_hp, *g, hp_max=s_hp)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 5 * 1e3)
for v in libs:
    r = sorted(res[v])
    print("pc_deconv_train N=%d  %-12s median %8.1f us  min %8.1f us" % (N, v, r[len(r) // 2], r[0]))
