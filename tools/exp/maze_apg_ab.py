#!/usr/bin/env python3
"""maze_step_kernel: actors per workgroup at large batches (GPU box).  Whole-library variants built with -DMAZE_APG_BIG=n
(`python tools/exp/maze_apg_ab.py --build` in the container), timed interleaved in one process at 4096 actors through
ops.maze_step and the fused policy + environment step."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "exp", "build")
VARIANTS = {"apg8": ["-DMAZE_APG_BIG=8"], "apg4": ["-DMAZE_APG_BIG=4"], "apg2": ["-DMAZE_APG_BIG=2"], "apg16": ["-DMAZE_APG_BIG=16"]}
if "--tiny" in sys.argv:        # <= 64 actors: one actor per workgroup (product) against two
    VARIANTS = {"tiny1": ["-DMAZE_APG_TINY=1"], "tiny2": ["-DMAZE_APG_TINY=2"]}


def so_of(v):
    return os.path.join(OUT, "libunreal_maze_%s.so" % v)


if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(ROOT, "unreal_amd", "csrc", "*.hip")))
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + fl +
                              srcs + ["-o", so_of(v)]) for v, fl in VARIANTS.items()]
    sys.exit(max(p.wait() for p in procs))

import torch  # noqa: E402
from unreal_amd import _lib, ops  # noqa: E402

libs = {}
for v in VARIANTS:
    if os.path.exists(so_of(v)):
        _lib.LIB_PATH = so_of(v)
        libs[v] = _lib._Lib()


def use(v):
    _lib._LIB = libs[v]


DEV = "cuda:0"
for B in ((1, 8, 16, 64) if "--tiny" in sys.argv else (2048, 4096, 8192)):
    use(next(iter(libs)))
    ring = ops.Ring(B, 8, DEV)
    ops.maze_reset(ring)
    acts = torch.randint(0, 4, (B,), dtype=torch.int32, device=DEV)
    res = {v: [] for v in libs}
    for v in libs:
        use(v)
        ops.maze_step(ring, acts)
    torch.cuda.synchronize()
    for rnd in range(7):
        for v in libs:
            use(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.maze_step(ring, acts)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 20 * 1e3)
    print("maze_step B=%5d  " % B + "   ".join("%s %6.1f us" % (v, sorted(r)[3]) for v, r in res.items()))
    del ring
    torch.cuda.empty_cache()
