#!/usr/bin/env python3
"""A/B timing of encoder_bwd build variants and of its phases (GPU box; the variant .so files are built beforehand in
the container: `python tools/exp/ablate_encoder_bwd.py --build`, they travel with the snapshot).
Variants = loop-unroll factors of the three phases (-DBWD_UNROLL_KS / _T / _KC); all variants are timed in interleaved
rounds in ONE process on the same random operands (uint8 0..255 frames)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "build")
VARIANTS = ["1 1 1", "3 1 1", "1 1 7"]


def so_of(v):
    return os.path.join(OUT, "libenc_%s.so" % v.replace(" ", "_"))


if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for v in VARIANTS:
        ks, t, kc = v.split()[:3]
        extra = []
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                       "-DUNREAL_ABLATE", "-DBWD_UNROLL_KS=" + ks, "-DBWD_UNROLL_T=" + t, "-DBWD_UNROLL_KC=" + kc] + extra + [
                                       os.path.join(ROOT, "unreal_amd/csrc/encoder.hip"), "-o", so_of(v)]))
    sys.exit(max(p.wait() for p in procs))

import torch
N = int(os.environ.get("ABL_N", 81920))
dev = "cuda:0"
pool = torch.randint(0, 256, (N * 21168,), dtype=torch.uint8, device=dev)
idx = torch.randperm(N, device=dev).to(torch.int32)
W2 = torch.randn(8192, device=dev) * .06
c1 = torch.relu(torch.randn(N * 6400, device=dev))
d2 = torch.randn(N * 2592, device=dev) * (torch.rand(N * 2592, device=dev) > 0.5)
g = [torch.zeros(n, device=dev) for n in (3072, 16, 8192, 32)]
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
libs = {v: ctypes.CDLL(so_of(v)) for v in VARIANTS if os.path.exists(so_of(v))}


def run(lib, ph):
    lib.exp_encoder_bwd_phases(ph, N, P(pool.data_ptr()), P(idx.data_ptr()), ctypes.c_float(1.0 / 255), P(W2.data_ptr()),
                               P(c1.data_ptr()), P(d2.data_ptr()), P(g[0].data_ptr()), P(g[1].data_ptr()),
                               P(g[2].data_ptr()), P(g[3].data_ptr()), P(st))


def timed(lib, ph, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(lib, ph)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for v, lib in libs.items():
    run(lib, 7)
torch.cuda.synchronize()
res = {v: [] for v in libs}
for rnd in range(5):
    for v, lib in libs.items():
        res[v].append(timed(lib, 7))
for v in libs:
    r = sorted(res[v])
    print("KS T KC = %s   all phases: median %.3f ms  min %.3f ms" % (v, r[len(r) // 2], r[0]))
best = min(libs, key=lambda v: sorted(res[v])[len(res[v]) // 2])
for ph in (0, 1, 2, 4):
    print("variant %s phases=%d  %.3f ms" % (best, ph, timed(libs[best], ph)))

if "--stamps" in sys.argv:
    lib = libs[best]
    buf = (ctypes.c_ulonglong * 64)()
    lib.exp_read_stamps(buf, 1)
    run(lib, 7)
    lib.exp_read_stamps(buf, 0)
    names = ["stage planes (tail) + loop", "DMA issue", "wait S0", "phase 1", "wait S1", "phase 2", "wait frame DMA",
             "wait S2", "prefetch issue + FR reads", "wait S2a", "u8->bf16 + writes", "wait S2b", "phase 3", "wait S3"]
    iters = (N + 511) // 512
    for w in range(4):
        tot = sum(buf[w * 16 + k] for k in range(14))
        print("wave %d of workgroup 3: %d ticks per frame" % (w, tot // iters))
        for k in range(14):
            print("   %-28s %5.1f %%  %7d" % (names[k], 100.0 * buf[w * 16 + k] / max(tot, 1), buf[w * 16 + k] // iters))
