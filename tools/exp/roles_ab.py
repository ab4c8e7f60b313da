#!/usr/bin/env python3
"""encoder_bwd: round-2 kernel (one frame per 256-thread workgroup, two per CU) vs the role-specialised round-3 kernel
(one 512-thread workgroup per CU: 4 consumer + 4 producer waves), variants (P1ALL, P3ALL).  GPU box; build in the
container first: python tools/exp/roles_ab.py --build.  Checks the new kernel's outputs against the old kernel's on the
same random operands (both are fp32-grade: agreement to ~1e-5 of the largest element), then times all of them in
interleaved rounds in one process."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(ROOT, "tools", "exp", "build", os.environ.get("ROLES_SO", "libenc_roles.so"))
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    sys.exit(subprocess.call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] +
                             os.environ.get("ROLES_FLAGS", "").split() + [os.path.join(ROOT, "tools/exp/encoder_ablate.hip"), "-o", SO]))
import torch
N = int(os.environ.get("ABL_N", 81920))
dev = "cuda:0"
torch.manual_seed(0)
pool = torch.randint(0, 256, (N * 21168,), dtype=torch.uint8, device=dev)
idx = torch.randperm(N, device=dev).to(torch.int32)
W2 = torch.randn(8192, device=dev) * .06
c1 = torch.relu(torch.randn(N * 6400, device=dev))
d2 = torch.randn(N * 2592, device=dev) * (torch.rand(N * 2592, device=dev) > 0.5)
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
lib = ctypes.CDLL(SO)
DESC = {0: "P2C=7 P3ALL", 1: "P2C=6 P3ALL", 2: "P2C=5 P3ALL", 3: "P2C=7 P3 consumers only"}
VARIANTS = [int(v) for v in os.environ.get("ROLES_VARIANTS", "0,1,2,3").split(",")]


def grads():
    return [torch.zeros(n, device=dev) for n in (3072, 16, 8192, 32)]


c1_max = c1.abs().max().reshape(1).contiguous()       # absmax slots (fp16x2 planes: per-tensor scales)
d2_max = d2.abs().max().reshape(1).contiguous()


def run(which, g, n=N):
    args = (n, P(pool.data_ptr()), P(idx.data_ptr()), ctypes.c_float(1.0 / 255), P(W2.data_ptr()), P(c1.data_ptr()),
            P(d2.data_ptr()), P(g[0].data_ptr()), P(g[1].data_ptr()), P(g[2].data_ptr()), P(g[3].data_ptr()))
    rc = lib.exp_encoder_bwd_phases(7, *args, P(st)) if which < 0 else lib.exp_encoder_bwd_roles(
        which, *args, P(c1_max.data_ptr()), P(d2_max.data_ptr()), P(st))
    assert rc == 0, rc


ok = True
for n in (() if os.environ.get("ROLES_NOCHECK") else (1, 3, 255, 256, 257, 1300, N)):            # ragged: fewer frames than workgroups, one extra, odd trip counts
    ref = grads(); run(-1, ref, n); torch.cuda.synchronize()
    for v in VARIANTS:
        g = grads(); run(v, g, n); torch.cuda.synchronize()
        for name, a, b in zip(("dW1", "db1", "dW2", "db2"), g, ref):
            err = float((a - b).abs().max()); sc = float(b.abs().max())
            good = err <= 2e-5 * sc + 1e-30 and bool(torch.isfinite(a).all())
            ok &= good
            if not good or n == N:
                print("N=%6d variant %d %s: max|d| %.3e of %.3e %s" % (n, v, name, err, sc, "ok" if good else "MISMATCH"))
print("CORRECT" if ok else "WRONG")


def timed(which, reps=3):
    g = grads()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(which, g)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {v: [] for v in [-1] + VARIANTS}
for rnd in range(5):
    for v in res:
        res[v].append(timed(v))
for v, r in res.items():
    r = sorted(r)
    print("%-44s median %.3f ms  min %.3f ms" % ("round-2 kernel" if v < 0 else "roles " + DESC[v], r[2], r[0]))
if "--stamps" in sys.argv:
    names = {0: "loop top", 1: "wait A", 2: "phase 1 (+P1 share)", 3: "wait S1", 4: "phase 2 (share)", 5: "wait B", 6: "phase 3 (share)",
             7: "image", 8: "c1 planes", 9: "d2 planes", 10: "load issue", 11: "seg2 tail"}
    for var in (100, 102):
        buf = (ctypes.c_ulonglong * 128)()
        lib.exp_read_rstamps(buf, 1)
        g = grads(); run(var, g)
        lib.exp_read_rstamps(buf, 0)
        iters = (N - 3 + 255) // 256
        print("variant %d: cycles per frame (workgroup 3, %d frames)" % (var, iters))
        for w in range(8):
            tot = sum(buf[w * 16 + k] for k in range(16))
            print("  wave %d total %.0f: " % (w, tot / iters) + ", ".join("%s %.0f" % (names[k], buf[w * 16 + k] / iters) for k in range(12) if buf[w * 16 + k]))
sys.exit(0 if ok else 1)
