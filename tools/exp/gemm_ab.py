#!/usr/bin/env python3
"""Split-GEMM A/B on the trainer's production shapes (GPU box).  Variants are whole-library builds with extra -D flags
(`python tools/exp/gemm_ab.py --build` in the container; they travel with the snapshot); every variant is timed in
interleaved rounds in ONE process through unreal_amd.ops (the library behind ops is swapped between timings), outputs are
compared with variant `base`."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "exp", "build")
VARIANTS = {"base": []}
# round 4 timed: dbuf = -DSPLIT_NT_DBUF=1,-DSPLIT_TN_DBUF=1; occ3 = -DSPLIT_NT_OCC3=1 (adopted); k256 = -DSPLIT_NT_K256=1;
# n256 = -DSPLIT_NT_N256=1; ragged = -DSPLIT_FORCE_RAGGED=1; and the compiler's scheduling strategies below
for _st in ("gcn-max-ilp", "gcn-max-memory-clause", "gcn-iterative-ilp", "gcn-iterative-minreg"):
    VARIANTS[_st.replace("gcn-", "")] = ["-mllvm", "-amdgpu-sched-strategy=" + _st]
# earlier rounds of this script (results in profiles/r03_gemm_nt_ablate.log): timing-only ablations "nosplit" / "noload" /
# "nosplit_noload" = -DSPLIT_ABLATE=2 / 1 / 3 (pass them as name=-DFLAG on the command line); the direct-store epilogue
# and the 128 x 256 tile variants were removed from the kernel after they lost
for a in sys.argv[1:]:             # extra variants: name=-DX=1,-DY=2
    if "=" in a and not a.startswith("--"):
        k, v = a.split("=", 1)
        VARIANTS[k] = v.split(",")


def so_of(v):
    return os.path.join(OUT, "libunreal_%s.so" % v)


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


dev = "cuda:0"
torch.manual_seed(0)
use("base")
SHAPES = [  # name, M, N, K, flags
    ("fc1 dgrad   81920 x 2592 x 256 (relu bits)", 81920, 2592, 256, "bits"),
    ("pc_fc1 fwd  81920 x 2592 x 256 (bias, relu)", 81920, 2592, 256, "relu"),
    ("fc1 fwd    163840 x 256 x 2592 (bias, relu)", 163840, 256, 2592, "relu"),
    ("lstm dgrad  81920 x 256 x 1024", 81920, 256, 1024, ""),
    ("pc_fc1 dgrad 81920 x 256 x 2592", 81920, 256, 2592, ""),
    ("fc rollout   4096 x 256 x 2592 (bias, relu)", 4096, 256, 2592, "relu"),
    ("fc group      512 x 256 x 2592 (bias, relu)", 512, 256, 2592, "relu"),
    ("bptt-shaped  4096 x 256 x 1024", 4096, 256, 1024, ""),
]
for name, M, N, K, fl in SHAPES:
    A = torch.randn(M * K, device=dev)
    Wsrc = torch.randn(N * K, device=dev) * 0.05
    outs = {}
    amax = torch.zeros(1, device=dev)
    ops.absmax(M, K, A, K, amax)
    bias = torch.randn(N, device=dev) * 0.1
    bits = torch.randint(-32768, 32767, (M * ((N + 15) // 16),), dtype=torch.int16, device=dev) if fl == "bits" else None
    res = {v: [] for v in libs}
    Ws = {}
    for v in libs:
        use(v)
        Ws[v] = ops.SplitWeights(Wsrc, N, K, K, False)
        outs[v] = torch.zeros(M * N, device=dev)

    def run(v):
        use(v)
        if fl == "bits":
            ops.gemm_split_nt(M, N, K, A, K, Ws[v], outs[v], N, mask=bits, ldm=(N + 15) // 16, flags=ops.GEMM_RELU_BITS, a_max=amax)
        elif fl == "relu":
            ops.gemm_split_nt(M, N, K, A, K, Ws[v], outs[v], N, bias=bias, flags=ops.GEMM_RELU, a_max=amax)
        else:
            ops.gemm_split_nt(M, N, K, A, K, Ws[v], outs[v], N, a_max=amax)

    for v in libs:
        run(v)
    torch.cuda.synchronize()
    for v in libs:
        if v != "base":
            print("   %s: max |d| vs base %.3e" % (v, float((outs[v] - outs["base"]).abs().max())))
    for rnd in range(5):
        for v in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 3 * 1e3)
    for v in libs:
        r = sorted(res[v])
        print("%-48s %-10s median %8.1f us  min %8.1f us  (%.0f fp32-equivalent TFLOP/s)" % (
            name, v, r[2], r[0], 2.0 * M * N * K / r[2] / 1e6))
    del A, Wsrc, outs, Ws
    torch.cuda.empty_cache()

# ---- wgrad (TN) shapes of the trainer: C[M,N] += A[K,M]^T B[K,N], split-K as the trainer picks it
from unreal_amd.model.model import _splitk  # noqa: E402
for name, M, N, K in [("fc1 wgrad   2592 x 256, K 81920", 2592, 256, 81920), ("lstm wgrad  256 x 1024, K 81920", 256, 1024, 81920),
                      ("pc_fc1 wgrad 256 x 2592, K 81920", 256, 2592, 81920), ("fc1 wgrad   2592 x 256, K 163840", 2592, 256, 163840)]:
    A = torch.randn(K * M, device=dev)
    B = torch.randn(K * N, device=dev) * 0.01
    sa, sb = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    use("base")
    ops.absmax(K, M, A, M, sa); ops.absmax(K, N, B, N, sb)
    sk = _splitk(M, N, K)
    outs = {v: torch.zeros(M * N, device=dev) for v in libs}
    res = {v: [] for v in libs}

    def run_tn(v):
        use(v)
        ops.gemm_split_tn(M, N, K, A, M, B, N, outs[v], N, splitk=sk, a_max=sa, b_max=sb)

    for v in libs:
        run_tn(v)
    torch.cuda.synchronize()
    for v in libs:
        if v != "base":
            print("   %s: max |d| vs base %.3e (split-K atomics: order-dependent last bits)" % (v, float((outs[v] - outs["base"]).abs().max())))
    for rnd in range(5):
        for v in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run_tn(v)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 3 * 1e3)
    for v in libs:
        r = sorted(res[v])
        print("%-48s %-10s median %8.1f us  min %8.1f us  (%.0f fp32-equivalent TFLOP/s)" % (name, v, r[2], r[0], 2.0 * M * N * K / r[2] / 1e6))
    del A, B, outs
    torch.cuda.empty_cache()
