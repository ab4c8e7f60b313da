#!/usr/bin/env python3
"""one experiment on the 6-pass tax of the split-operand GEMMs.
fp32 = hi + lo with BOTH terms fp16 and a power-of-two scale per tensor (max |x| -> [2^14, 2^15)); products hh, hl, lh on
v_mfma_f32_32x32x16_f16 (3 passes instead of the 6 of bf16x3).  HARD GATE: on the trainer's LIVE operands its RMS and max
error against fp64 must be <= the plain-fp32-MFMA kernel's (unreal_gemm_f32) on the same data; then speed.
GPU box.  Build in the container first: python tools/exp/f16x2_gate.py --build   (two builds of csrc/gemm_split.hip,
-DSPLIT_F16=0 / 1; the product library is untouched)."""
import ctypes, json, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "build")
if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    ps = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                            "-DSPLIT_NT_MODE=%d" % m, "-DSPLIT_TN_MODE=%d" % m,
                            os.path.join(ROOT, "unreal_amd/csrc/gemm_split.hip"), "-o", os.path.join(OUT, "libsplit_f16_%d.so" % m)]) for m in (0, 1)]
    sys.exit(max(p.wait() for p in ps))
sys.path.insert(0, ROOT)
import argparse
import torch
from bench import build_trainer
from unreal_amd import ops
from unreal_amd.model.model import _splitk
dev = torch.device("cuda", 0)
P = ctypes.c_void_p
libs = {m: ctypes.CDLL(os.path.join(OUT, "libsplit_f16_%d.so" % m)) for m in (0, 1)}
# extra builds of the wgrad kernel: "label:mode:file" (tools/exp/build), e.g. accumulator-flush variants
EXTRA_TN = [x.split(":") for x in os.environ.get("GATE_TN_LIBS", "").split(",") if x]
st = torch.cuda.current_stream().cuda_stream
ACTORS = int(os.environ.get("GATE_ACTORS", 4096))
flags, net, tr = build_trainer(argparse.Namespace(actors=ACTORS, history=40), 0, 1, dev)
while not tr._full:
    tr.process(None, 0)
for _ in range(3):
    tr.process(None, 0)
TRIALS = int(os.environ.get("GATE_TRIALS", 3))      # live operands of TRIALS different trainer states
TRIAL = int(os.environ.get("GATE_TRIAL", 0))
for _ in range(2 * TRIAL):
    tr.process(None, 0)
net.refresh_shadows()
net.begin_pass()
tr._rollout(); net.grads.flat.zero_(); tr.losses.zero_(); tr._train_base()
torch.cuda.synchronize()
rows = tr.n_step_TD * ACTORS
ws, gws = tr.base_ws, tr.gws
f2 = ws.f2[:rows * 2592].view(rows, 2592)
d_fc = gws.d_fc[:rows * 256].view(rows, 256)
d_gates = gws.d_gates[:rows * 1024].view(rows, 1024)
W_fc1 = net.params.shaped("W_base_fc1")                 # [2592, 256]
W_l = net.params.shaped("lstm_kernel")[:256]            # [256, 1024]: fc rows of the LSTM kernel


def slot_of(t):
    """absmax slot of a tensor (device float holding max |x|), reduced by the library's own kernel"""
    s = torch.zeros(1, device=dev)
    t2 = t if t.dim() == 2 else t.view(1, -1)
    assert libs[1].unreal_absmax_f32(t2.shape[0], t2.shape[1], P(t2.data_ptr()), t2.stride(0), P(s.data_ptr()), P(st)) == 0
    return s


KEEP = []      # slots must outlive the launches that read them


def planes(lib, mode, W, transpose):
    """weight shadow [N][Kpad] planes of W (rows x cols fp32): bf16x3 (round 2) or fp16x2 with the weight's absmax slot"""
    r, c = W.shape
    orows, ocols = (c, r) if transpose else (r, c)
    ld = (ocols + 31) // 32 * 32
    npl = 2 if mode else 3
    buf = torch.zeros(npl * orows * ld, dtype=torch.int16, device=dev)
    wmax = slot_of(W)
    KEEP.append(wmax)
    if mode:
        rc = lib.unreal_split_f16x2(r, c, P(W.data_ptr()), W.stride(0), int(transpose), 0, P(buf.data_ptr()), ld, ctypes.c_long(orows * ld),
                                    P(wmax.data_ptr()), P(st))
    else:
        rc = lib.unreal_split_bf16x3(r, c, P(W.data_ptr()), W.stride(0), int(transpose), 0, P(buf.data_ptr()), ld, ctypes.c_long(orows * ld), P(st))
    assert rc == 0
    return buf, ld, orows * ld, wmax


def nt(lib, mode, A, W, transpose, N, K):
    M = A.shape[0]
    buf, ld, plane, wmax = planes(lib, mode, W, transpose)
    amax = slot_of(A)
    KEEP.append(amax)
    C = torch.empty(M, N, device=dev)
    def run():
        rc = lib.unreal_gemm_f32_split_nt(M, N, K, P(A.data_ptr()), A.stride(0), P(amax.data_ptr()), P(buf.data_ptr()), ld, ctypes.c_long(plane),
                                          P(wmax.data_ptr()), P(C.data_ptr()), N, None, None, None, 0, 0, 1, P(st))
        assert rc == 0, rc
    return C, run


def tn(lib, mode, A, B):
    K, M = A.shape; N = B.shape[1]
    sa, sb = slot_of(A), slot_of(B)
    KEEP.extend([sa, sb])
    C = torch.zeros(M, N, device=dev)
    sk = _splitk(M, N, K)
    def run():
        C.zero_()
        rc = lib.unreal_gemm_f32_split_tn(M, N, K, P(A.data_ptr()), A.stride(0), P(sa.data_ptr()), P(B.data_ptr()), B.stride(0), P(sb.data_ptr()),
                                          P(C.data_ptr()), N, None, sk, P(st))
        assert rc == 0
    return C, run


def ref64(A, B, tA=False):
    out = None
    if tA:
        out = torch.zeros(A.shape[1], B.shape[1], dtype=torch.float64, device=dev)
        for r0 in range(0, A.shape[0], 8192):
            out += A[r0:r0 + 8192].double().t() @ B[r0:r0 + 8192].double()
        return out
    return torch.cat([A[r0:r0 + 8192].double() @ B.double() for r0 in range(0, A.shape[0], 8192)])


def timed(run, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def err(C, R):
    e = (C.double() - R)
    return float(e.pow(2).mean().sqrt() / R.pow(2).mean().sqrt()), float(e.abs().max() / R.abs().max())


def tail(C, R):
    """|error| / max |R| at the 99.99th and 99.9999th percentile (the maximum of 2e8 outputs is one sample)"""
    e = (C.double() - R).abs().flatten() / R.abs().max()
    k = e.numel()
    top = torch.topk(e[::1] if k <= (1 << 27) else e[::2], max(1, int(1e-4 * min(k, 1 << 27)))).values
    return float(top[-1]), float(top[max(0, int(len(top) * 0.01) - 1)])


cases = []
# (name, kind, operands...)
cases.append(("fc forward  f2 x W_fc1      [%d x 256, K 2592]" % rows, "nt", f2, W_fc1, True, 256, 2592, lambda: ref64(f2, W_fc1)))
cases.append(("fc dgrad    d_fc x W_fc1^T  [%d x 2592, K 256]" % rows, "nt", d_fc, W_fc1, False, 2592, 256, lambda: ref64(d_fc, W_fc1.t())))
cases.append(("lstm dgrad  d_gates x Wl^T  [%d x 256, K 1024]" % rows, "nt", d_gates, W_l, False, 256, 1024, lambda: ref64(d_gates, W_l.t())))
cases.append(("fc wgrad    f2^T x d_fc     [2592 x 256, K %d]" % rows, "tn", f2, d_fc, None, None, None, lambda: ref64(f2, d_fc, tA=True)))
xc = ws.xcat[:rows * ws.xld].view(rows, ws.xld)[:, :256].contiguous()
cases.append(("lstm wgrad  fc^T x d_gates  [256 x 1024, K %d]" % rows, "tn", xc, d_gates, None, None, None, lambda: ref64(xc, d_gates, tA=True)))
report = {"actors": ACTORS, "rows": rows, "cases": []}
for name, kind, A, B, transpose, N, K, reff in cases:
    R = reff()
    res = {}
    # plain fp32 MFMA kernel of the product library (the gate's yardstick)
    if kind == "nt":
        C32 = torch.empty(A.shape[0], N, device=dev)
        Wnk = (B.t() if transpose else B).contiguous()          # [N][K]
        run32 = lambda: ops.gemm(False, True, A.shape[0], N, K, A, A.stride(0), Wnk, Wnk.stride(0), C32, N)
    else:
        C32 = torch.zeros(A.shape[1], B.shape[1], device=dev)
        sk = _splitk(A.shape[1], B.shape[1], A.shape[0])
        def run32():
            C32.zero_()
            ops.gemm(True, False, A.shape[1], B.shape[1], A.shape[0], A, A.stride(0), B, B.stride(0), C32, B.shape[1], flags=ops.GEMM_ATOMIC if hasattr(ops, "GEMM_ATOMIC") else 4, splitk=sk)
    try:
        run32(); torch.cuda.synchronize()
    except ValueError as ex:            # (a layout the plain kernel's wrapper refuses: no yardstick for this case)
        print("%s\n   skipped: %s" % (name, ex))
        continue
    res["fp32_mfma"] = err(C32, R) + (timed(run32),)
    tails = {"fp32_mfma": tail(C32, R)}
    for mode, label in ((0, "bf16x3_6pass"), (1, "f16x2_3pass")):
        C, run = (nt(libs[mode], mode, A, B, transpose, N, K) if kind == "nt" else tn(libs[mode], mode, A, B))
        run(); torch.cuda.synchronize()
        res[label] = err(C, R) + (timed(run),)
        tails[label] = tail(C, R)
    if kind == "tn":
        for label, mode, fn in EXTRA_TN:
            C, run = tn(ctypes.CDLL(os.path.join(OUT, fn)), int(mode), A, B)
            run(); torch.cuda.synchronize()
            res[label] = err(C, R) + (timed(run),)
    g = res["f16x2_3pass"][0] <= res["fp32_mfma"][0] and res["f16x2_3pass"][1] <= res["fp32_mfma"][1]
    print("%s\n   %s" % (name, "\n   ".join("%-22s rms %.3e max %.3e %.3f ms" % ((k,) + v) for k, v in res.items())))
    print("   |err| / max |R| at the 99.99 / 99.9999 percentile: " + "; ".join("%s %.3e / %.3e" % ((k,) + v) for k, v in tails.items()))
    print("   gate (f16x2 error <= fp32-MFMA error, rms AND max): %s (rms %s, max %s: ratio %.2f);  speed f16x2 vs bf16x3: %.2fx" % (
        "PASS" if g else "FAIL", "pass" if res["f16x2_3pass"][0] <= res["fp32_mfma"][0] else "fail",
        "pass" if res["f16x2_3pass"][1] <= res["fp32_mfma"][1] else "fail", res["f16x2_3pass"][1] / res["fp32_mfma"][1],
        res["bf16x3_6pass"][2] / res["f16x2_3pass"][2]))
    # dynamic range of the activation operand relative to its tensor maximum (what a per-tensor scale has to cover)
    a = A.abs().flatten()[::97]
    nz = a[a > 0]
    q = torch.quantile(torch.log2(nz / a.max()).float(), torch.tensor([0.001, 0.01, 0.5], device=dev))
    res["log2_ratio_to_max_q001_q01_q50"] = [float(x) for x in q]
    res["gate"] = bool(g)
    report["cases"].append({"name": name, **{k: (list(v) if isinstance(v, tuple) else v) for k, v in res.items()}})
print(json.dumps(report))
