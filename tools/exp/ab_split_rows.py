#!/usr/bin/env python3
"""Same-process A/B of the NT split GEMM with and without the bank-spreading row deal of its LDS stores
(-DSPLIT_ROW_DEAL=0/1).  `--build` in the container, run on the GPU box."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "exp", "build")
if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    ps = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSPLIT_ROW_DEAL=%d" % v, "-DSPLIT_TN_DEAL=%d" % v,
                            os.path.join(ROOT, "unreal_amd/csrc/gemm_split.hip"), "-o", os.path.join(OUT, "libsplit_deal%d.so" % v)]) for v in (0, 1)]
    sys.exit(max(p.wait() for p in ps))
import torch
libs = {v: ctypes.CDLL(os.path.join(OUT, "libsplit_deal%d.so" % v)) for v in (0, 1)}
SHAPES = (("fc_fwd", 81920, 256, 2592), ("pc_fc1", 81920, 2592, 256), ("d_fc", 81920, 256, 1024), ("lstm_x", 81920, 1024, 261),
          ("fc_roll", 4096, 256, 2592), ("dh_rec", 4096, 256, 1024), ("lstm_h", 4096, 1024, 256))
P = ctypes.c_void_p
for name, M, N, K in SHAPES:
    lda = (K + 3) // 4 * 4; ldw = (K + 31) // 32 * 32
    A = torch.randn(M * lda, device="cuda"); C = torch.zeros(M * N, device="cuda")
    B = torch.randn(3 * N * ldw, device="cuda").to(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    res = {0: [], 1: []}
    for rnd in range(5):
        for v, lib in libs.items():
            run = lambda: lib.unreal_gemm_f32_split_nt(M, N, K, P(A.data_ptr()), lda, P(B.data_ptr()), ldw, ctypes.c_long(N * ldw), P(C.data_ptr()), N, None, None, 0, 0, 1, P(st))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 5)
    m = {v: sorted(r)[2] for v, r in res.items()}
    print("%-8s M=%d N=%d K=%d: lane-order rows %.4f ms   dealt rows %.4f ms   (%.1f %%)" % (name, M, N, K, m[0], m[1], 100 * (m[1] / m[0] - 1)), flush=True)
    del A, B, C
for name, M, N, K, sk in (("tn_fc1", 2592, 256, 81920, 24), ("tn_lstm", 256, 1024, 81920, 64), ("tn_pc", 256, 2592, 81920, 24)):
    A = torch.randn(K * M, device="cuda"); B = torch.randn(K * N, device="cuda"); C = torch.zeros(M * N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    res = {0: [], 1: []}
    for rnd in range(5):
        for v, lib in libs.items():
            run = lambda: lib.unreal_gemm_f32_split_tn(M, N, K, P(A.data_ptr()), M, P(B.data_ptr()), N, P(C.data_ptr()), N, None, sk, P(st))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 5)
    m = {v: sorted(r)[2] for v, r in res.items()}
    print("%-8s M=%d N=%d K=%d: lane order %.4f ms   dealt %.4f ms   (%.1f %%)" % (name, M, N, K, m[0], m[1], 100 * (m[1] / m[0] - 1)), flush=True)
    del A, B, C
