#!/usr/bin/env python3
"""Micro-benchmarks of the heavy kernels at the shapes bench.py drives them with (B = 4096, T = 20).
Run on the GPU box:  python tools/bench_kernels.py [filter]
Prints average launch time (HIP events, 5 reps after 2 warm-ups) and achieved fp32 TFLOP/s or GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unreal_amd import ops  # noqa: E402

DEV = "cuda:0"
FILT = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, flop=None, bytes_=None):
    s = "%-46s %9.3f ms" % (name, ms)
    if flop:
        s += "  %7.1f TFLOP/s (%4.1f%% of 157.3)" % (flop / ms / 1e9, flop / ms / 1e9 / 1.573)
    if bytes_:
        s += "  %7.1f GB/s" % (bytes_ / ms / 1e6)
    print(s, flush=True)


def rnd(*shape):
    return torch.randn(*shape, device=DEV)


def gemm_case(name, ta, tb, M, N, K, flags=0, splitk=1, mask=False, bias=False, ldc=None):
    if FILT and FILT not in name and FILT != "gemm":
        return
    lda = (M if ta else K)
    ldb = (K if tb else N)
    A = rnd((K if ta else M) * lda)
    B = rnd((N if tb else K) * ldb)
    ldc = ldc or N
    C = torch.zeros(M * ldc, device=DEV)
    mk = rnd(M * N) if mask else None
    bs = rnd(N) if bias else None
    f = lambda: ops.gemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=bs, mask=mk, ldm=N if mask else 0,
                         flags=flags, splitk=splitk)
    report("gemm %s M=%d N=%d K=%d sk=%d" % (name, M, N, K, splitk), timeit(f), flop=2.0 * M * N * K)


def main():
    R, B = 81920, 4096
    from unreal_amd.model.model import _splitk
    gemm_case("NN fc_fwd", 0, 0, R, 256, 2592, flags=ops.GEMM_RELU, bias=True, ldc=264)
    gemm_case("NN lstm_x", 0, 0, R, 1024, 261)
    gemm_case("NN pc_fc1", 0, 0, R, 2592, 256, flags=ops.GEMM_RELU, bias=True)
    gemm_case("NN fc_fwd_roll", 0, 0, B, 256, 2592, flags=ops.GEMM_RELU, bias=True, ldc=264)
    gemm_case("NN lstm_x_roll", 0, 0, B, 1024, 261)
    gemm_case("NN lstm_h_step", 0, 0, B, 1024, 256, flags=ops.GEMM_ACCUM)
    gemm_case("TN dW_fc1", 1, 0, 2592, 256, R, flags=ops.GEMM_ATOMIC, splitk=_splitk(2592, 256, R))
    gemm_case("TN dW_lstm_x", 1, 0, 261, 1024, R, flags=ops.GEMM_ATOMIC, splitk=_splitk(261, 1024, R))
    gemm_case("TN dW_lstm_h", 1, 0, 256, 1024, R - B, flags=ops.GEMM_ATOMIC, splitk=_splitk(256, 1024, R - B))
    gemm_case("TN dW_pc_fc1", 1, 0, 256, 2592, R, flags=ops.GEMM_ATOMIC, splitk=_splitk(256, 2592, R))
    gemm_case("NT d_fc", 0, 1, R, 256, 1024, flags=ops.GEMM_RELU_MASK, mask=True)
    gemm_case("NT d_f2", 0, 1, R, 2592, 256, flags=ops.GEMM_RELU_MASK, mask=True)
    gemm_case("NT d_feat_pc", 0, 1, R, 256, 2592)
    gemm_case("NT dh_rec_step", 0, 1, B, 256, 1024)

    if not FILT or FILT in "split gemm":
        for name, M, N, K in (("fc_fwd", R, 256, 2592), ("lstm_x", R, 1024, 261), ("pc_fc1", R, 2592, 256),
                              ("d_fc", R, 256, 1024), ("d_f2", R, 2592, 256), ("fc_roll", B, 256, 2592),
                              ("lstm_h_step", B, 1024, 256), ("dh_rec_step", B, 256, 1024)):
            lda = (K + 3) // 4 * 4
            A_ = rnd(M * lda); Bt_ = ops.SplitWeights(rnd(N * K), N, K, K, False); C_ = torch.zeros(M * N, device=DEV)
            report("split_nt %s M=%d N=%d K=%d" % (name, M, N, K),
                   timeit(lambda: ops.gemm_split_nt(M, N, K, A_, lda, Bt_, C_, N)), flop=2.0 * M * N * K)
            if M == B and K >= 1024:
                for sk in (2, 4, 8):
                    report("split_nt %s splitk=%d (atomic)" % (name, sk),
                           timeit(lambda: ops.gemm_split_nt(M, N, K, A_, lda, Bt_, C_, N, flags=ops.GEMM_ATOMIC, splitk=sk)),
                           flop=2.0 * M * N * K)
            del A_, Bt_, C_

    if not FILT or FILT in "split gemm lstm":
        K_x, xld = 261, 264
        Wk = rnd((K_x + 256) * 1024) * .05
        sh_x = ops.SplitWeights(Wk, K_x, 1024, 1024, True)
        sh_h = ops.SplitWeights(Wk, 256, 1024, 1024, True, offset=K_x * 1024, row_perm=1)
        sh_xh = ops.LstmKernelShadow(Wk, K_x)
        x_, h_, c_, b_ = rnd(B * xld), rnd(B * 256), rnd(B * 256), rnd(1024)
        g_, c2, h2 = torch.zeros(B * 1024, device=DEV), torch.zeros(B * 256, device=DEV), torch.zeros(B * 256, device=DEV)

        def chain():
            ops.gemm_split_nt(B, 1024, K_x, x_, xld, sh_x, g_, 1024)
            ops.lstm_step_fwd(B, h_, sh_h, g_, b_, c_, c2, h2)
        report("lstm step, chain (x GEMM + h step) rows=%d" % B, timeit(chain), flop=2.0 * B * 1024 * (K_x + 256))
        report("lstm step, h half only rows=%d" % B, timeit(lambda: ops.lstm_step_fwd(B, h_, sh_h, g_, b_, c_, c2, h2)),
               flop=2.0 * B * 1024 * 256)
        report("lstm step, whole kernel rows=%d" % B,
               timeit(lambda: ops.lstm_step_fwd(B, h_, sh_xh, g_, b_, c_, c2, h2, x=x_, ldx=xld, Kx=K_x)),
               flop=2.0 * B * 1024 * (K_x + 256))

    if not FILT or FILT in "split gemm tn":
        for name, M, N, K in (("dW_fc1", 2592, 256, R), ("dW_lstm_x", 256, 1024, R), ("dW_lstm_h", 256, 1024, R - B),
                              ("dW_pc_fc1", 256, 2592, R)):
            A_ = rnd(K * M); B_ = rnd(K * N); C_ = torch.zeros(M * N, device=DEV)
            sk = _splitk(M, N, K)
            report("split_tn %s M=%d N=%d K=%d sk=%d" % (name, M, N, K, sk),
                   timeit(lambda: ops.gemm_split_tn(M, N, K, A_, M, B_, N, C_, N, splitk=sk)), flop=2.0 * M * N * K)
            del A_, B_, C_

    if not FILT or FILT in "encoder":
        for N in (R, B):
            pool = torch.randint(0, 2, (N * ops.FRAME_BYTES,), dtype=torch.uint8, device=DEV)
            idx = torch.randperm(N, device=DEV).to(torch.int32)
            W1, b1, W2, b2 = rnd(3072) * .07, rnd(16) * .07, rnd(8192) * .06, rnd(32) * .06
            f2 = torch.zeros(N * 2592, device=DEV)
            c1 = torch.zeros(N * 6400, device=DEV)
            report("encoder_fwd N=%d (save c1)" % N,
                   timeit(lambda: ops.encoder_fwd(pool, idx, 1.0, W1, b1, W2, b2, f2, c1)), flop=2 * 1892352.0 * N,
                   bytes_=N * (21168 + 10368 + 25600.0))
            report("encoder_fwd N=%d (no c1)" % N,
                   timeit(lambda: ops.encoder_fwd(pool, idx, 1.0, W1, b1, W2, b2, f2, None)), flop=2 * 1892352.0 * N,
                   bytes_=N * (21168 + 10368.0))
            d2 = rnd(N * 2592)
            g = [torch.zeros(n, device=DEV) for n in (3072, 16, 8192, 32)]
            report("encoder_bwd N=%d" % N,
                   timeit(lambda: ops.encoder_bwd(pool, idx, 1.0, W2, c1, d2, *g)), flop=2 * 2555904.0 * N,
                   bytes_=N * (21168 + 10368 + 25600.0))
            del pool, f2, c1, d2

    if not FILT or FILT in "pc":
        N, A = R, 4
        hp = torch.relu(rnd(N * 2592))
        Wv, bv, Wa, ba = rnd(512) * .04, rnd(1), rnd(2048) * .04, rnd(4)
        qmax = torch.zeros(N * 400, device=DEV)
        s_hp, s_dd = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)      # absmax slots (the producers' in the trainer)
        ops.absmax(N, 2592, hp, 2592, s_hp)
        report("pc_deconv_fwd qmax N=%d" % N, timeit(lambda: ops.pc_deconv_fwd(N, A, hp, Wv, bv, Wa, ba, qmax=qmax, hp_max=s_hp)),
               flop=2 * 207360.0 * N, bytes_=N * (10368 + 1600.0))
        act = torch.randint(0, 4, (N,), dtype=torch.int32, device=DEV)
        tgt = rnd(N * 400)
        mask = torch.ones(N, dtype=torch.int32, device=DEV)
        d_dec = torch.zeros(N * 400 * 5, device=DEV)
        loss = torch.zeros(1, device=DEV)
        report("pc_deconv_fwd train N=%d" % N,
               timeit(lambda: ops.pc_deconv_fwd(N, A, hp, Wv, bv, Wa, ba, action=act, target=tgt, mask=mask, lam=0.05,
                                                grad_scale=1.0, d_dec=d_dec, loss=loss, hp_max=s_hp, ddec_max=s_dd)),
               flop=2 * 207360.0 * N, bytes_=N * (10368 + 1600 + 8000.0))
        d_hp = torch.zeros(N * 2592, device=DEV)
        g = [torch.zeros(n, device=DEV) for n in (512, 1, 2048, 4)]
        report("pc_deconv_bwd N=%d" % N,
               timeit(lambda: ops.pc_deconv_bwd(N, A, hp, d_dec, Wv, Wa, d_hp, *g, hp_max=s_hp, ddec_max=s_dd)), flop=2 * 2 * 207360.0 * N,
               bytes_=N * (10368 * 2 + 8000.0))
        report("pc_deconv_train N=%d (forward loss + backward, one launch)" % N,
               timeit(lambda: ops.pc_deconv_train(N, A, hp, Wv, bv, Wa, ba, act, tgt, mask, 0.05, 1.0, loss, d_hp, *g,
                                                  hp_max=s_hp)), flop=3 * 2 * 207360.0 * N, bytes_=N * (10368 * 2 + 1600.0))

    if not FILT or FILT in "env":
        ring = ops.Ring(B, 8, DEV)
        ops.maze_reset(ring)
        acts = torch.randint(0, 4, (B,), dtype=torch.int32, device=DEV)
        report("maze_step B=%d" % B, timeit(lambda: ops.maze_step(ring, acts), reps=20), bytes_=B * (21168 + 1600.0))
        n = 1898877 // 4 * 4
        v, ms, mom, gr = rnd(n), torch.ones(n, device=DEV), torch.zeros(n, device=DEV), rnd(n)
        sc, nm = torch.zeros(256, device=DEV), torch.zeros(1, device=DEV)
        report("grad_norm + rmsprop", timeit(lambda: (ops.grad_norm(gr, sc, nm),
                                                      ops.rmsprop_step(v, ms, mom, gr, 1e-3, .99, 0., .1, 40., nm))),
               bytes_=n * 32.0)


if __name__ == "__main__":
    main()
