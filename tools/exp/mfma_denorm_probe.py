#!/usr/bin/env python3
"""fp16 subnormal operands on v_mfma_f32_16x16x32_f16 (GPU box).  Build in the container first:
hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/exp/mfma_denorm_probe.hip -o tools/exp/build/libdenorm_probe.so"""
import ctypes, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "build", "libdenorm_probe.so"))
rng = np.random.RandomState(0)
P = ctypes.c_void_p
for name, a_bits in (("bytes as fp16 subnormals (A = b * 2^-24)", rng.randint(0, 256, (16, 32)).astype(np.uint16)),
                     ("10-bit subnormals", rng.randint(0, 1024, (16, 32)).astype(np.uint16))):
    for side in ("A", "B"):
        w = (rng.randn(32, 16) * 37.0).astype(np.float16)
        A = torch.tensor(a_bits.astype(np.int16)).cuda()
        B = torch.tensor(w.view(np.int16)).cuda()
        D = torch.zeros(16, 16, device="cuda")
        if side == "A":
            lib.denorm_probe(P(A.data_ptr()), P(B.data_ptr()), P(D.data_ptr()), P(torch.cuda.current_stream().cuda_stream))
            want = (a_bits.astype(np.float64) * 2.0 ** -24) @ w.astype(np.float64)
        else:       # subnormals as the B operand: D = W^T[16][32] x bits^T[32][16]
            At = torch.tensor(np.ascontiguousarray(w.T).view(np.int16)).cuda()
            Bt = torch.tensor(np.ascontiguousarray(a_bits.T).astype(np.int16)).cuda()
            lib.denorm_probe(P(At.data_ptr()), P(Bt.data_ptr()), P(D.data_ptr()), P(torch.cuda.current_stream().cuda_stream))
            want = w.T.astype(np.float64) @ (a_bits.T.astype(np.float64) * 2.0 ** -24)
        torch.cuda.synchronize()
        got = D.cpu().numpy().astype(np.float64)
        err = np.abs(got - want).max() / np.abs(want).max()
        print("%s, subnormal operand = %s: max |err| / max |want| = %.3e   (all-zero result: %s)" % (
            name, side, err, bool((got == 0).all())))

# which single products survive?  A[i][0] = subnormal bit pattern (1, 3, 255, 1023), B[0][col] = 2^e (incl. fp16 subnormals)
print("single products  subnormal(bits) x 2^e : got / want  (1.0 = exact, 0 = flushed)")
for bits in (1, 3, 255, 1023):
    for e0 in (-24, -8):
        a_bits = np.zeros((16, 32), dtype=np.uint16)
        a_bits[:, 0] = bits
        es = np.arange(e0, e0 + 16)
        w = np.zeros((32, 16), dtype=np.float16)
        w[0, :] = (2.0 ** es).astype(np.float16)
        A = torch.tensor(a_bits.astype(np.int16)).cuda()
        B = torch.tensor(w.view(np.int16)).cuda()
        D = torch.zeros(16, 16, device="cuda")
        lib.denorm_probe(P(A.data_ptr()), P(B.data_ptr()), P(D.data_ptr()), P(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        want = bits * 2.0 ** -24 * w[0, :].astype(np.float64)
        got = D.cpu().numpy()[0].astype(np.float64)
        print("  bits %4d, e = %3d..%3d: " % (bits, es[0], es[-1]) + " ".join("%.3g" % (g / x) for g, x in zip(got, want)))
# a sum of many small products beside one large one: is the small part kept to fp32 accuracy?
a_bits = np.ones((16, 32), dtype=np.uint16)
for big in (0.0, 1.0, 1024.0, 32768.0 - 16):
    w = np.full((32, 16), 2.0 ** -3, dtype=np.float16)
    w[0, :] = big
    A = torch.tensor(a_bits.astype(np.int16)).cuda()
    B = torch.tensor(w.view(np.int16)).cuda()
    D = torch.zeros(16, 16, device="cuda")
    lib.denorm_probe(P(A.data_ptr()), P(B.data_ptr()), P(D.data_ptr()), P(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    want = float((w[:, 0].astype(np.float64) * 2.0 ** -24).sum())
    got = float(D[0, 0])
    print("  32 products of 2^-24 with (%g, 31 x 2^-3): got %.9e want %.9e  rel err %.2e" % (big, got, want, abs(got - want) / want))
