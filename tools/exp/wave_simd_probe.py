#!/usr/bin/env python3
"""Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID.SIMD_ID per wave; GPU box only.)"""
import ctypes, os, subprocess
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
src = os.path.join(OUT, "probe.hip")
open(src, "w").write(r'''
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(512) void probe(int* out) {
  __shared__ char big[100 * 1024];          // one workgroup per CU, like the encoder kernels
  big[threadIdx.x] = 0;
  int simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);     // HW_ID[5:4]
  int cu = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4);       // HW_ID[11:8]
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = simd | (cu << 8) | (big[1] << 20);
}
extern "C" int run(int* out, int blocks) { hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, out); return (int)hipDeviceSynchronize(); }
''')
so = os.path.join(OUT, "probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", src, "-o", so])
lib = ctypes.CDLL(so)
out = torch.zeros(8 * 8, dtype=torch.int32, device="cuda")
lib.run(ctypes.c_void_p(out.data_ptr()), 8)
o = out.cpu().numpy().reshape(8, 8)
for b in range(8):
    print("block %d: SIMD of waves 0..7 =" % b, [int(v) & 3 for v in o[b]], " CU", int(o[b][0]) >> 8 & 15)
