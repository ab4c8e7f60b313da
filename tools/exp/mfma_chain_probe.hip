// How many cycles does v_mfma_f32_16x16x32_bf16 take per instruction when the accumulator chain is dependent?
// One wave per SIMD (256-thread workgroup, one per CU), operands in registers, NCH independent accumulators used
// round-robin: NCH = 1 is the 24-deep chain of the encoder's dgrad tile.  Cycles by s_memtime around 4096 MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/exp/mfma_chain_probe.hip -o tools/exp/build/libmfma_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe16(unsigned long long* out, float* sink, int iters) {
  u32x4 a4 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u};
  u32x4 b4 = {0x3f803f80u, 0x3f803f00u + threadIdx.x, 0x3f803f80u, 0x3f003f80u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
  f32x4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k % NCH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k % NCH], 0, 0, 0);
  }
  asm volatile("s_nop 0" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe32(unsigned long long* out, float* sink, int iters) {
  u32x4 a4 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u};
  u32x4 b4 = {0x3f803f80u, 0x3f803f00u + threadIdx.x, 0x3f803f80u, 0x3f003f80u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
  f32x16 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 24; ++k) acc[k % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k % NCH], 0, 0, 0);
  }
  asm volatile("s_nop 0" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][15];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

extern "C" int probe(int kind, int nch, int waves, unsigned long long* out, float* sink, int iters, void* stream) {
#define L(K, N, W) hipLaunchKernelGGL((K<N, W>), dim3(256), dim3(64 * W), 0, (hipStream_t)stream, out, sink, iters)
  if (kind == 16) {
    if (waves == 4) { if (nch == 1) L(probe16, 1, 4); else if (nch == 2) L(probe16, 2, 4); else if (nch == 3) L(probe16, 3, 4); else if (nch == 4) L(probe16, 4, 4); else L(probe16, 8, 4); }
    else { if (nch == 1) L(probe16, 1, 8); else if (nch == 2) L(probe16, 2, 8); else if (nch == 4) L(probe16, 4, 8); else L(probe16, 8, 8); }
  } else {
    if (waves == 4) { if (nch == 1) L(probe32, 1, 4); else if (nch == 2) L(probe32, 2, 4); else L(probe32, 4, 4); }
    else { if (nch == 1) L(probe32, 1, 8); else L(probe32, 2, 8); }
  }
  return (int)hipGetLastError();
}
