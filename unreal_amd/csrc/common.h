// Shared definitions for the gfx950 (MI355X) kernels of the UNREAL hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UNREAL_OK 0
#define UNREAL_EINVAL (-22)
#define UNREAL_ELAUNCH (-5)

#define FRAME_H 84
#define FRAME_W 84
#define FRAME_C 3
#define FRAME_ROW_BYTES (FRAME_W * FRAME_C)                 // 252
#define FRAME_BYTES (FRAME_H * FRAME_W * FRAME_C)           // 21168 = 16 * 1323
#define PC_CELLS 400                                        // 20 x 20 pixel-change map
#define C1_POS 400                                          // conv1 output positions (20 x 20)
#define C1_CH 16
#define C2_POS 81                                           // conv2 output positions (9 x 9)
#define C2_CH 32
#define F2_DIM 2592                                         // 9 * 9 * 32
#define LSTM_N 256

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

static inline int unreal_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? UNREAL_OK : UNREAL_ELAUNCH;
}

// ---- fp16 hi + lo operands (gemm_split.hip): per-tensor power-of-two scales -------------------------------------
// An "absmax slot" is one float in device memory holding max |x| over a tensor.  Producers commit their outputs' maximum
// with ONE atomic per workgroup (its waves' maxima meet in LDS first; non-negative floats order like unsigned integers); consumers turn it into the power of two
// that puts the largest element into [2^14, 2^15) -- inside fp16's range with a factor two to spare, and deep enough that
// hi + lo carry 22 significant bits for every element down to 2^-17 of the maximum.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Tens of thousands of waves committing to ONE address would serialise at ~12 ns per atomic (0.6 ms for the 13 k
// workgroups of a big GEMM), so a wave first reads the slot past its L1 (agent-scope relaxed load: the L2 that also
// executes the atomic) and only issues the atomic when it would raise the value -- the number of atomics is then the
// number of running-maximum records (~log of the wave count).  A stale read can only cause a redundant atomic.
__device__ __forceinline__ void absmax_commit(float* slot, float m) {      // m >= 0; call with the whole wave active
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && slot) {
    unsigned int* s = reinterpret_cast<unsigned int*>(slot);
    const unsigned int cur = __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__float_as_uint(m) > cur) atomicMax(s, __float_as_uint(m));
  }
}
// scale = 2^(14 - floor(log2 m)) (1 for m == 0 / subnormal / inf / nan); inverse exact
__device__ __forceinline__ float pow2_scale(float m) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);           // biased exponent of m
  const int se = (e == 0 || e == 255) ? 127 : min(max(268 - e, 27), 227);
  return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ float pow2_inv(float scale) { return __uint_as_float((254u << 23) - __float_as_uint(scale)); }

// ReLU masks are read from the hi PLANE of a non-negative activation (conv1 output in unreal_encoder_bwd, the
// pixel-control fc output in unreal_pc_deconv_bwd: "hi != 0 <=> x > 0").  A positive x below 2^-39 of the tensor maximum
// rounds to fp16 zero under the tensor's scale and would be masked out -- an error of the size of the whole gradient
// element, not of x.  So the hi halfword of a positive x is kept at least at the smallest fp16 subnormal (2^-24 in scaled
// units: under 2^-38 of the tensor maximum, inside the format's own absolute floor).  hi_pair = two fp16 of x0, x1 >= 0.
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int keep_positive_visible(unsigned int hi_pair, float x0, float x1) {
  const u16x2_t f = {(unsigned short)(x0 > 0.f ? 1 : 0), (unsigned short)(x1 > 0.f ? 1 : 0)};
  return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, hi_pair), f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
