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
