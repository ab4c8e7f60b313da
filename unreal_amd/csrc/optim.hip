// Global-norm clip + shared RMSProp over one flat fp32 parameter buffer (K12 + K13), gfx950.
//
// Reference: /root/reference/train/rmsprop_applier.py:38-43 (slots: rms = 1, momentum = 0),
// :83-93 (training_ops.apply_rms_prop), :121 (tf.clip_by_global_norm), known-answer test
// train/rmsprop_applier_test.py:29-51:
//     ms  += (g*g - ms) * (1 - decay)
//     mom  = momentum * mom + lr * g / sqrt(ms + eps)        (eps INSIDE the sqrt)
//     var -= mom
// with g = grad * clip_norm * min(1/||grad||, 1/clip_norm).
// HBM-bound: 28 B per parameter per step, 16 B/lane loads and stores.  The norm is reduced in a
// fixed order (same result on every rank, so replicated parameters stay bit-identical).
#include "common.h"

namespace {

constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void norm_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ norm_out) {
  __shared__ float red[4];
  float s = threadIdx.x < nb ? partial[threadIdx.x] : 0.f;
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ var, float* __restrict__ ms, float* __restrict__ mom,
                                                      const float* __restrict__ grad, long n, float lr, float decay,
                                                      float momentum, float eps, float clip_norm,
                                                      const float* __restrict__ norm) {
  float scale = 1.f;
  if (clip_norm > 0.f) {
    const float nv = norm[0];
    scale = nv > 0.f ? clip_norm * fminf(1.f / nv, 1.f / clip_norm) : 1.f;
  }
  const float omd = 1.f - decay;
  const long n4 = n >> 2;
  f32x4* v4 = reinterpret_cast<f32x4*>(var);
  f32x4* s4 = reinterpret_cast<f32x4*>(ms);
  f32x4* m4 = reinterpret_cast<f32x4*>(mom);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(grad);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 g = g4[i], s = s4[i], m = m4[i], v = v4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = g[e] * scale;
      s[e] = s[e] + (ge * ge - s[e]) * omd;
      m[e] = m[e] * momentum + (ge * lr) / sqrtf(s[e] + eps);
      v[e] = v[e] - m[e];
    }
    s4[i] = s; m4[i] = m; v4[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long i = (n4 << 2) + threadIdx.x;
    float ge = grad[i] * scale;
    float s = ms[i] + (ge * ge - ms[i]) * omd;
    float m = mom[i] * momentum + (ge * lr) / sqrtf(s + eps);
    ms[i] = s; mom[i] = m; var[i] = var[i] - m;
  }
}

// dst[i] = src[i] for 4-byte words: the state hand-overs of the hot path (LSTM state, index lists) are launched as an
// ordinary kernel of this library rather than as hipMemcpyDtoDAsync: every dispatch of process() is then a kernel of
// this library that shows up by name in a kernel trace (the runtime's blit kernel does not).  NOTE: round 2 blamed a
// rocprofv3 --pmc crash on that blit kernel; the crash came back with no D2D copy left and went away once the replay
// fill stopped queueing ~24 k dispatches without a host sync -- same trigger both times as far as the evidence goes
// (profiles/r02_pmc_fault.log).  Trainer._fill_experience now bounds that queue itself.
__global__ void copy_words_kernel(long n, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int vec) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    const long n4 = n >> 2;
    for (long k = i; k < n4; k += stride) d4[k] = s4[k];
    for (long k = (n4 << 2) + i; k < n; k += stride) dst[k] = src[k];
  } else {
    for (long k = i; k < n; k += stride) dst[k] = src[k];
  }
}

__global__ void axpy_kernel(long n, float alpha, const float* __restrict__ x, float* __restrict__ y) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += alpha * x[i];
}

}  // namespace

extern "C" {

int unreal_axpy_f32(long n, float alpha, const float* x, float* y, void* stream) {
  if (n <= 0 || !x || !y) return UNREAL_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, alpha, x, y);
  return unreal_launch_status();
}

int unreal_copy_words(long n, const void* src, void* dst, void* stream) {
  if (n <= 0 || !src || !dst || ((((uintptr_t)src) | ((uintptr_t)dst)) & 3)) return UNREAL_EINVAL;
  const int vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  long work = vec ? (n + 3) / 4 : n;
  int blocks = (int)((work + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(copy_words_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, (const uint32_t*)src,
                     (uint32_t*)dst, vec);
  return unreal_launch_status();
}

// norm_out[0] = ||grad||_2 ; scratch must hold 256 floats
int unreal_grad_norm(const float* grad, long n, float* scratch, float* norm_out, void* stream) {
  if (!grad || n <= 0 || !scratch || !norm_out || (((uintptr_t)grad) & 15)) return UNREAL_EINVAL;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, (hipStream_t)stream, grad, n, scratch);
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, NORM_BLOCKS, norm_out);
  return unreal_launch_status();
}

// clip_norm <= 0 disables clipping (norm may then be null)
int unreal_rmsprop_step(float* var, float* ms, float* mom, const float* grad, long n, float lr, float decay,
                        float momentum, float eps, float clip_norm, const float* norm, void* stream) {
  if (!var || !ms || !mom || !grad || n <= 0 || (clip_norm > 0.f && !norm)) return UNREAL_EINVAL;
  if ((((uintptr_t)var) | ((uintptr_t)ms) | ((uintptr_t)mom) | ((uintptr_t)grad)) & 15) return UNREAL_EINVAL;
  long n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, var, ms, mom, grad, n, lr, decay,
                     momentum, eps, clip_norm, norm);
  return unreal_launch_status();
}

}  // extern "C"
