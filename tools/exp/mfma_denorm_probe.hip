// Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs?  A = bytes zero-extended to 16 bits (= b * 2^-24 as fp16
// subnormals), B = ordinary fp16 values.  If it does, a uint8 pixel becomes an MFMA operand with a byte permute alone
// (no v_cvt_f32_ubyte / v_cvt_pkrtz): D = 2^-24 * sum_k b[k] * w[k], exact in fp32.
#include <hip/hip_runtime.h>
#include <cstdint>
typedef _Float16 fh8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// A [16][32] uint16 bit patterns (row-major), B [32][16] uint16 bit patterns (k-major), D [16][16] fp32
__global__ void probe(const uint16_t* A, const uint16_t* B, float* D) {
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  uint16_t a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = A[i * 32 + 8 * q + j]; b[j] = B[(8 * q + j) * 16 + i]; }
  u32x4 av, bv;
  for (int e = 0; e < 4; ++e) { av[e] = a[2 * e] | ((unsigned)a[2 * e + 1] << 16); bv[e] = b[2 * e] | ((unsigned)b[2 * e + 1] << 16); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fh8, av), __builtin_bit_cast(fh8, bv), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = acc[r];
}

extern "C" int denorm_probe(const uint16_t* A, const uint16_t* B, float* D, void* stream) {
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
  return (int)hipGetLastError();
}
