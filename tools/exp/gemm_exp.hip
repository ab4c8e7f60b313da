// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// C[M,N] (+)= op(A)[M,K] * op(B)[K,N]  (+ bias[N]) (ReLU) (ReLU-mask) ; row-major everywhere.
// Replaces the tf.matmul call sites of /root/reference/model/model.py:314,334,423 and the matmul
// halves of BasicLSTMCell (model.py:110,346-351), plus their tf.gradients counterparts
// (/root/reference/train/rmsprop_applier.py:100-105): NN = forward, NT = dgrad, TN = wgrad.
//
// Design (CDNA4): 256 threads = 4 waves in a 2x2 grid, each wave owns (BM/2)x(BN/2) of the block
// tile as 32x32 MFMA accumulators.  Operand tiles are staged through LDS in their NATIVE global
// orientation (no transposing stores):
//   RK ("rows, k contiguous"):  S[row][k], row stride BK+4 floats -> one ds_read_b128 yields the 4
//       k-values a lane feeds to 4 consecutive MFMAs (the k order inside a group of 8 is permuted
//       identically for A and B: lane half kq at step s supplies k = 8g + 4kq + s);
//   KM ("k major"):            S[k][col], row stride BMN+4 floats -> conflict-free ds_read_b32.
// 16 B/lane global loads are register-prefetched one K-tile ahead; one barrier per K-tile.
#include "../../unreal_amd/csrc/common.h"
#ifndef EXP
#define EXP 0
#endif

namespace {

constexpr int BK = 32;
constexpr int FLAG_RELU = 1, FLAG_ACCUM = 2, FLAG_ATOMIC = 4, FLAG_RELU_MASK = 8;

struct GemmArgs {
  int M, N, K;
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  const float* bias;
  const float* mask; int ldm;
  int flags;
  int ktiles_per_split;
  int vecA, vecB;
};

// Load a [ROWS x BK] (RK) or [BK x ROWS] (KM) operand tile slice owned by this thread into regs.
template <int ROWS, bool KM>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int rows_total, int K, int r0, int k0,
                                          int vec, f32x4 (&reg)[ROWS * BK / 1024]) {
  constexpr int NV = ROWS * BK / 1024;   // float4 per thread
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    int id = threadIdx.x + 256 * p;
    int r, k;
    size_t off;
    bool full, any;
    if (!KM) {
      r = id / (BK / 4);
      k = (id % (BK / 4)) * 4;
      off = (size_t)(r0 + r) * ld + (k0 + k);
      any = (r0 + r) < rows_total && (k0 + k) < K;
      full = (r0 + r) < rows_total && (k0 + k + 3) < K;
    } else {
      k = id / (ROWS / 4);
      r = (id % (ROWS / 4)) * 4;
#if EXP == 3
      off = (size_t)(r0 + id / (BK / 4)) * ld + (k0 + (id % (BK / 4)) * 4);
#else
      off = (size_t)(k0 + k) * ld + (r0 + r);
#endif
      any = (k0 + k) < K && (r0 + r) < rows_total;
      full = (k0 + k) < K && (r0 + r + 3) < rows_total;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (full && vec) {
      v = *reinterpret_cast<const f32x4*>(P + off);
    } else if (any) {
      int lim = KM ? (rows_total - (r0 + r)) : (K - (k0 + k));
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < lim) v[e] = P[off + e];
    }
    reg[p] = v;
  }
}

template <int ROWS, bool KM>
__device__ __forceinline__ void store_tile(float* S, const f32x4 (&reg)[ROWS * BK / 1024]) {
  constexpr int NV = ROWS * BK / 1024;
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    int id = threadIdx.x + 256 * p;
    if (!KM) {
      int r = id / (BK / 4), k = (id % (BK / 4)) * 4;
      *reinterpret_cast<f32x4*>(S + r * (BK + 4) + k) = reg[p];
    } else {
#if EXP == 4
      int r = id / (BK / 4), k = (id % (BK / 4)) * 4;
      *reinterpret_cast<f32x4*>(S + r * (BK + 4) + k) = reg[p];
#else
      int k = id / (ROWS / 4), r = (id % (ROWS / 4)) * 4;
      *reinterpret_cast<f32x4*>(S + k * (ROWS + 4) + r) = reg[p];
#endif
    }
  }
}

// fragment for MFMA k-group g: the 4 k-values this lane supplies (k = 8g + 4kq + s, s = 0..3)
template <int ROWS, bool KM>
__device__ __forceinline__ f32x4 read_frag(const float* S, int row, int g, int kq) {
  if (!KM) {
    return *reinterpret_cast<const f32x4*>(S + row * (BK + 4) + 8 * g + 4 * kq);
  } else {
#if EXP == 2
    return *reinterpret_cast<const f32x4*>(S + row * (BK + 4) + 8 * g + 4 * kq);
#else
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = S[(8 * g + 4 * kq + s) * (ROWS + 4) + row];
    return v;
#endif
  }
}

template <int BM, int BN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_ELEMS = A_KM ? BK * (BM + 4) : BM * (BK + 4);
  constexpr int B_ELEMS = B_KM ? BK * (BN + 4) : BN * (BK + 4);
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * A_ELEMS;

  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int nk_total = (p.K + BK - 1) / BK;
  const int kt0 = blockIdx.z * p.ktiles_per_split;
  const int kt1 = min(nk_total, kt0 + p.ktiles_per_split);
  if (kt0 >= kt1) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kq = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[BM * BK / 1024], rb[BN * BK / 1024];
  load_tile<BM, A_KM>(p.A, p.lda, p.M, p.K, m0, kt0 * BK, p.vecA, ra);
  load_tile<BN, B_KM>(p.B, p.ldb, p.N, p.K, n0, kt0 * BK, p.vecB, rb);
  store_tile<BM, A_KM>(As0, ra);
  store_tile<BN, B_KM>(Bs0, rb);
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    const bool more = (kt + 1) < kt1;
    if (more) {
      load_tile<BM, A_KM>(p.A, p.lda, p.M, p.K, m0, (kt + 1) * BK, p.vecA, ra);
      load_tile<BN, B_KM>(p.B, p.ldb, p.N, p.K, n0, (kt + 1) * BK, p.vecB, rb);
    }
    const float* Ac = As0 + cur * A_ELEMS;
    const float* Bc = Bs0 + cur * B_ELEMS;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, A_KM>(Ac, wm * (BM / 2) + i * 32 + li, g, kq);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = read_frag<BN, B_KM>(Bc, wn * (BN / 2) + j * 32 + li, g, kq);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MFMA32(af[i][s], bf[j][s], acc[i][j]);
    }
    if (more) {
      store_tile<BM, A_KM>(As0 + (cur ^ 1) * A_ELEMS, ra);
      store_tile<BN, B_KM>(Bs0 + (cur ^ 1) * B_ELEMS, rb);
    }
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool first_split = blockIdx.z == 0;
  if (p.flags & FLAG_ATOMIC) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + li;
        if (col >= p.N) continue;
        const float bv = (p.bias && first_split) ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
          if (row < p.M) atomicAdd(p.C + (size_t)row * p.ldc + col, acc[i][j][r] + bv);
        }
      }
    return;
  }
  // stage the tile through LDS so every lane moves 16 B of one row: full 128 B lines for C, mask and accumulate
  constexpr int CLD = BN + 4;
  static_assert(BM * CLD <= 2 * (A_ELEMS + B_ELEMS), "C tile must fit in the operand LDS");
  float* Cs = smem;     // operand buffers are dead: the K loop ended with a barrier
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq) * CLD + wn * (BN / 2) + j * 32 + li] = acc[i][j][r];
  __syncthreads();
  const bool vecC = ((p.ldc & 3) == 0) && ((((uintptr_t)p.C) & 15) == 0) &&
                    (!p.bias || ((((uintptr_t)p.bias) & 15) == 0)) &&
                    (!(p.flags & FLAG_RELU_MASK) || (((p.ldm & 3) == 0) && ((((uintptr_t)p.mask) & 15) == 0)));
  constexpr int CV = BN / 4;                      // f32x4 per tile row
  for (int id = threadIdx.x; id < BM * CV; id += 256) {
    const int r = id / CV, c4 = (id % CV) * 4;
    const int row = m0 + r, col = n0 + c4;
    if (row >= p.M || col >= p.N) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * CLD + c4);
    float* cp = p.C + (size_t)row * p.ldc + col;
    if (vecC && col + 3 < p.N) {
      if (p.bias) { const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + col); v += b4; }
      if (p.flags & FLAG_ACCUM) v += *reinterpret_cast<const f32x4*>(cp);
      if (p.flags & FLAG_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.flags & FLAG_RELU_MASK) {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(p.mask + (size_t)row * p.ldm + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = m4[e] > 0.f ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(cp) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (col + e >= p.N) break;
        float x = v[e] + (p.bias ? p.bias[col + e] : 0.f);
        if (p.flags & FLAG_ACCUM) x += cp[e];
        if (p.flags & FLAG_RELU) x = fmaxf(x, 0.f);
        if (p.flags & FLAG_RELU_MASK) x = (p.mask[(size_t)row * p.ldm + col + e] > 0.f) ? x : 0.f;
        cp[e] = x;
      }
    }
  }
}

template <int BM, int BN>
int launch(int ta, int tb, const GemmArgs& a, int splitk, hipStream_t st) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splitk), block(256);
  if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, false, true>), grid, block, 0, st, a);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false>), grid, block, 0, st, a);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false>), grid, block, 0, st, a);
  return unreal_launch_status();
}

}  // namespace

extern "C" int exp_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                               int ldb, float* C, int ldc, const float* bias, const float* mask, int ldm, int flags,
                               int splitk, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return UNREAL_EINVAL;
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return UNREAL_EINVAL;
  if ((flags & FLAG_RELU_MASK) && (!mask || ldm < N)) return UNREAL_EINVAL;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && !(flags & FLAG_ATOMIC)) return UNREAL_EINVAL;
  if ((flags & FLAG_ATOMIC) && (flags & (FLAG_RELU | FLAG_RELU_MASK | FLAG_ACCUM))) return UNREAL_EINVAL;
  GemmArgs a;
  a.M = M; a.N = N; a.K = K;
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.bias = bias; a.mask = mask; a.ldm = ldm; a.flags = flags;
  const int nk = (K + BK - 1) / BK;
  if (splitk > nk) splitk = nk;
  a.ktiles_per_split = (nk + splitk - 1) / splitk;
  splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
  a.vecA = ((lda & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
  a.vecB = ((ldb & 3) == 0) && ((((uintptr_t)B) & 15) == 0);
  const long blocks128 = (long)((M + 127) / 128) * ((N + 127) / 128) * splitk;
  if (blocks128 >= 384) return launch<128, 128>(transA, transB, a, splitk, (hipStream_t)stream);
  return launch<64, 64>(transA, transB, a, splitk, (hipStream_t)stream);
}
