// fp32-equivalent GEMMs on the 16-bit matrix cores of gfx950 ("split-operand").  Two operand formats:
//   round 3 (all kernels): fp16 hi + lo with ONE power-of-two scale per tensor, 3 term pairs (see "fp16x2" below)
//   round 2 (SPLIT_NT_MODE / SPLIT_TN_MODE 0, kept for A/B): 3 x bf16, 6 term pairs, described first:
//
//   C[M,N] = A[M,K] * W[N,K]^T (+ bias) (ReLU) (ReLU-mask) (accumulate)
//
// A is an fp32 activation matrix (k contiguous).  W is a WEIGHT matrix that the host keeps as a pre-split shadow:
// three bf16 planes W = W1 + W2 + W3 (round-to-nearest at each level, residuals exact in fp32: 8+8+8 mantissa
// bits), rows zero-padded to a multiple of 32 k (unreal_split_bf16x3, refreshed after every optimiser step).  A's
// tile is split the same way when it is staged into LDS.  Each 32x32x16 product tile is accumulated in fp32 from
// the six term pairs whose weight is >= 2^-16:
//   a1b1, a1b2, a2b1, a1b3, a2b2, a3b1          (v_mfma_f32_32x32x16_bf16, exact bf16 x bf16 products)
// The dropped pairs (a2b3, a3b2, a3b3) are below 2^-25 |ab| with random sign, i.e. under the rounding error of the
// fp32 product they replace, so the result carries fp32-grade error (parity tests use the SAME tolerance as the
// fp32-MFMA kernel of gemm.hip, and compare the two kernels' errors on the same data).
// Used for the forward and dgrad GEMMs of /root/reference/model/model.py:314,334,423 + BasicLSTMCell (model.py:110);
// wgrad (both operands activations, k-major) stays on the fp32 MFMA path of gemm.hip.
//
// Layout: 128x128x32 (or 64x64x32) block tile, 4 waves as 2x2.  LDS image per operand: [plane(3)][row][k(32) bf16]
// with 80-byte rows (ds_read_b128 fragment reads are conflict-free).  One LDS buffer; the next K tile travels in
// registers and is split/stored between the two barriers while the second half of the current tile's MFMAs runs.
// Blocks are numbered so that the column blocks sharing one A row panel run back-to-back on ONE XCD (its L2 then
// serves the re-reads; with the default order they land on 8 different L2s and A is fetched from HBM 2..8 times).
//
// fp16x2 (NT kernels).  x * s = hi + lo, both fp16 (round to nearest; hi + lo carries 22 significant bits), s the power of
// two that puts the tensor's largest |x| into [2^14, 2^15).  The tensor maxima travel in "absmax slots" (common.h): the
// weight's is reduced when its shadow is made, an activation's is committed by the kernel that PRODUCES it (encoder,
// GEMM / LSTM / deconv epilogues: one atomic per workgroup) -- unreal_absmax_f32 is the stand-alone reduction for callers that
// have none.  A product tile accumulates hh + hl + lh on v_mfma_f32_32x32x16_f16 (the dropped ll pair is < 2^-22 |ab|) and
// is un-scaled exactly in the epilogue.  Gate (tools/exp/f16x2_gate.py, profiles/r03_f16x2_gate.log):
// on the trainer's live operands at production shape the error against fp64 is BELOW the plain-fp32-MFMA kernel's
// (rms and max) for fc forward, fc dgrad and the LSTM dgrad, at 1.3 - 1.7x the speed of the 6-pass kernel.  The wgrad
// form first FAILED that gate at K = 81,920 (rms 4.6e-7 vs 2.3e-7; the 6-pass form it replaces: 1.5e-6): the MFMA's own
// accumulation is biased over a 3,400-deep chain.  With the accumulators flushed into a second fp32 set every 8 K tiles
// (SPLIT_TN_FLUSH) it passes (1.0e-7 / 2.9e-7 against 2.3e-7 / 6.6e-7) at 1.28x the speed of the shipped 6-pass kernel.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int MODE_BF16X3 = 0, MODE_F16X2 = 1;
#ifndef SPLIT_NT_MODE
#define SPLIT_NT_MODE 1     // 0: the round-2 bf16x3 NT kernels (A/B: tools/exp/f16x2_gate.py)
#endif
#ifndef SPLIT_TN_MODE
#define SPLIT_TN_MODE 1     // 0: the round-2 bf16x3 wgrad kernel
#endif
__host__ __device__ constexpr int npl(int mode) { return mode == MODE_F16X2 ? 2 : 3; }   // operand planes
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int MODE> struct Frag { typedef __bf16 type __attribute__((ext_vector_type(8))); };
template <> struct Frag<MODE_F16X2> { typedef f16x8 type; };
constexpr int ROW_B = 80;                    // bytes per LDS row: 32 bf16 + 16 pad
constexpr int FLAG_RELU = 1, FLAG_ACCUM = 2, FLAG_ATOMIC = 4, FLAG_RELU_MASK = 8, FLAG_RELU_BITS = 16;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef SPLIT_ABLATE
#define SPLIT_ABLATE 0      // tools/exp/split_ablate.py: 1 no global loads in the loop, 2 no split/store, 4 no MFMA,
                            // 8 every block reads the same 8 rows of A (cache hits)
#endif

#ifndef SPLIT_NT_DBUF
#define SPLIT_NT_DBUF 0     // 1: two LDS operand images, one barrier per K tile (tools/exp/gemm_ab.py)
#endif
#ifndef SPLIT_NT_OCC3
#define SPLIT_NT_OCC3 1     // three 128 x 128 workgroups per CU (168 VGPRs, epilogue staged in two 64-row halves so that a
                            // workgroup needs 40,960 B of LDS instead of 67,648).  Round 4, tools/exp/gemm_ab.py: the K = 256
                            // products (13,440 tiles of 8 K steps: prologue / epilogue-bound) 0.89-0.90x the time, the long-K
                            // products unchanged, outputs bit-identical (profiles/r04_gemm_occ3_ab.log).  0: two per CU.
#endif
#ifndef SPLIT_ROW_DEAL
#define SPLIT_ROW_DEAL 1    // 0: tile rows in lane order (tools/exp/ab_split_rows.py times both)
#endif
// Which tile row an 8-lane (A, ds_write_b64) or 4-lane (W, ds_write_b128) block stages.  A store instruction's lane
// group covers two such blocks; with 80-byte LDS rows, rows r and r + 1 overlap in 4 of the 32 store banks (every store
// 2-way conflicted: 13 % of the kernel's cycles were LDS conflict cycles, all from stores), rows r and r + 4 do not
// (80 * 4 = 320 B = 16 banks past a multiple of 128 B).  So blocks 2m and 2m + 1 take rows (m, m + 4) of each 8 rows.
__device__ __forceinline__ int row_deal(int j) {
  return SPLIT_ROW_DEAL ? ((j & ~7) | ((j & 1) << 2) | ((j >> 1) & 3)) : j;
}

struct SplitArgs {
  int M, N, K;
  const float* A; int lda;
  const unsigned short* W; int ldw; long plane;     // bf16 planes: W + t*plane + n*ldw + k
  float* C; int ldc;
  const float* bias;
  const float* mask; int ldm;
  int flags;
  int vecA;
  int nbx, nby;
  int splitk, ktiles_per_split;        // > 1: K slabs, added into C with fp32 atomics (FLAG_ATOMIC) or ...
  long slab_stride;                    // ... != 0: stored as partial products, slab s at C + s * slab_stride (unreal_gemm_f32_split_nt_slabs)
  // fused LSTM step (EPI == 1): C is the gate buffer [M][1024] (in: input-half pre-activations, out: activated gates)
  const float* c_prev; float* c_out; float* h_out; int ld_h;
  // DUAL: A operand = [A (K1 valid columns, k < K1pad) | A2 (K - K1pad columns)]
  const float* A2; int lda2, K1, K1pad;
  // fused BPTT step (EPI == 2): the product is dh_rec of the EARLIER time step, whose gate backward runs in the epilogue
  const float* dh_above; float* dc_io; const float* gates_act; const float* c_new; float* dpre;     // c_prev: above
  // fp16x2: absmax slots of A (max of the slot and a_floor) and of W (the one its shadow was made with); optional
  // slots that receive max |output| (plain epilogue: C after bias / ReLU / mask; EPI == 2: dpre)
  const float* a_absmax; float a_floor; const float* w_absmax; float* c_absmax0; float* c_absmax1;
};

// One 16-byte piece of the A tile per call (piece p of ROWS*BK/1024).  Branch-free on purpose: rows past the end
// re-read the last valid row (their C rows are never stored); a vector that would start past the row is pulled back
// inside it and k >= K is zeroed when the piece is split (a_piece_store), so any lda >= K works.  Divergent or even
// uniform branches here break the K loop into several basic blocks and the MFMA / VALU / load interleave is lost
// (with divergent ones the compiler also serialises the loads with s_waitcnt vmcnt(0)).
template <int ROWS, bool VEC>
__device__ __forceinline__ f32x4 a_piece_load(const float* __restrict__ P, int ld, int rows_total, int K, int r0, int k0, int p,
                                              int tid) {
  const int id = tid + 256 * p;
  const int r = (SPLIT_ABLATE & 8) ? (id / (BK / 4)) & 7 : min(r0 + row_deal(id / (BK / 4)), rows_total - 1);
  const int k = k0 + (id % (BK / 4)) * 4;
  const float* row = P + (size_t)r * ld;
  if (VEC) return *reinterpret_cast<const f32x4*>(row + min(k, ld - 4));
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = row[min(k + e, K - 1)];
  return v;
}

// 4 fp32 values -> three bf16x4 terms (8 bytes each): round to nearest, residual exact in fp32, twice.
// PACKED selects how the residuals are subtracted: as 2-vectors (v_pk_add_f32) or with scalar v_sub_f32.  Same values
// either way; A/B on one device: the NT kernels (one operand split) are 1-5 % faster with the scalar form, the TN
// kernel (both operands split, twice the VALU) 5 % faster with the packed one.
template <bool PACKED, int MODE = MODE_BF16X3>
__device__ __forceinline__ void split4_terms(float x0, float x1, float x2, float x3, u32x2 (&pl)[3]) {
  if (MODE == MODE_F16X2) {   // x (already scaled) = hi + lo, both fp16, round to nearest; pl[2] unused
    const f16x2 h01 = __builtin_convertvector((f32x2){x0, x1}, f16x2), h23 = __builtin_convertvector((f32x2){x2, x3}, f16x2);
    const f32x2 r01 = (f32x2){x0, x1} - __builtin_convertvector(h01, f32x2), r23 = (f32x2){x2, x3} - __builtin_convertvector(h23, f32x2);
    const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
    pl[0] = (u32x2){__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
    pl[1] = (u32x2){__builtin_bit_cast(unsigned int, l01), __builtin_bit_cast(unsigned int, l23)};
    pl[2] = pl[1];
    return;
  }
  if (PACKED) {
    f32x2 x01 = {x0, x1}, x23 = {x2, x3};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bf16x2 h01 = __builtin_convertvector(x01, bf16x2), h23 = __builtin_convertvector(x23, bf16x2);
      pl[t] = (u32x2){__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
      if (t < 2) {
        x01 = x01 - __builtin_convertvector(h01, f32x2);
        x23 = x23 - __builtin_convertvector(h23, f32x2);
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bf16x2 h01 = __builtin_convertvector((f32x2){x0, x1}, bf16x2), h23 = __builtin_convertvector((f32x2){x2, x3}, bf16x2);
      const unsigned int w01 = __builtin_bit_cast(unsigned int, h01), w23 = __builtin_bit_cast(unsigned int, h23);
      pl[t] = (u32x2){w01, w23};
      if (t < 2) {
        x0 -= __uint_as_float(w01 << 16); x1 -= __uint_as_float(w01 & 0xffff0000u);
        x2 -= __uint_as_float(w23 << 16); x3 -= __uint_as_float(w23 & 0xffff0000u);
      }
    }
  }
}

// split 4 consecutive-k fp32 values into three bf16 planes and store 8 bytes per plane; klim = K - k0 of that tile
template <int ROWS, int MODE, bool RAGGED = true>
__device__ __forceinline__ void a_piece_store(unsigned char* S, f32x4 v, int klim, int p, int tid, float scale) {
  const int id = tid + 256 * p;
  const int r = row_deal(id / (BK / 4)), k = (id % (BK / 4)) * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (!RAGGED || k + e < klim) ? (MODE == MODE_F16X2 ? v[e] * scale : v[e]) : 0.f;
  u32x2 pl[3];
  split4_terms<false, MODE>(v[0], v[1], v[2], v[3], pl);
#pragma unroll
  for (int t = 0; t < npl(MODE); ++t)
    *reinterpret_cast<u32x2*>(S + (t * ROWS + r) * ROW_B + k * 2) = pl[t];
}

// weight tile piece: 3 planes x ROWS rows x 64 B, 16 B per lane (planes are zero-padded in k, rows clamped)
template <int ROWS>
__device__ __forceinline__ u32x4 w_piece_load(const unsigned short* __restrict__ W, int ldw, long plane, int rows_total,
                                              int r0, int k0, int p, int tid) {
  const int id = tid + 256 * p;
  const int t = id / (ROWS * 4), r = row_deal((id / 4) % ROWS), c = id % 4;
  const int row = min(r0 + r, rows_total - 1);
  return *reinterpret_cast<const u32x4*>(W + t * plane + (size_t)row * ldw + k0 + c * 8);
}

template <int ROWS>
__device__ __forceinline__ void w_piece_store(unsigned char* S, u32x4 v, int p, int tid) {
  const int id = tid + 256 * p;
  const int t = id / (ROWS * 4), r = row_deal((id / 4) % ROWS), c = id % 4;
  *reinterpret_cast<u32x4*>(S + (t * ROWS + r) * ROW_B + c * 16) = v;
}

// SWZ: the 16-byte k chunk c of row r sits at chunk c ^ ((r >> 4) & 3) (transposing stores of the TN kernel)
template <int BM, int BN, int MODE, bool SWZ = false>
__device__ __forceinline__ void read_frags(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int li, int kh,
                                           int ks, typename Frag<MODE>::type (&af)[BM / 64][npl(MODE)],
                                           typename Frag<MODE>::type (&bf)[BN / 64][npl(MODE)]) {
  typedef typename Frag<MODE>::type frag8;
  const int c = 2 * ks + kh;
#pragma unroll
  for (int i = 0; i < BM / 64; ++i)
#pragma unroll
    for (int t = 0; t < npl(MODE); ++t) {
      const int r = wm * (BM / 2) + i * 32 + li;
      af[i][t] = *reinterpret_cast<const frag8*>(As + (t * BM + r) * ROW_B + (SWZ ? (c ^ ((r >> 4) & 3)) : c) * 16);
    }
#pragma unroll
  for (int j = 0; j < BN / 64; ++j)
#pragma unroll
    for (int t = 0; t < npl(MODE); ++t) {
      const int r = wn * (BN / 2) + j * 32 + li;
      bf[j][t] = *reinterpret_cast<const frag8*>(Bs + (t * BN + r) * ROW_B + (SWZ ? (c ^ ((r >> 4) & 3)) : c) * 16);
    }
}

template <int TM, int TN>
__device__ __forceinline__ void mma_frags(const bf16x8 (&af)[TM][3], const bf16x8 (&bf)[TN][3], f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x16 c = acc[i][j];
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);   // smallest terms first
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
      acc[i][j] = c;
    }
}

template <int TM, int TN>
__device__ __forceinline__ void mma_frags(const f16x8 (&af)[TM][2], const f16x8 (&bf)[TN][2], f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x16 c = acc[i][j];
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bf[j][0], c, 0, 0, 0);    // lo * hi, hi * lo, hi * hi
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][0], c, 0, 0, 0);
      acc[i][j] = c;
    }
}

// DEEP: activation tiles travel two K tiles ahead instead of one.  Measured on one device, same process: +20 % on the
// 64x64 kernel's long-K shapes (4096 x 256 x 2592: 76 -> 62 us), -3 % on the 128x128 kernel (16 more live registers).
//
// KW (64x64 tiles of the 4096-row rollout / BPTT steps): KW groups of 4 waves share ONE output tile and deal its K tiles
// round-robin (group g takes tiles g, g + KW, ...), each with its own LDS image; their partial tiles are summed in a
// fixed order in the epilogue.  A 4096 x 256 product has only 256 tiles of 64 x 64 -- one 4-wave workgroup per CU, one
// wave per SIMD, nothing to hide an LDS read or a barrier behind; with KW = 4 the same tile keeps 4 waves per SIMD busy
// without fetching any operand byte twice (smaller tiles would) and without a zero-fill + atomics pass (split-K over
// workgroups would).
//
// DUAL (the LSTM step of a rollout): the A operand is the row-wise concatenation [A | A2] -- k < K1pad comes from A
// (K1 valid columns, zero beyond), the rest from A2 -- multiplied by ONE weight shadow laid out the same way, so the
// input half and the recurrent half of the gates are one product (BasicLSTMCell's own [x, h] @ kernel, model.py:110).
#ifdef SPLIT_STAMPS      // diagnostic build only (tools/exp/lstm_step_stamps.py): where one workgroup's time goes
__device__ unsigned long long g_split_stamps[8];
#define SSTAMP(k)                                                                  \
  do {                                                                             \
    if (blockIdx.x == 9 && threadIdx.x == 0) {                                     \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                  \
      g_split_stamps[k] += t_ - t_prev_;                                           \
      t_prev_ = t_;                                                                \
    }                                                                              \
  } while (0)
#else
#define SSTAMP(k)
#endif

template <bool DUAL>
struct ASrc { const float* P; int ld, K, k0; };

template <bool DUAL>
__device__ __forceinline__ ASrc<DUAL> a_src(const SplitArgs& p, int kg) {
  ASrc<DUAL> s;
  if (DUAL) {
    const bool second = kg >= p.K1pad;
    s.P = second ? p.A2 : p.A; s.ld = second ? p.lda2 : p.lda;
    s.K = second ? p.K - p.K1pad : p.K1; s.k0 = second ? kg - p.K1pad : kg;
  } else {
    s.P = p.A; s.ld = p.lda; s.K = p.K; s.k0 = kg;
  }
  return s;
}

// RAG false (K a multiple of 32, one wave group, one A source: the big products of the trainer): no k of a tile lies past
// K, the per-element selects of the A split go
template <int BM, int BN, bool VEC, bool DEEP = (BM == 64), int EPI = 0, int KW = 1, bool DUAL = false, bool RAG = true>
__global__ __launch_bounds__(256 * KW, KW == 1 ? ((BM == 64 && BN == 128) ? 4 : (SPLIT_NT_OCC3 && BM == 128 && EPI == 0) ? 3 : 2) : 1) void gemm_split_nt_kernel(SplitArgs p) {
  constexpr int MODE = SPLIT_NT_MODE, NPL = npl(MODE);
  typedef typename Frag<MODE>::type frag8;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int PA = BM * BK / 1024, PW = NPL * BN / 64;
  constexpr int A_BYTES = NPL * BM * ROW_B, B_BYTES = NPL * BN * ROW_B;
  // fp16x2: the tensors' power-of-two scales (exact), read once; everything below is scale-free in the bf16x3 mode
  float SA = 1.f, INV_A = 1.f, INV_W = 1.f;
  if (MODE == MODE_F16X2) {
    SA = pow2_scale(fmaxf(p.a_absmax ? *p.a_absmax : 0.f, p.a_floor));
    INV_A = pow2_inv(SA);
    INV_W = pow2_inv(pow2_scale(*p.w_absmax));
  }
  constexpr bool HALF = SPLIT_NT_OCC3 && BM == 128 && EPI == 0 && KW == 1;      // epilogue staged in two 64-row halves
  constexpr int C_BYTES = (HALF ? BM / 2 : BM) * (BN + 4) * 4;
  // DBUF: two operand images -- tile it + 1 is split / stored into the other one while tile it is multiplied, so a K tile
  // costs ONE workgroup barrier (stores visible) instead of two (all reads done; stores visible).  128 x 128: 2 x 40,960 B
  // per workgroup, two workgroups per CU use the 160 KB exactly.
  constexpr int NBUF = SPLIT_NT_DBUF ? 2 : 1;
  constexpr int OP_BYTES = A_BYTES + B_BYTES;
  constexpr int RED_BYTES = 64;                  // the epilogues' per-wave maxima (absmax commits) live behind the C tile
  constexpr int SMEM = (NBUF * OP_BYTES) > (C_BYTES + RED_BYTES) ? (NBUF * OP_BYTES) : (C_BYTES + RED_BYTES);
  static_assert(4 * KW * 4 <= RED_BYTES, "one float per wave");
  __shared__ __attribute__((aligned(16))) unsigned char smem_all[KW * SMEM];
#ifdef SPLIT_STAMPS
  unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
  const int tid = KW == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 255);
  const int grp = KW == 1 ? 0 : (int)(threadIdx.x >> 8);
  unsigned char* smem = smem_all + grp * SMEM;
  unsigned char* As = smem;
  unsigned char* Bs = smem + A_BYTES;

  // XCD-aware block order: workgroup L runs on XCD L % 8; slot s = L / 8 of that XCD sweeps x first, and the row
  // panels are dealt to the XCDs round-robin, so the nbx blocks that share one A panel follow each other on one L2.
  const int L = blockIdx.x, xcd = L & 7, s = L >> 3;
  const int slab = s % p.splitk, s2 = s / p.splitk;     // K slabs of one tile follow each other on the XCD
  const int bx = s2 % p.nbx, by = (s2 / p.nbx) * 8 + xcd;
  if (by >= p.nby) return;
  const int m0 = by * BM, n0 = bx * BN;
  if (EPI == 0) p.C += (size_t)slab * p.slab_stride;    // partial products of the K slabs side by side (0: one C)
  const int kt_base = slab * p.ktiles_per_split;
  const int nkt = min((p.K + BK - 1) / BK - kt_base, p.ktiles_per_split);
  if (nkt <= 0) return;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[2][PA];      // DEEP: A pieces travel TWO tiles ahead (set it&1 holds tile it+1); W pieces always one (L2)
  u32x4 rw[PW];
  const int shift = (int)((unsigned)by * 7u % (unsigned)nkt);    // de-synchronise the panel sweeps of different rows
  // this group's i-th K tile is tile i * KW + grp of the slab; past the end: a valid tile is fetched and stored as zeros
#define KT_AT(i) (kt_base + (((i) + shift) >= nkt ? (i) + shift - nkt : (i) + shift))
#define KG_AT(i) (KT_AT(min((i) * KW + grp, nkt - 1)) * BK)
#define KLIM(i, src) ((KW == 1 || (i) * KW + grp < nkt) ? (src).K - (src).k0 : 0)
#define A_LOAD(src, q) a_piece_load<BM, VEC>((src).P, (src).ld, p.M, (src).K, m0, (src).k0, q, tid)
  const int nit = (nkt + KW - 1) / KW;
  {
    const int k0 = KG_AT(0);
    const ASrc<DUAL> s0 = a_src<DUAL>(p, k0);
#pragma unroll
    for (int q = 0; q < PA; ++q) ra[0][q] = A_LOAD(s0, q);
#pragma unroll
    for (int q = 0; q < PW; ++q) rw[q] = w_piece_load<BN>(p.W, p.ldw, p.plane, p.N, n0, k0, q, tid);
#pragma unroll
    for (int q = 0; q < PA; ++q) a_piece_store<BM, MODE, RAG>(As, ra[0][q], KLIM(0, s0), q, tid, SA);
#pragma unroll
    for (int q = 0; q < PW; ++q) w_piece_store<BN>(Bs, rw[q], q, tid);
    const int k1 = KG_AT(1), k2 = KG_AT(2);
    const ASrc<DUAL> s1 = a_src<DUAL>(p, k1), s2_ = a_src<DUAL>(p, k2);
#pragma unroll
    for (int q = 0; q < PA; ++q) ra[0][q] = A_LOAD(s1, q);
    if (DEEP) {
#pragma unroll
      for (int q = 0; q < PA; ++q) ra[1][q] = A_LOAD(s2_, q);
    }
#pragma unroll
    for (int q = 0; q < PW; ++q) rw[q] = w_piece_load<BN>(p.W, p.ldw, p.plane, p.N, n0, k1, q, tid);
  }
  __syncthreads();
  SSTAMP(0);       // prologue: first tile fetched, split, stored; next tiles requested

  // One K tile.  Tile it+1 leaves register set S (split, LDS store) and tile it+3 is fetched into it, piece by piece,
  // next to the second-half MFMAs: loads spread between MFMAs keep the CU's vector-memory queue short (8 waves
  // issuing 10 loads each at once stall at issue, and the in-order MFMAs behind them with it).  An A fetch then has
  // TWO iterations to land: with one, the streaming operand's HBM time added to the MFMA time instead of hiding under
  // it (all resident blocks ask for their next tile in the same phase).  Everything is unconditional -- the last passes
  // re-store / re-fetch a valid tile that is never read -- so the section stays ONE basic block and the compiler
  // interleaves it (explicit sched_group_barrier pipelines measured 5-10 % slower than its own schedule).
#define SPLIT_NT_TILE(S, IT)                                                                                        \
  {                                                                                                                 \
    frag8 af[TM][NPL], bf[TN][NPL];                                                                                  \
    if (!(SPLIT_ABLATE & 4)) {                                                                                      \
      read_frags<BM, BN, MODE>(As, Bs, wm, wn, li, kh, 0, af, bf);                                                  \
      mma_frags<TM, TN>(af, bf, acc);                                                                               \
      read_frags<BM, BN, MODE>(As, Bs, wm, wn, li, kh, 1, af, bf);   /* second half: read before, multiplied after */ \
    }                                                                                                               \
    __syncthreads();                                   /* every wave has read tile IT */                            \
    const int kcur = KG_AT((IT) + 1), kw = KG_AT((IT) + 2), ka = KG_AT((IT) + (DEEP ? 3 : 2));                      \
    const ASrc<DUAL> scur = a_src<DUAL>(p, kcur), snext = a_src<DUAL>(p, ka);                                       \
    const int klim = KLIM((IT) + 1, scur);                                                                          \
    _Pragma("unroll") for (int q = 0; q < PA; ++q) {                                                                \
      if (!(SPLIT_ABLATE & 2) || (IT) == 0) a_piece_store<BM, MODE, RAG>(As, ra[S][q], klim, q, tid, SA);                \
      if (!(SPLIT_ABLATE & 1)) ra[S][q] = A_LOAD(snext, q);                                                         \
    }                                                                                                               \
    _Pragma("unroll") for (int q = 0; q < PW; ++q) {                                                                \
      if (!(SPLIT_ABLATE & 2) || (IT) == 0) w_piece_store<BN>(Bs, rw[q], q, tid);                                   \
      if (!(SPLIT_ABLATE & 1)) rw[q] = w_piece_load<BN>(p.W, p.ldw, p.plane, p.N, n0, kw, q, tid);                  \
    }                                                                                                               \
    if (!(SPLIT_ABLATE & 4)) mma_frags<TM, TN>(af, bf, acc);                                                        \
    __syncthreads();                                   /* tile IT + 1 is visible */                                 \
  }
  // DBUF form of one K tile: fragments of tile IT come from image IT & 1; tile IT + 1 goes into the other image at any
  // point of the step (everybody left that image before the previous step's barrier), so the split VALU, the LDS stores
  // and the next global loads spread over all 24 MFMAs; the one barrier sits before the second half's MFMAs, which then
  // run while the slower waves arrive.
#define SPLIT_NT_TILE_DB(S, IT)                                                                                     \
  {                                                                                                                 \
    frag8 af[TM][NPL], bf[TN][NPL];                                                                                  \
    const unsigned char* Ac = As + ((IT) & 1) * OP_BYTES;                                                           \
    const unsigned char* Bc = Bs + ((IT) & 1) * OP_BYTES;                                                           \
    unsigned char* An = As + (((IT) + 1) & 1) * OP_BYTES;                                                           \
    unsigned char* Bn = Bs + (((IT) + 1) & 1) * OP_BYTES;                                                           \
    read_frags<BM, BN, MODE>(Ac, Bc, wm, wn, li, kh, 0, af, bf);                                                    \
    const int kcur = KG_AT((IT) + 1), kw = KG_AT((IT) + 2), ka = KG_AT((IT) + (DEEP ? 3 : 2));                      \
    const ASrc<DUAL> scur = a_src<DUAL>(p, kcur), snext = a_src<DUAL>(p, ka);                                       \
    const int klim = KLIM((IT) + 1, scur);                                                                          \
    _Pragma("unroll") for (int q = 0; q < PA; ++q) {                                                                \
      a_piece_store<BM, MODE, RAG>(An, ra[S][q], klim, q, tid, SA);                                                      \
      ra[S][q] = A_LOAD(snext, q);                                                                                  \
    }                                                                                                               \
    _Pragma("unroll") for (int q = 0; q < PW; ++q) {                                                                \
      w_piece_store<BN>(Bn, rw[q], q, tid);                                                                         \
      rw[q] = w_piece_load<BN>(p.W, p.ldw, p.plane, p.N, n0, kw, q, tid);                                           \
    }                                                                                                               \
    mma_frags<TM, TN>(af, bf, acc);                                                                                 \
    read_frags<BM, BN, MODE>(Ac, Bc, wm, wn, li, kh, 1, af, bf);                                                    \
    __syncthreads();                                   /* tile IT + 1 is visible; everybody has read tile IT */      \
    mma_frags<TM, TN>(af, bf, acc);                                                                                 \
  }
  if (SPLIT_NT_DBUF) {
    if (DEEP) {
      for (int it = 0; it < nit; it += 2) {
        SPLIT_NT_TILE_DB(0, it)
        if (it + 1 < nit) SPLIT_NT_TILE_DB(1, it + 1)
      }
    } else {
      for (int it = 0; it < nit; ++it) SPLIT_NT_TILE_DB(0, it)
    }
    __syncthreads();     // the epilogue parks its tile over the operand images: every wave must have read the last one
  } else if (DEEP) {
    for (int it = 0; it < nit; it += 2) {
      SPLIT_NT_TILE(0, it)
      if (it + 1 < nit) SPLIT_NT_TILE(1, it + 1)
    }
  } else {
    for (int it = 0; it < nit; ++it) SPLIT_NT_TILE(0, it)
  }
#undef SPLIT_NT_TILE
#undef SPLIT_NT_TILE_DB
  SSTAMP(1);       // K loop
#undef A_LOAD
#undef KLIM
#undef KG_AT
#undef KT_AT

  if (EPI == 0 && (p.flags & FLAG_ATOMIC)) {
    // split-K: partial tile added into C (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + li;
        if (col >= p.N) continue;
        const float bv = (p.bias && slab == 0 && grp == 0) ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const float v = MODE == MODE_F16X2 ? (acc[i][j][r] * INV_A) * INV_W : acc[i][j][r];
          if (row < p.M) atomicAdd(p.C + (size_t)row * p.ldc + col, v + bv);
        }
      }
    return;
  }
  // epilogue through LDS: every lane moves 16 B of one row (same as gemm.hip; storing the 32 x 32 accumulator registers
  // as they stand -- 2 x 128 B per instruction, no LDS round trip -- measured 0.55-0.9x: profiles/r03_gemm_nt_ablate.log);
  // with KW groups each parks its partial
  // tile in its own LDS image and the partials are summed group 0 first
  constexpr int CLD = BN + 4;
  float* Cs = reinterpret_cast<float*>(smem);
  if (HALF) {
    // (EPI == 0, one wave group) the 128 x 128 tile leaves in two 64-row halves through a 33,792-byte staging area
    const uint16_t* mbits = reinterpret_cast<const uint16_t*>(p.mask);
    const bool vecC = ((p.ldc & 3) == 0) && ((((uintptr_t)p.C) & 15) == 0) &&
                      (!p.bias || ((((uintptr_t)p.bias) & 15) == 0)) &&
                      (!(p.flags & FLAG_RELU_MASK) || (((p.ldm & 3) == 0) && ((((uintptr_t)p.mask) & 15) == 0)));
    constexpr int CV = BN / 4;
    float omax = 0.f;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      if (half) __syncthreads();                        // every lane has stored its pieces of the first half
      if (wm == half) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * CLD + wn * (BN / 2) + j * 32 + li] =
                  MODE == MODE_F16X2 ? (acc[i][j][r] * INV_A) * INV_W : acc[i][j][r];
      }
      __syncthreads();
      for (int id = threadIdx.x; id < (BM / 2) * CV; id += 256) {
        const int r = id / CV, c4 = (id % CV) * 4;
        const int row = m0 + half * (BM / 2) + r, col = n0 + c4;
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
          if (p.flags & FLAG_RELU_BITS) {
            const unsigned w = mbits[(size_t)row * p.ldm + (col >> 4)] >> (col & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((w >> e) & 1u) ? v[e] : 0.f;
          }
          *reinterpret_cast<f32x4*>(cp) = v;
          omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col + e >= p.N) break;
            float x = v[e] + (p.bias ? p.bias[col + e] : 0.f);
            if (p.flags & FLAG_ACCUM) x += cp[e];
            if (p.flags & FLAG_RELU) x = fmaxf(x, 0.f);
            if (p.flags & FLAG_RELU_MASK) x = (p.mask[(size_t)row * p.ldm + col + e] > 0.f) ? x : 0.f;
            if (p.flags & FLAG_RELU_BITS) x = ((mbits[(size_t)row * p.ldm + ((col + e) >> 4)] >> ((col + e) & 15)) & 1u) ? x : 0.f;
            cp[e] = x;
            omax = fmaxf(omax, fabsf(x));
          }
        }
      }
    }
    if (p.c_absmax0) {
      __syncthreads();                                  // the staging area is free again
      float* wmx0 = reinterpret_cast<float*>(smem_all);
      omax = wave_max(omax);
      if ((threadIdx.x & 63) == 0) wmx0[threadIdx.x >> 6] = omax;
      __syncthreads();
      if (threadIdx.x < 64) absmax_commit(p.c_absmax0, fmaxf(fmaxf(wmx0[0], wmx0[1]), fmaxf(wmx0[2], wmx0[3])));
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * CLD + wn * (BN / 2) + j * 32 + li] =
            MODE == MODE_F16X2 ? (acc[i][j][r] * INV_A) * INV_W : acc[i][j][r];
  __syncthreads();
  SSTAMP(2);       // partial tiles parked in LDS
  const float* C0 = reinterpret_cast<const float*>(smem_all);
  constexpr int GSTRIDE = SMEM / 4;                   // floats between the partial tiles of consecutive groups
  if (EPI == 1) {
    // BasicLSTMCell gate math fused into the recurrent GEMM (model/model.py:110, TF BasicLSTMCell: gates i,j,f,o,
    // forget_bias 1).  The weight shadow is gate-interleaved (unreal_split_bf16x3 row_perm = 1): this block's 64
    // columns are hidden units u0..u0+15 of the four gates, so the tile holds everything one unit needs.
    // FLAG_ACCUM: C holds the input-half pre-activations on entry (added after the product, as the chain does).
    const int u0 = bx * (BN / 4);
    const bool add_pre = (p.flags & FLAG_ACCUM) != 0;
    constexpr int ITEMS = BM * (BN / 4) / (256 * KW);            // (row, unit) pairs per lane: 4 (64 x 64) or 8
    // every global read of the epilogue is issued before the first gate is computed: C and c_prev may alias the
    // outputs as far as the compiler knows, so left in the loop each pair would wait for its own loads in turn
    float cprev[ITEMS], pre4[ITEMS][4];
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
      const int id = threadIdx.x + e * 256 * KW;
      const int row = min(m0 + id / (BN / 4), p.M - 1), u = u0 + id % (BN / 4);
      cprev[e] = p.c_prev[(size_t)row * 256 + u];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) pre4[e][g4] = add_pre ? p.C[(size_t)row * p.ldc + g4 * 256 + u] : 0.f;
    }
    static_assert((256 * KW) % (BN / 4) == 0, "a lane keeps its hidden unit over the pairs it handles");
    float bb[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) bb[g4] = p.bias[g4 * 256 + u0 + threadIdx.x % (BN / 4)];
#pragma unroll
    for (int e = 0; e < ITEMS; ++e) {
      const int id = threadIdx.x + e * 256 * KW;
      const int r = id / (BN / 4), ul = id % (BN / 4);
      const int row = m0 + r, u = u0 + ul;
      const float* cs = C0 + r * CLD + (ul >> 4) * 64 + (ul & 15);      // 16 units x 4 gates per 64 columns
      float s4[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float a = cs[g4 * 16];
#pragma unroll
        for (int g = 1; g < KW; ++g) a += cs[g * GSTRIDE + g4 * 16];
        s4[g4] = add_pre ? a + pre4[e][g4] : a;
      }
      const float gi = 1.f / (1.f + expf(-(s4[0] + bb[0])));
      const float gj = tanhf(s4[1] + bb[1]);
      const float gf = 1.f / (1.f + expf(-(s4[2] + bb[2] + 1.0f)));
      const float go = 1.f / (1.f + expf(-(s4[3] + bb[3])));
      const float c = cprev[e] * gf + gi * gj;
      if (row >= p.M) continue;
      float* pre = p.C + (size_t)row * p.ldc;
      pre[u] = gi; pre[256 + u] = gj; pre[512 + u] = gf; pre[768 + u] = go;
      p.c_out[(size_t)row * 256 + u] = c;
      p.h_out[(size_t)row * p.ld_h + u] = tanhf(c) * go;
    }
    SSTAMP(3);     // gate math + stores issued
    return;
  }
  if (EPI == 2) {
    // BPTT through one BasicLSTMCell step, fused (unreal_lstm_bptt_step): this tile is dh_rec[row][u] = d_gates(t) . Wh^T
    // for 64 hidden units u; with it the gate backward of step t-1 is purely element-wise -- same arithmetic, in the same
    // order, as unreal_lstm_gates_bwd after the stand-alone product, so the two paths agree bit for bit.
    constexpr int ITEMS = BM * BN / (256 * KW), CH = ITEMS < 4 ? ITEMS : 4;
    float omax = 0.f;
#pragma unroll
    for (int e0 = 0; e0 < ITEMS; e0 += CH) {
      float dha[CH], dcv[CH], cn[CH], cp[CH], ga[CH][4];
#pragma unroll
      for (int e = 0; e < CH; ++e) {                   // all loads of the chunk first (outputs may alias for the compiler)
        const int id = threadIdx.x + (e0 + e) * 256 * KW;
        const size_t g = (size_t)min(m0 + id / BN, p.M - 1) * 256 + n0 + id % BN;
        dha[e] = p.dh_above[g]; dcv[e] = p.dc_io[g]; cn[e] = p.c_new[g]; cp[e] = p.c_prev[g];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) ga[e][g4] = p.gates_act[(g / 256) * 1024 + g4 * 256 + g % 256];
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const int id = threadIdx.x + (e0 + e) * 256 * KW;
        const int r = id / BN, cidx = id % BN;
        float rec = C0[r * CLD + cidx];
#pragma unroll
        for (int g = 1; g < KW; ++g) rec += C0[g * GSTRIDE + r * CLD + cidx];
        if (m0 + r >= p.M) continue;
        const size_t g = (size_t)(m0 + r) * 256 + n0 + cidx;
        const float i = ga[e][0], j = ga[e][1], f = ga[e][2], o = ga[e][3];
        const float dh = dha[e] + rec;
        const float tc = tanhf(cn[e]);
        const float dc = dcv[e] + dh * o * (1.f - tc * tc);
        float* d = p.dpre + (g / 256) * 1024 + g % 256;
        const float d0 = dc * j * i * (1.f - i), d1 = dc * i * (1.f - j * j), d2 = dc * cp[e] * f * (1.f - f),
                    d3 = dh * tc * o * (1.f - o);
        d[0] = d0; d[256] = d1; d[512] = d2; d[768] = d3;
        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(d0), fabsf(d1))), fmaxf(fabsf(d2), fabsf(d3)));
        p.dc_io[g] = dc * f;
      }
    }
    // max |d_gates| of the step just produced: the next step's A scale.  One commit per WORKGROUP: the 4096 waves of a
    // step finish together, all would find the slot at its old value and each issue its (serialised, ~12 ns) atomic
    {
      float* wmx = reinterpret_cast<float*>(smem_all + C_BYTES);     // group 0's image, past its C tile: nobody reads it
      omax = wave_max(omax);
      if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = omax;
      __syncthreads();
      if (threadIdx.x < 64) {
        float m = wmx[0];
#pragma unroll
        for (int g = 1; g < 4 * KW; ++g) m = fmaxf(m, wmx[g]);
        absmax_commit(p.c_absmax0, m);
        absmax_commit(p.c_absmax1, m);
      }
    }
    return;
  }
  // FLAG_RELU_BITS: mask is a bit matrix (uint16 words, row stride ldm words, bit j % 16 of word j / 16 = column j kept)
  const uint16_t* mbits = reinterpret_cast<const uint16_t*>(p.mask);
  const bool vecC = ((p.ldc & 3) == 0) && ((((uintptr_t)p.C) & 15) == 0) &&
                    (!p.bias || ((((uintptr_t)p.bias) & 15) == 0)) &&
                    (!(p.flags & FLAG_RELU_MASK) || (((p.ldm & 3) == 0) && ((((uintptr_t)p.mask) & 15) == 0)));
  constexpr int CV = BN / 4;
  float omax = 0.f;
  static_assert((BM * CV) % (256 * KW) == 0, "every lane makes the same number of passes (absmax_commit needs whole waves)");
  for (int id = threadIdx.x; id < BM * CV; id += 256 * KW) {
    const int r = id / CV, c4 = (id % CV) * 4;
    const int row = m0 + r, col = n0 + c4;
    if (row >= p.M || col >= p.N) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(C0 + r * CLD + c4);
#pragma unroll
    for (int g = 1; g < KW; ++g) v += *reinterpret_cast<const f32x4*>(C0 + g * GSTRIDE + r * CLD + c4);
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
      if (p.flags & FLAG_RELU_BITS) {                 // col is a multiple of 4: the four bits sit in one word
        const unsigned w = mbits[(size_t)row * p.ldm + (col >> 4)] >> (col & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((w >> e) & 1u) ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(cp) = v;
      omax = fmaxf(fmaxf(omax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (col + e >= p.N) break;
        float x = v[e] + (p.bias ? p.bias[col + e] : 0.f);
        if (p.flags & FLAG_ACCUM) x += cp[e];
        if (p.flags & FLAG_RELU) x = fmaxf(x, 0.f);
        if (p.flags & FLAG_RELU_MASK) x = (p.mask[(size_t)row * p.ldm + col + e] > 0.f) ? x : 0.f;
        if (p.flags & FLAG_RELU_BITS) x = ((mbits[(size_t)row * p.ldm + ((col + e) >> 4)] >> ((col + e) & 15)) & 1u) ? x : 0.f;
        cp[e] = x;
        omax = fmaxf(omax, fabsf(x));
      }
    }
  }
  if (p.c_absmax0) {                                    // (block-uniform pointer; every lane left the loop together)
    float* wmx0 = reinterpret_cast<float*>(smem_all + C_BYTES);     // one commit per workgroup (see the BPTT epilogue)
    omax = wave_max(omax);
    if ((threadIdx.x & 63) == 0) wmx0[threadIdx.x >> 6] = omax;
    __syncthreads();
    if (threadIdx.x < 64) {
      float m = wmx0[0];
#pragma unroll
      for (int g = 1; g < 4 * KW; ++g) m = fmaxf(m, wmx0[g]);
      absmax_commit(p.c_absmax0, m);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad:  C[M,N] += A[K,M]^T * B[K,N]   (both operands activations, k = the row index; split-K with fp32 atomics)
//
// Same MFMA core; the difference is the staging.  A thread owns a 4(k) x 4(m) patch of the 32(k) x 128(m) tile: four
// 16-byte loads (one per k row, 512 B contiguous per row across 32 lanes), a register transpose, and per m a split of
// the 4 k values into 3 bf16 planes stored as 8 bytes at [plane][m][k] -- the same LDS image the NT kernel reads.
// The stores of 32 lanes go to rows 4 apart (320 B = 16 banks), so the 16-byte chunk index is XOR-swizzled with row
// bits 4-5: 2-way instead of 8-way bank conflicts; the fragment reads apply the same swizzle and stay conflict-free.
struct SplitTnArgs {
  int M, N, K;
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  float* colsum;                       // nullable: colsum[n] += sum_k B[k][n] (the bias gradient that goes with dW)
  int ntx, nty, splitk, ktiles_per_split;
  const float* a_absmax; const float* b_absmax;     // fp16x2: absmax slots of A and B
};

// Thread -> (m4: which 4 columns of the 128, k4: which 4 k rows of the 32) of the TN tile it stages.  Lane order
// (m4 = tid & 31, k4 = tid >> 5) leaves every ds_write_b64 2-way conflicted (a 16-lane store group spans 16 consecutive
// m4 at one k4: rows 8 apart are 640 B = a multiple of the 128-byte store window; 28 % of the kernel's cycles were
// conflict cycles).  Dealing tid bit 0 to k4 and bits 1-5 to m4 bits 0, 2, 3, 1, 4 makes a store group cover both 8-byte
// halves, both row parities and four different swizzled chunks: 16 distinct slots.  A wave still reads two whole
// 512-byte rows per load instruction.
#ifndef SPLIT_TN_DEAL
#define SPLIT_TN_DEAL 1
#endif
__device__ __forceinline__ int tn_m4() {
  const int t = threadIdx.x;
  return SPLIT_TN_DEAL ? (((t >> 1) & 1) | (((t >> 4) & 1) << 1) | (((t >> 2) & 3) << 2) | (((t >> 5) & 1) << 4)) : (t & 31);
}
__device__ __forceinline__ int tn_k4() {
  const int t = threadIdx.x;
  return SPLIT_TN_DEAL ? ((t & 1) | ((t >> 6) << 1)) : (t >> 5);
}

__device__ __forceinline__ void tn_piece_load(const float* __restrict__ P, int ld, int K, int c0, int k0, f32x4 (&reg)[4]) {
  const int m4 = tn_m4(), k4 = tn_k4();
  const int col = min(c0 + 4 * m4, ld - 4);                 // columns past the matrix: pulled inside, never stored
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int k = min(k0 + 4 * k4 + kk, K - 1);             // rows past K: re-read the last row, zeroed in the store
    reg[kk] = *reinterpret_cast<const f32x4*>(P + (size_t)k * ld + col);
  }
}

// RAGGED false (K a multiple of 32: every production wgrad): no k row of a tile lies past K, the 32 selects per tile go
template <int ROWS, int MODE, bool RAGGED = true>
__device__ __forceinline__ void tn_piece_store(unsigned char* S, const f32x4 (&reg)[4], int klim, float scale) {
  const int m4 = tn_m4(), k4 = tn_k4();
  bool ok[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) ok[kk] = !RAGGED || 4 * k4 + kk < klim;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 4 * m4 + j;
    u32x2 pl[3];
    const float sc = MODE == MODE_F16X2 ? scale : 1.f;
    split4_terms<true, MODE>(ok[0] ? reg[0][j] * sc : 0.f, ok[1] ? reg[1][j] * sc : 0.f, ok[2] ? reg[2][j] * sc : 0.f,
                             ok[3] ? reg[3][j] * sc : 0.f, pl);
    const int off = (((k4 >> 1) ^ ((r >> 4) & 3)) * 2 + (k4 & 1)) * 8;       // swizzled 16-B chunk, 8-B half
#pragma unroll
    for (int t = 0; t < npl(MODE); ++t)
      *reinterpret_cast<u32x2*>(S + (t * ROWS + r) * ROW_B + off) = pl[t];
  }
}

// column sums of the B tile a thread holds (rows past K masked), weighted by w (0 or 1)
template <bool RAGGED = true>
__device__ __forceinline__ void tn_colsum_acc(const f32x4 (&reg)[4], int klim, float w, float (&cs)[4]) {
  const int k4 = tn_k4();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const float wk = (!RAGGED || 4 * k4 + kk < klim) ? w : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = fmaf(wk, reg[kk][j], cs[j]);
  }
}

template <bool RAGGED>
__global__ __launch_bounds__(256, 2) void gemm_split_tn_kernel(SplitTnArgs p) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  constexpr int MODE = SPLIT_TN_MODE, NPL = npl(MODE);
  typedef typename Frag<MODE>::type frag8;
  constexpr int A_BYTES = NPL * BM * ROW_B, B_BYTES = NPL * BN * ROW_B;
#ifndef SPLIT_TN_DBUF
#define SPLIT_TN_DBUF 0     // 1: two LDS operand images, one barrier per K tile (as SPLIT_NT_DBUF)
#endif
  constexpr int OP_BYTES = A_BYTES + B_BYTES, NBUF = SPLIT_TN_DBUF ? 2 : 1;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * OP_BYTES > 8 * BN * 4 ? NBUF * OP_BYTES : 8 * BN * 4];
  float SA = 1.f, SB = 1.f, INV_A = 1.f, INV_B = 1.f;
  if (MODE == MODE_F16X2) {
    SA = pow2_scale(*p.a_absmax); SB = pow2_scale(*p.b_absmax);
    INV_A = pow2_inv(SA); INV_B = pow2_inv(SB);
  }
  unsigned char* As = smem;
  unsigned char* Bs = smem + A_BYTES;

  // XCD-aware order: one K slab per XCD at a time, its C tiles swept x-fastest, so the slab of B (re-read by every
  // row of tiles) and the slab of A (re-read by every column) are served by that XCD's L2.
  const int L = blockIdx.x, xcd = L & 7, s = L >> 3;
  const int tiles = p.ntx * p.nty;
  const int slab = (s / tiles) * 8 + xcd, tile = s % tiles;
  if (slab >= p.splitk) return;
  const int bx = tile % p.ntx, by = tile / p.ntx;
  const int m0 = by * BM, n0 = bx * BN;
  const int nk_total = (p.K + BK - 1) / BK;
  const int kt0 = slab * p.ktiles_per_split, kt1 = min(nk_total, kt0 + p.ktiles_per_split);
  if (kt0 >= kt1) return;
  const int nkt = kt1 - kt0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifndef SPLIT_TN_FLUSH
#define SPLIT_TN_FLUSH 8      // K tiles between two flushes of the MFMA accumulators (0: never)
#endif
  // Long-K accuracy: the MFMA adds its 16 products into the running accumulator with a rounding that is biased (measured:
  // at K = 81,920 the error of a 3,400-long chain grows linearly with the chain, fp16x2 4.6e-7 and bf16x3 1.5e-6 rms against
  // 2.3e-7 for the fp32 MFMA's round-to-nearest chain).  So every SPLIT_TN_FLUSH tiles the accumulators are added into a
  // second fp32 set with ordinary VALU adds (round to nearest) and restarted from zero: the MFMA chains stay short.
  f32x16 tot[TM][TN];
  if (SPLIT_TN_FLUSH > 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  }
  f32x4 ra[4], rb[4];
  // bias gradient: the row of tiles by == 0 also sums the columns of every B tile it stages (multiplying by 0
  // elsewhere keeps the K loop one basic block)
  const float csw = (p.colsum != nullptr && by == 0) ? 1.f : 0.f;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  tn_piece_load(p.A, p.lda, p.K, m0, kt0 * BK, ra);
  tn_piece_load(p.B, p.ldb, p.K, n0, kt0 * BK, rb);
  tn_piece_store<BM, MODE, RAGGED>(As, ra, p.K - kt0 * BK, SA);
  tn_piece_store<BN, MODE, RAGGED>(Bs, rb, p.K - kt0 * BK, SB);
  tn_colsum_acc<RAGGED>(rb, p.K - kt0 * BK, csw, cs);
  {
    const int k1 = (kt0 + min(1, nkt - 1)) * BK;
    tn_piece_load(p.A, p.lda, p.K, m0, k1, ra);
    tn_piece_load(p.B, p.ldb, p.K, n0, k1, rb);
  }
  __syncthreads();

  if (SPLIT_TN_DBUF) {
    for (int it = 0; it < nkt; ++it) {
      frag8 af[TM][NPL], bf[TN][NPL];
      const unsigned char* Ac = As + (it & 1) * OP_BYTES;
      const unsigned char* Bc = Bs + (it & 1) * OP_BYTES;
      unsigned char* An = As + ((it + 1) & 1) * OP_BYTES;
      unsigned char* Bn = Bs + ((it + 1) & 1) * OP_BYTES;
      read_frags<BM, BN, MODE, true>(Ac, Bc, wm, wn, li, kh, 0, af, bf);
      const int kcur = (kt0 + min(it + 1, nkt - 1)) * BK, knext = (kt0 + min(it + 2, nkt - 1)) * BK;
      tn_piece_store<BM, MODE, RAGGED>(An, ra, p.K - kcur, SA);
      tn_piece_load(p.A, p.lda, p.K, m0, knext, ra);
      tn_piece_store<BN, MODE, RAGGED>(Bn, rb, p.K - kcur, SB);
      tn_colsum_acc<RAGGED>(rb, p.K - kcur, (it + 1 < nkt) ? csw : 0.f, cs);
      tn_piece_load(p.B, p.ldb, p.K, n0, knext, rb);
      mma_frags<TM, TN>(af, bf, acc);
      read_frags<BM, BN, MODE, true>(Ac, Bc, wm, wn, li, kh, 1, af, bf);
      __syncthreads();                                 // tile it + 1 is visible; everybody has read tile it
      mma_frags<TM, TN>(af, bf, acc);
      if (SPLIT_TN_FLUSH > 0 && (it + 1) % SPLIT_TN_FLUSH == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
      }
    }
    __syncthreads();                                   // the colsum reduction below reuses the operand images
  } else
  for (int it = 0; it < nkt; ++it) {
    frag8 af[TM][NPL], bf[TN][NPL];
    read_frags<BM, BN, MODE, true>(As, Bs, wm, wn, li, kh, 0, af, bf);
    mma_frags<TM, TN>(af, bf, acc);
    read_frags<BM, BN, MODE, true>(As, Bs, wm, wn, li, kh, 1, af, bf);
    __syncthreads();                                   // every wave has read tile `it`
    // unconditional like the NT kernel: one basic block, the compiler interleaves split VALU, loads and MFMAs
    const int kcur = (kt0 + min(it + 1, nkt - 1)) * BK, knext = (kt0 + min(it + 2, nkt - 1)) * BK;
    tn_piece_store<BM, MODE, RAGGED>(As, ra, p.K - kcur, SA);
    tn_piece_load(p.A, p.lda, p.K, m0, knext, ra);
    tn_piece_store<BN, MODE, RAGGED>(Bs, rb, p.K - kcur, SB);
    tn_colsum_acc<RAGGED>(rb, p.K - kcur, (it + 1 < nkt) ? csw : 0.f, cs);      // the last pass re-stages a tile already counted
    tn_piece_load(p.B, p.ldb, p.K, n0, knext, rb);
    mma_frags<TM, TN>(af, bf, acc);
    if (SPLIT_TN_FLUSH > 0 && (it + 1) % SPLIT_TN_FLUSH == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
    }
    __syncthreads();                                   // tile `it + 1` is visible
  }
  if (SPLIT_TN_FLUSH > 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += tot[i][j][r];
  }

  if (csw != 0.f) {      // block-uniform; the K loop ended with a barrier, so the operand LDS is free
    float* red = reinterpret_cast<float*>(smem);
    const int m4 = tn_m4(), k4 = tn_k4();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[k4 * BN + 4 * m4 + j] = cs[j];
    __syncthreads();
    if (threadIdx.x < BN && n0 + threadIdx.x < p.N) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) v += red[k * BN + threadIdx.x];
      atomicAdd(p.colsum + n0 + threadIdx.x, v);
    }
  }
  // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); 32 lanes = 128 B per atomic
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + li;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < p.M) atomicAdd(p.C + (size_t)row * p.ldc + col, MODE == MODE_F16X2 ? (acc[i][j][r] * INV_A) * INV_B : acc[i][j][r]);
      }
    }
}

// weights -> bf16x3 planes: dst[t][r][c] = term_t(src[r][c]) (transpose = 0) or dst[t][c][r] (transpose = 1);
// 32x32 tiles through LDS so both sides stay coalesced.  Padding columns of dst are left as the caller zeroed them.
__global__ __launch_bounds__(256) void split_planes_kernel(int rows, int cols, const float* __restrict__ src, int ld_src,
                                                           int transpose, unsigned short* __restrict__ dst, int ld_dst,
                                                           long plane, int row_perm) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int y = ty; y < 32; y += 8)
    if (r0 + y < rows && c0 + tx < cols) t[y][tx] = src[(size_t)(r0 + y) * ld_src + c0 + tx];
  __syncthreads();
  for (int y = ty; y < 32; y += 8) {
    int orow = transpose ? c0 + y : r0 + y;
    const int ocol = transpose ? r0 + tx : c0 + tx;
    if (row_perm == 1) orow = ((orow & 255) >> 4) * 64 + (orow >> 8) * 16 + (orow & 15);   // LSTM gate interleave
    const bool ok = transpose ? (c0 + y < cols && r0 + tx < rows) : (r0 + y < rows && c0 + tx < cols);
    if (!ok) continue;
    float x = transpose ? t[tx][y] : t[y][tx];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const __bf16 h = (__bf16)x;
      dst[k * plane + (size_t)orow * ld_dst + ocol] = __builtin_bit_cast(unsigned short, h);
      x -= (float)h;
    }
  }
}

// weights -> fp16x2 planes (hi, lo of src * 2^k, k from the weight's absmax slot): same tiling / layouts as above
__global__ __launch_bounds__(256) void split_planes_f16x2_kernel(int rows, int cols, const float* __restrict__ src, int ld_src,
                                                                 int transpose, unsigned short* __restrict__ dst, int ld_dst,
                                                                 long plane, int row_perm, const float* __restrict__ w_absmax) {
  __shared__ float t[32][33];
  const float sw = pow2_scale(*w_absmax);
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int y = ty; y < 32; y += 8)
    if (r0 + y < rows && c0 + tx < cols) t[y][tx] = src[(size_t)(r0 + y) * ld_src + c0 + tx];
  __syncthreads();
  for (int y = ty; y < 32; y += 8) {
    int orow = transpose ? c0 + y : r0 + y;
    const int ocol = transpose ? r0 + tx : c0 + tx;
    if (row_perm == 1) orow = ((orow & 255) >> 4) * 64 + (orow >> 8) * 16 + (orow & 15);   // LSTM gate interleave
    const bool ok = transpose ? (c0 + y < cols && r0 + tx < rows) : (r0 + y < rows && c0 + tx < cols);
    if (!ok) continue;
    const float x = (transpose ? t[tx][y] : t[y][tx]) * sw;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    dst[(size_t)orow * ld_dst + ocol] = __builtin_bit_cast(unsigned short, h);
    dst[plane + (size_t)orow * ld_dst + ocol] = __builtin_bit_cast(unsigned short, l);
  }
}

// slot = max(slot, max |x[r][c]|): the stand-alone reduction for tensors whose producer does not commit its own maximum
__global__ __launch_bounds__(256) void absmax_kernel(int rows, int cols, const float* __restrict__ x, int ld, float* slot) {
  float m = 0.f;
  if (ld == cols && (cols & 3) == 0 && ((((uintptr_t)x) & 15) == 0)) {      // contiguous: 16 bytes per lane
    const long n4 = (long)rows * cols / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const f32x4 v = x4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
  } else {                                     // a column window of a wider matrix: one row per wave at a time
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < rows; r += nw)
      for (int c = lane; c < cols; c += 64) m = fmaxf(m, fabsf(x[(size_t)r * ld + c]));
  }
  // a short kernel's waves all finish together and would all find the slot still at its old value (one ~12 ns atomic
  // each, serialised: 2048 waves = 25 us for a 2.6 MB weight): one commit per WORKGROUP, and few workgroups
  __shared__ float wm[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) absmax_commit(slot, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}

// ---- all weight shadows of a network in ONE pass (round 4) ---------------------------------------------------------
// A learner update re-splits nine weight matrices; done matrix by matrix that is ~28 launches of ~5 us (slot fill,
// maximum, split) in front of every pass -- 0.14 ms of a 22 ms call, 4-5 % of a 3 ms pass at 8 actors.  Here: descriptor
// tables on the device (built once by the host side), one launch that reduces every matrix's maximum, one that writes
// every matrix's planes.  Same arithmetic per element as absmax_kernel / split_planes_f16x2_kernel: identical bits.
struct MultiAbsDesc { const float* src; float* wmax; long n; long block0; };                    // contiguous matrix, 8192 floats per block
struct MultiSplitDesc { const float* src; unsigned short* dst; const float* wmax; long rows, cols, ld_src, transpose, row_perm, ld_dst, plane, tiles_x, block0; };

__global__ __launch_bounds__(256) void multi_absmax_kernel(const MultiAbsDesc* __restrict__ descs, int n_desc) {
  int d = 0;
  while (d + 1 < n_desc && (long)blockIdx.x >= descs[d + 1].block0) ++d;
  const MultiAbsDesc D = descs[d];
  const long e0 = ((long)blockIdx.x - D.block0) * 8192, e1 = min(D.n, e0 + 8192);
  float m = 0.f;
  for (long i = e0 + threadIdx.x; i < e1; i += 256) m = fmaxf(m, fabsf(D.src[i]));
  __shared__ float wm[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) absmax_commit(D.wmax, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}

__global__ __launch_bounds__(256) void multi_split_kernel(const MultiSplitDesc* __restrict__ descs, int n_desc) {
  int d = 0;
  while (d + 1 < n_desc && (long)blockIdx.x >= descs[d + 1].block0) ++d;
  const MultiSplitDesc D = descs[d];
  __shared__ float t[32][33];
  const float sw = pow2_scale(*D.wmax);
  const long tile = (long)blockIdx.x - D.block0;
  const int c0 = (int)(tile % D.tiles_x) * 32, r0 = (int)(tile / D.tiles_x) * 32;
  const int rows = (int)D.rows, cols = (int)D.cols;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int y = ty; y < 32; y += 8)
    if (r0 + y < rows && c0 + tx < cols) t[y][tx] = D.src[(size_t)(r0 + y) * D.ld_src + c0 + tx];
  __syncthreads();
  for (int y = ty; y < 32; y += 8) {
    int orow = D.transpose ? c0 + y : r0 + y;
    const int ocol = D.transpose ? r0 + tx : c0 + tx;
    if (D.row_perm == 1) orow = ((orow & 255) >> 4) * 64 + (orow >> 8) * 16 + (orow & 15);   // LSTM gate interleave
    const bool ok = D.transpose ? (c0 + y < cols && r0 + tx < rows) : (r0 + y < rows && c0 + tx < cols);
    if (!ok) continue;
    const float x = (D.transpose ? t[tx][y] : t[y][tx]) * sw;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    D.dst[(size_t)orow * D.ld_dst + ocol] = __builtin_bit_cast(unsigned short, h);
    D.dst[D.plane + (size_t)orow * D.ld_dst + ocol] = __builtin_bit_cast(unsigned short, l);
  }
}

}  // namespace

// K slabs of unreal_gemm_f32_split_nt_slabs -> C: the S partial products of an element are added in slab order (fixed:
// bit-reproducible, unlike the atomic epilogue), then bias / ReLU, and max |C| leaves with one commit per workgroup.
// VEC: four columns per thread (N, ldp, ldc multiples of 4, 16-byte aligned pointers).
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int M, int N, int S, const float* __restrict__ part, long slab_stride,
                                                            int ldp, const float* __restrict__ bias, int relu,
                                                            float* __restrict__ C, int ldc, float* c_absmax) {
  constexpr int W = VEC ? 4 : 1;
  const int ncol = (N + W - 1) / W;
  const long total = (long)M * ncol;
  float mx = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int row = (int)(e / ncol), col = (int)(e % ncol) * W;
    const float* src = part + (size_t)row * ldp + col;
    if (VEC) {
      f32x4 v = *reinterpret_cast<const f32x4*>(src);
      for (int sl = 1; sl < S; ++sl) v += *reinterpret_cast<const f32x4*>(src + (size_t)sl * slab_stride);
      if (bias) v += *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (relu) v[k] = fmaxf(v[k], 0.f);
        mx = fmaxf(mx, fabsf(v[k]));
      }
      *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col) = v;
    } else {
      float v = src[0];
      for (int sl = 1; sl < S; ++sl) v += src[(size_t)sl * slab_stride];
      if (bias) v += bias[col];
      if (relu) v = fmaxf(v, 0.f);
      mx = fmaxf(mx, fabsf(v));
      C[(size_t)row * ldc + col] = v;
    }
  }
  if (c_absmax) {                       // block-uniform
    __shared__ float wm[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x < 64) absmax_commit(c_absmax, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
  }
}

extern "C" {

int unreal_absmax_f32(int rows, int cols, const float* x, int ld, float* slot, void* stream) {
  if (rows <= 0 || cols <= 0 || !x || !slot || ld < cols) return UNREAL_EINVAL;
  const long n = (long)rows * cols;
  const int grid = (int)min((n + 8191) / 8192, 512L);        // >= 8 x 16 bytes per lane
  hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rows, cols, x, ld, slot);
  return unreal_launch_status();
}

int unreal_shadow_refresh_multi(const void* abs_descs, int n_abs, int abs_blocks, const void* split_descs, int n_split,
                                int split_blocks, void* stream) {
  if (!abs_descs || !split_descs || n_abs <= 0 || n_split <= 0 || abs_blocks <= 0 || split_blocks <= 0) return UNREAL_EINVAL;
  hipLaunchKernelGGL(multi_absmax_kernel, dim3(abs_blocks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const MultiAbsDesc*>(abs_descs), n_abs);
  hipLaunchKernelGGL(multi_split_kernel, dim3(split_blocks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const MultiSplitDesc*>(split_descs), n_split);
  return unreal_launch_status();
}

int unreal_split_f16x2(int rows, int cols, const float* src, int ld_src, int transpose, int row_perm, uint16_t* dst,
                       int ld_dst, long plane_stride, const float* w_absmax, void* stream) {
  if (rows <= 0 || cols <= 0 || !src || !dst || ld_src < cols || !w_absmax) return UNREAL_EINVAL;
  const int orows = transpose ? cols : rows, ocols = transpose ? rows : cols;
  if (ld_dst < ocols || plane_stride < (long)orows * ld_dst) return UNREAL_EINVAL;
  if (row_perm != 0 && (row_perm != 1 || orows != 1024)) return UNREAL_EINVAL;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(split_planes_f16x2_kernel, grid, dim3(256), 0, (hipStream_t)stream, rows, cols, src, ld_src, transpose,
                     dst, ld_dst, plane_stride, row_perm, w_absmax);
  return unreal_launch_status();
}

// slab_stride != 0 (unreal_gemm_f32_split_nt_slabs): the splitk K slabs leave as plain partial products, slab s at
// C + s * slab_stride; *splitk_eff receives the number of slabs actually used
static int split_nt_launch(int M, int N, int K, const float* A, int lda, const float* a_absmax, const uint16_t* W3, int ldw,
                           long plane_stride, const float* w_absmax, float* C, int ldc, float* c_absmax, const float* bias,
                           const void* mask, int ldm, int flags, int splitk, long slab_stride, int* splitk_eff, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W3 || !C) return UNREAL_EINVAL;
  if (SPLIT_NT_MODE == MODE_F16X2 && (!a_absmax || !w_absmax)) return UNREAL_EINVAL;
  const int kpad = (K + BK - 1) / BK * BK;
  if (lda < K || ldw < kpad || (ldw & 7) || (plane_stride & 7) || plane_stride < (long)N * ldw || ldc < N ||
      (((uintptr_t)W3) & 15))
    return UNREAL_EINVAL;
  if (flags & ~(FLAG_RELU | FLAG_ACCUM | FLAG_ATOMIC | FLAG_RELU_MASK | FLAG_RELU_BITS)) return UNREAL_EINVAL;
  if ((flags & FLAG_RELU_MASK) && (!mask || ldm < N)) return UNREAL_EINVAL;
  if ((flags & FLAG_RELU_BITS) && (!mask || ldm < (N + 15) / 16 || (flags & FLAG_RELU_MASK))) return UNREAL_EINVAL;
  if ((flags & FLAG_ATOMIC) && (flags & (FLAG_RELU | FLAG_RELU_MASK | FLAG_RELU_BITS | FLAG_ACCUM))) return UNREAL_EINVAL;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && !(flags & FLAG_ATOMIC) && !slab_stride) return UNREAL_EINVAL;
  // the atomic epilogue adds partial tiles into C: no workgroup ever sees a finished element, so max |C| cannot be
  // committed there -- a slot left at 0 would silently turn the consumer's scale into 1
  if ((flags & FLAG_ATOMIC) && c_absmax) return UNREAL_EINVAL;
  SplitArgs a;
  a.M = M; a.N = N; a.K = K;
  a.A = A; a.lda = lda; a.W = W3; a.ldw = ldw; a.plane = plane_stride; a.C = C; a.ldc = ldc;
  a.bias = bias; a.mask = static_cast<const float*>(mask); a.ldm = ldm; a.flags = flags;
  a.vecA = ((lda & 3) == 0) && lda >= 4 && ((((uintptr_t)A) & 15) == 0);
  a.c_prev = nullptr; a.c_out = nullptr; a.h_out = nullptr; a.ld_h = 0;
  a.A2 = nullptr; a.lda2 = 0; a.K1 = K; a.K1pad = 0;
  a.dh_above = nullptr; a.dc_io = nullptr; a.gates_act = nullptr; a.c_new = nullptr; a.dpre = nullptr;
  a.a_absmax = a_absmax; a.a_floor = 0.f; a.w_absmax = w_absmax; a.c_absmax0 = c_absmax;
  a.c_absmax1 = nullptr;
  {
    const int nk = (K + BK - 1) / BK;
    if (splitk > nk) splitk = nk;
    a.ktiles_per_split = (nk + splitk - 1) / splitk;
    a.splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
    a.slab_stride = a.splitk > 1 ? slab_stride : 0;
    if (splitk_eff) *splitk_eff = a.splitk;
  }
  const long blocks128 = (long)((M + 127) / 128) * ((N + 127) / 128) * a.splitk;
  // Experiment knobs of round 4 (tools/exp/gemm_ab.py, profiles/r04_ab_summary.md; all default 0, none of their kernels is
  // instantiated in the product build): SPLIT_NT_N256 = 64 x 256 tiles for the N = 256 products (0.89x), SPLIT_NT_K256 =
  // 64 x 128 tiles at four workgroups per CU for the short-K products (0.89x), SPLIT_NT_NORAG = the 128 x 128 kernel without
  // its "k < K" selects when K % 32 == 0 (0.87-0.95x: fewer instructions, worse schedule); SPLIT_FORCE_RAGGED = the masked
  // instantiations everywhere (A/B of the wgrad kernel's non-ragged form, which is kept: +1 %).
#ifndef SPLIT_NT_K256
#define SPLIT_NT_K256 0
#endif
#ifndef SPLIT_NT_N256
#define SPLIT_NT_N256 0
#endif
#ifndef SPLIT_FORCE_RAGGED
#define SPLIT_FORCE_RAGGED 0
#endif
#ifndef SPLIT_NT_NORAG
#define SPLIT_NT_NORAG 0
#endif
#ifndef SPLIT_NT_DEEP128      // A tiles of the 128 x 128 kernel two K tiles ahead (round 3: 0.97-1.02x)
#define SPLIT_NT_DEEP128 0
#endif
  bool launched = false;
#if SPLIT_NT_N256
  if (!launched && blocks128 >= 384 && N > 128 && N <= 256 && a.vecA) {
    a.nbx = 1; a.nby = (M + 63) / 64;
    const int grid = a.splitk * a.nbx * ((a.nby + 7) / 8 * 8);
    hipLaunchKernelGGL((gemm_split_nt_kernel<64, 256, true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    launched = true;
  }
#endif
#if SPLIT_NT_K256
  if (!launched && blocks128 >= 384 && a.ktiles_per_split <= 8 && a.vecA) {
    a.nbx = (N + 127) / 128; a.nby = (M + 63) / 64;
    const int grid = a.splitk * a.nbx * ((a.nby + 7) / 8 * 8);
    hipLaunchKernelGGL((gemm_split_nt_kernel<64, 128, true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    launched = true;
  }
#endif
  if (launched) {
  } else if (blocks128 >= 384) {
    a.nbx = (N + 127) / 128; a.nby = (M + 127) / 128;
    const int grid = a.splitk * a.nbx * ((a.nby + 7) / 8 * 8);
#if SPLIT_NT_NORAG
    if (!SPLIT_FORCE_RAGGED && a.vecA && K % BK == 0)
      hipLaunchKernelGGL((gemm_split_nt_kernel<128, 128, true, SPLIT_NT_DEEP128 != 0, 0, 1, false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else
#endif
    if (a.vecA) hipLaunchKernelGGL((gemm_split_nt_kernel<128, 128, true, SPLIT_NT_DEEP128 != 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_split_nt_kernel<128, 128, false, SPLIT_NT_DEEP128 != 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    a.nbx = (N + 63) / 64; a.nby = (M + 63) / 64;
    const int grid = a.splitk * a.nbx * ((a.nby + 7) / 8 * 8);
    // few tiles (the N = 256 products of a 4096-row step): KW wave groups share each tile's K range
    const long tiles = (long)a.nbx * a.nby * a.splitk;
    const int kw = (tiles <= 256 && a.ktiles_per_split >= 8) ? 4 : (tiles <= 512 && a.ktiles_per_split >= 4) ? 2 : 1;
#define LAUNCH64(VEC_, KW_) hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, VEC_, true, 0, KW_>), dim3(grid), \
                                               dim3(256 * KW_), 0, (hipStream_t)stream, a)
    if (a.vecA) { if (kw == 4) LAUNCH64(true, 4); else if (kw == 2) LAUNCH64(true, 2); else LAUNCH64(true, 1); }
    else { if (kw == 4) LAUNCH64(false, 4); else if (kw == 2) LAUNCH64(false, 2); else LAUNCH64(false, 1); }
#undef LAUNCH64
  }
  return unreal_launch_status();
}

int unreal_gemm_f32_split_nt(int M, int N, int K, const float* A, int lda, const float* a_absmax, const uint16_t* W3, int ldw,
                             long plane_stride, const float* w_absmax, float* C, int ldc, float* c_absmax, const float* bias,
                             const void* mask, int ldm, int flags, int splitk, void* stream) {
  return split_nt_launch(M, N, K, A, lda, a_absmax, W3, ldw, plane_stride, w_absmax, C, ldc, c_absmax, bias, mask, ldm, flags,
                         splitk, 0, nullptr, stream);
}

// Few rows, long K (the fc 2592 -> 256 of a rollout step at <= 1024 rows: 4-64 tiles of 81 dependent K steps on a 256-CU
// chip): the K range is cut into `splitk` slabs that run as separate workgroups and leave their partial products in
// `partials` (splitk x M x pad4(N) floats), a second launch adds the slabs of every element in slab order, applies bias /
// ReLU (flags: 0 or FLAG_RELU) and commits max |C|.  Deterministic (no atomics); the sum order differs from the one-launch
// kernel's, the per-product arithmetic does not.
int unreal_gemm_f32_split_nt_slabs(int M, int N, int K, const float* A, int lda, const float* a_absmax, const uint16_t* W3,
                                   int ldw, long plane_stride, const float* w_absmax, float* C, int ldc, float* c_absmax,
                                   const float* bias, int flags, int splitk, float* partials, long partial_floats,
                                   void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !C || !partials || ldc < N || (flags & ~FLAG_RELU) || splitk < 2) return UNREAL_EINVAL;
  const int ldp = (N + 3) & ~3;
  const long slab_stride = (long)M * ldp;
  const int nk = (K + BK - 1) / BK;
  if (splitk > nk) splitk = nk;
  if (partial_floats < slab_stride * splitk || (((uintptr_t)partials) & 15)) return UNREAL_EINVAL;
  int S = 1;
  const int rc = split_nt_launch(M, N, K, A, lda, a_absmax, W3, ldw, plane_stride, w_absmax, partials, ldp, nullptr, nullptr,
                                 nullptr, 0, 0, splitk, slab_stride, &S, stream);
  if (rc != UNREAL_OK) return rc;
  const bool vec = (N & 3) == 0 && (ldc & 3) == 0 && ((((uintptr_t)C) | ((uintptr_t)bias)) & 15) == 0;
  const long items = (long)M * (vec ? N / 4 : N);
  const int blocks = (int)min((items + 255) / 256, (long)1024);
  if (vec)
    hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, M, N, S, partials,
                       slab_stride, ldp, bias, flags & FLAG_RELU, C, ldc, c_absmax);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, M, N, S, partials,
                       slab_stride, ldp, bias, flags & FLAG_RELU, C, ldc, c_absmax);
  return unreal_launch_status();
}

int unreal_lstm_step_fwd(int rows, const float* x, int ldx, int Kx, const float* x_absmax, const float* h_prev, int ld_hprev,
                         const uint16_t* W3, int ldw, long plane_stride, const float* w_absmax, float* gates, const float* bias,
                         const float* c_prev, float* c_out, float* h_out, int ld_h, void* stream) {
  if (rows <= 0 || !h_prev || !W3 || !gates || !bias || !c_prev || !c_out || !h_out) return UNREAL_EINVAL;
  if (SPLIT_NT_MODE == MODE_F16X2 && (!w_absmax || (x && !x_absmax))) return UNREAL_EINVAL;
  const int kxpad = x ? (Kx + BK - 1) / BK * BK : 0;
  if (x && (Kx <= 0 || ldx < Kx)) return UNREAL_EINVAL;
  if (ld_hprev < 256 || ld_h < 256 || ldw < kxpad + 256 || (ldw & 7) || (plane_stride & 7) ||
      plane_stride < 1024L * ldw || (((uintptr_t)W3) & 15))
    return UNREAL_EINVAL;
  SplitArgs a;
  a.M = rows; a.N = 1024; a.K = kxpad + 256;
  a.W = W3; a.ldw = ldw; a.plane = plane_stride; a.C = gates; a.ldc = 1024;
  a.bias = bias; a.mask = nullptr; a.ldm = 0;
  a.c_prev = c_prev; a.c_out = c_out; a.h_out = h_out; a.ld_h = ld_h;
  a.dh_above = nullptr; a.dc_io = nullptr; a.gates_act = nullptr; a.c_new = nullptr; a.dpre = nullptr;
  // A = [x | h_prev]: |h| < 1 by construction (tanh * sigmoid), so the scale covers max(1, max |x|)
  a.a_absmax = x ? x_absmax : nullptr; a.a_floor = 1.f; a.w_absmax = w_absmax; a.c_absmax0 = nullptr; a.c_absmax1 = nullptr;
  a.splitk = 1; a.ktiles_per_split = a.K / BK; a.slab_stride = 0;
  a.nbx = 16; a.nby = (rows + 63) / 64;
  const int grid = a.nbx * ((a.nby + 7) / 8 * 8);
  const bool vec_h = ((ld_hprev & 3) == 0) && ((((uintptr_t)h_prev) & 15) == 0);
  if (!x) {                              // gates holds the input-half pre-activations: the recurrent half is added
    a.A = h_prev; a.lda = ld_hprev; a.A2 = nullptr; a.lda2 = 0; a.K1 = 256; a.K1pad = 0;
    a.flags = FLAG_ACCUM;
    a.vecA = vec_h;
    if (a.vecA) hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, true, true, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, false, true, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  } else {                               // [x | h_prev] @ the whole kernel in one product
    a.A = x; a.lda = ldx; a.K1 = Kx; a.K1pad = kxpad; a.A2 = h_prev; a.lda2 = ld_hprev;
    a.flags = 0;
    a.vecA = vec_h && ((ldx & 3) == 0) && ldx >= 4 && ((((uintptr_t)x) & 15) == 0);
#ifdef LSTM_FORCE_64
    if (false) {
#else
    if (a.vecA && rows >= 2048) {
#endif
      // a full step (4096 rows): 128 x 128 tiles halve the operand bytes per flop (the 64 x 64 step moves 350 MB from
      // L2 for 13.7 GMAC); two wave groups share each tile's K range so that the 256 tiles still put 8 waves on a CU
      a.nbx = 8; a.nby = (rows + 127) / 128;
      const int grid128 = a.nbx * ((a.nby + 7) / 8 * 8);
#ifndef LSTM_BIG_KW
#define LSTM_BIG_KW 1
#endif
      // more than one tile per CU (the 8192-row steps of the batched replay branches): plain 128 x 128 workgroups, two
      // resident per CU, so one's gate epilogue runs beside the other's K loop
      if (LSTM_BIG_KW == 1 && (long)a.nbx * a.nby > 256)
        hipLaunchKernelGGL((gemm_split_nt_kernel<128, 128, true, false, 1, 1, true>), dim3(grid128), dim3(256), 0, (hipStream_t)stream, a);
      else
        hipLaunchKernelGGL((gemm_split_nt_kernel<128, 128, true, false, 1, 2, true>), dim3(grid128), dim3(512), 0, (hipStream_t)stream, a);
    } else if (a.vecA) {
      // small steps (grouped updates: 512 / 64 rows per launch): 16 x rows/64 tiles of 64 x 64 leave most CUs without a
      // workgroup and one wave per SIMD on the others -- wave groups share a tile's 17 K tiles (as in the plain products)
      const long tiles = (long)a.nbx * a.nby;
      if (tiles <= 256) hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, true, true, 1, 4, true>), dim3(grid), dim3(1024), 0, (hipStream_t)stream, a);
      else if (tiles <= 512) hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, true, true, 1, 2, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, true, true, 1, 1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    } else hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, false, true, 1, 1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  }
  return unreal_launch_status();
}

int unreal_lstm_bptt_step(int rows, const float* d_gates, const float* a_absmax, const uint16_t* Wh3, int ldw, long plane_stride,
                          const float* w_absmax, const float* dh_above, float* dc_io, const float* gates_act,
                          const float* c_prev, const float* c_new, float* dpre, float* dpre_absmax0, float* dpre_absmax1,
                          void* stream) {
  if (rows <= 0 || !d_gates || !Wh3 || !dh_above || !dc_io || !gates_act || !c_prev || !c_new || !dpre) return UNREAL_EINVAL;
  if (SPLIT_NT_MODE == MODE_F16X2 && (!a_absmax || !w_absmax)) return UNREAL_EINVAL;
  if (ldw < 1024 || (ldw & 7) || (plane_stride & 7) || plane_stride < 256L * ldw || (((uintptr_t)Wh3) & 15) ||
      (((uintptr_t)d_gates) & 15))
    return UNREAL_EINVAL;
  SplitArgs a;
  a.M = rows; a.N = 256; a.K = 1024;
  a.A = d_gates; a.lda = 1024; a.W = Wh3; a.ldw = ldw; a.plane = plane_stride; a.C = nullptr; a.ldc = 256;
  a.bias = nullptr; a.mask = nullptr; a.ldm = 0; a.flags = 0; a.vecA = 1;
  a.c_prev = c_prev; a.c_out = nullptr; a.h_out = nullptr; a.ld_h = 0;
  a.A2 = nullptr; a.lda2 = 0; a.K1 = 1024; a.K1pad = 0;
  a.dh_above = dh_above; a.dc_io = dc_io; a.gates_act = gates_act; a.c_new = c_new; a.dpre = dpre;
  a.a_absmax = a_absmax; a.a_floor = 0.f; a.w_absmax = w_absmax; a.c_absmax0 = dpre_absmax0; a.c_absmax1 = dpre_absmax1;
  a.splitk = 1; a.ktiles_per_split = 1024 / BK; a.slab_stride = 0;
  a.nbx = 4; a.nby = (rows + 63) / 64;
  const int grid = a.nbx * ((a.nby + 7) / 8 * 8);
  const long tiles = (long)a.nbx * a.nby;
#ifdef BPTT_FORCE_KW
  const int kw = BPTT_FORCE_KW;
#else
  const int kw = tiles <= 256 ? 4 : tiles <= 512 ? 2 : 1;
#endif
  if (kw == 4) hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, true, true, 2, 4>), dim3(grid), dim3(1024), 0, (hipStream_t)stream, a);
  else if (kw == 2) hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, true, true, 2, 2>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((gemm_split_nt_kernel<64, 64, true, true, 2, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return unreal_launch_status();
}

int unreal_gemm_f32_split_tn(int M, int N, int K, const float* A, int lda, const float* a_absmax, const float* B, int ldb,
                             const float* b_absmax, float* C, int ldc, float* colsum, int splitk, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N) return UNREAL_EINVAL;
  if ((lda & 3) || (ldb & 3) || lda < 4 || ldb < 4 || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15)) return UNREAL_EINVAL;
  SplitTnArgs a;
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.colsum = colsum;
  a.a_absmax = a_absmax; a.b_absmax = b_absmax;
  if (SPLIT_TN_MODE == MODE_F16X2 && (!a.a_absmax || !a.b_absmax)) return UNREAL_EINVAL;
  a.ntx = (N + 127) / 128; a.nty = (M + 127) / 128;
  const int nk = (K + BK - 1) / BK;
  if (splitk < 1) splitk = 1;
  if (splitk > nk) splitk = nk;
  a.ktiles_per_split = (nk + splitk - 1) / splitk;
  a.splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
  const long grid = (long)a.ntx * a.nty * ((a.splitk + 7) / 8 * 8);
  if (grid > 0x7fffffffL) return UNREAL_EINVAL;
  if (!SPLIT_FORCE_RAGGED && K % BK == 0) hipLaunchKernelGGL(gemm_split_tn_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gemm_split_tn_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  return unreal_launch_status();
}

int unreal_split_bf16x3(int rows, int cols, const float* src, int ld_src, int transpose, int row_perm, uint16_t* dst,
                        int ld_dst, long plane_stride, void* stream) {
  if (rows <= 0 || cols <= 0 || !src || !dst || ld_src < cols) return UNREAL_EINVAL;
  const int orows = transpose ? cols : rows, ocols = transpose ? rows : cols;
  if (ld_dst < ocols || plane_stride < (long)orows * ld_dst) return UNREAL_EINVAL;
  if (row_perm != 0 && (row_perm != 1 || orows != 1024)) return UNREAL_EINVAL;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(split_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream, rows, cols, src, ld_src, transpose,
                     dst, ld_dst, plane_stride, row_perm);
  return unreal_launch_status();
}

#ifdef SPLIT_STAMPS
int exp_split_read_stamps(unsigned long long* host8, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_split_stamps), sizeof(unsigned long long) * 8);
  if (reset) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_split_stamps), z, sizeof(z));
  }
  return 0;
}
#endif

}  // extern "C"
