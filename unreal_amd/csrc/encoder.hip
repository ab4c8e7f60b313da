// Fused conv encoder for 84x84x3 uint8 frames on gfx950: conv1 8x8 s4 (3->16) + ReLU -> conv2 4x4 s2
// (16->32) + ReLU, forward and backward, as implicit GEMMs on v_mfma_f32_16x16x32_bf16 with fp32-grade error:
// every fp32 operand is split ONCE into three bf16 terms (8+8+8 mantissa bits, residuals exact) and a product tile
// accumulates the six term pairs of weight >= 2^-16 in fp32 (three where one operand is a uint8 pixel, which is
// exact in one term) -- the scheme of gemm_split.hip.
//
// Replaces tf.nn.conv2d + bias + relu of /root/reference/model/model.py:281-289,786-787 and their
// tf.gradients (train/rmsprop_applier.py:100-105).  Weights are in TF HWIO layout:
//   W1[(ky*8+kx)*3+cin][16], W2[(ky*4+kx)*16+cin][32]; outputs NHWC (flatten = model.py:331).
//
// Both kernels process ONE frame per 256-thread workgroup at a time and run TWO workgroups per CU (<= 80 KiB of LDS
// each), so that one workgroup's staging / epilogue VALU overlaps the other's MFMAs; the uint8 frame reaches LDS by
// LDS-DMA (global_load_lds, 1 KiB per wave instruction) issued a frame ahead; the frame is read from HBM exactly
// once and conv1's output never leaves the CU on the inference path.
// MFMA operand convention (16x16x32): lane l = (i = l&15, q = l>>4) supplies A[row i][k = 8q + j] and
// B[k = 8q + j][col i], j = 0..7; C/D: lane holds rows 4q..4q+3 of column i.
#include "common.h"

namespace {

constexpr int FR_LDS = 21184;            // FRAME_BYTES rounded up to 64
constexpr int FR_V = (FRAME_BYTES / 16 + 255) / 256;   // 16 B vectors per thread to move one frame (6)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- conv1 forward as EXACT-PRODUCT bf16 MFMAs -----------------------------------------------------
// The uint8 pixel is exact in bf16 (8 significant bits) and every fp32 weight is split once per kernel into
// three bf16 terms w = wh + wm + wl (8 + 8 + 8 = 24 mantissa bits, residuals computed exactly in fp32), so
// three v_mfma_f32_16x16x32_bf16 per 32-deep K chunk give products that are exact in fp32 and are
// accumulated in fp32 -- fp32-grade numerics at 16/3 of the fp32 MFMA rate.  Lane (i = l&15, q = l>>4)
// supplies the 8 patch elements k = 32kc + 8q + j of position i and of output channel i; 8 consecutive k never
// straddle a patch row (24 bytes per ky), so the pixel fragment is two aligned 32-bit LDS reads of the uint8 frame.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef u32x4 u32x4v;

__device__ __forceinline__ uint32_t bf16_rne_bits(float x) {     // fp32 -> bf16 (round to nearest even), as bits
  uint32_t u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ void split3(float w, uint32_t (&t)[3]) {
  t[0] = bf16_rne_bits(w);
  float r = w - __uint_as_float(t[0] << 16);
  t[1] = bf16_rne_bits(r);
  r = r - __uint_as_float(t[1] << 16);
  t[2] = bf16_rne_bits(r);                  // exact: at most 8 significant bits are left
}

// wb[kc][term]: 8 bf16 (k = 32kc + 8q + j) of output channel j_out, packed two per dword
__device__ __forceinline__ void load_w1_bf16x3(const float* __restrict__ W1, int q, int j_out, u32x4v (&wb)[6][3]) {
#pragma unroll
  for (int kc = 0; kc < 6; ++kc) {
    uint32_t pk[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t lo[3], hi[3];
      split3(W1[(32 * kc + 8 * q + 2 * e) * 16 + j_out], lo);
      split3(W1[(32 * kc + 8 * q + 2 * e + 1) * 16 + j_out], hi);
#pragma unroll
      for (int t = 0; t < 3; ++t) pk[t][e] = lo[t] | (hi[t] << 16);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) wb[kc][t] = (u32x4v){pk[t][0], pk[t][1], pk[t][2], pk[t][3]};
  }
}

// 8 uint8 (two dwords) -> 8 bf16: float(byte) has <= 8 significant bits, so its upper 16 bits ARE the bf16
__device__ __forceinline__ bf16x8 u8x8_to_bf16(uint32_t w0, uint32_t w1) {
  float f[8];
  f[0] = (float)(w0 & 0xffu); f[1] = (float)((w0 >> 8) & 0xffu); f[2] = (float)((w0 >> 16) & 0xffu); f[3] = (float)(w0 >> 24);
  f[4] = (float)(w1 & 0xffu); f[5] = (float)((w1 >> 8) & 0xffu); f[6] = (float)((w1 >> 16) & 0xffu); f[7] = (float)(w1 >> 24);
  u32x4v r;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    r[e] = __builtin_amdgcn_perm(__float_as_uint(f[2 * e + 1]), __float_as_uint(f[2 * e]), 0x07060302u);
  return __builtin_bit_cast(bf16x8, r);
}

// ------------------------------------------------------------------------------------------------
// Backward (its LDS helpers are shared with the forward kernel below).  Input d2 = dL/d(conv2 pre-activation) [N][81][32] (ReLU mask already applied by the
// producer), c1 = saved conv1 activation [N][400][16], the uint8 frame.  Produces dW2, dW1 (register
// accumulators across all frames of the workgroup, flushed once with float atomics), db2, db1.
//   (1) dW2[(ky,kx,c)][n] += sum_pos c1[2oy+ky][2ox+kx][c] * d2[pos][n]            M=256 N=32 K=81
//   (2) d1[2a+pa][2b+pb][c] = sum_{da,db,n} d2[a-da][b-db][n] * W2[pa+2da][pb+2db][c][n]
//       per output parity (pa,pb): M=16 channels, N=100 positions, K=128; masked by c1 > 0
//   (3) dW1[(ky,kx,cin)][c] += scale * sum_pos u8[4oy+ky][4ox+kx][cin] * d1[pos][c]  M=192 N=16 K=400
// ALL three phases run on v_mfma_f32_16x16x32_bf16 with fp32-grade error (the scheme of csrc/gemm_split.hip):
// every fp32 operand element is split ONCE into three bf16 terms (round-to-nearest, residuals exact: 8+8+8
// mantissa bits) -- c1 and d2 when the frame is staged into LDS, W2 once per kernel (its fragments stay in
// registers), d1 in the epilogue of phase (2) -- and each product tile accumulates the six term pairs of weight
// >= 2^-16 in fp32 (dropped pairs < 2^-24 |ab|); the uint8 pixel is exact in one bf16 term (3 MFMAs per tile).
// That is 6/16 of the fp32 MFMA's matrix-pipe time for phases (1)-(2).
//
// Organisation: one frame per 256-thread workgroup at a time, TWO workgroups per CU (exactly 80 KiB of LDS each) that
// run independently -- one's staging / epilogue VALU overlaps the other's MFMAs on the same SIMDs.  LDS per workgroup:
//   Y   [0, 42816): FR = the uint8 frame as it comes from HBM (LDS-DMA, issued at the top of the frame's iteration and
//       waited for only before phase (3)) | spare | Z = d2 planes [3][111 rows][32 n] bf16 with a ZERO HALO: position
//       (y,x) lives in row (y+1)*10 + (x+1), rows of y = -1, y = 9 and x = -1 are zero (x = 9 wraps onto the next
//       row's x = -1), so the tap (a-da, b-db) of output position m = 10a + b is row m + 11 - (10da + db): phase (2)
//       addresses its operands with compile-time offsets from one lane-constant base, and "outside" taps read zeros.
//       Between phases (2) and (3) the frame is expanded to bf16 [84][252] over the whole of Y (FR and Z are dead by
//       then), so phase (3) needs no conversion in its loop.
//   X   c1 planes [3][400 pos][16 ch] bf16; phase (2) overwrites them IN PLACE with the d1 planes (the ReLU mask of
//       an element is read from its own c1 hi term just before it is overwritten) + one zero row
// Reductions over POSITIONS (phases 1 and 3: the position is the row index of the LDS images) take both operands
// through ds_read_b64_tr_b16 (a 4-row x 16-column block, transposed in flight; each lane supplies the address of
// one row, so the strided conv taps need no im2col copy); phase (2) reduces over d2's channel index, contiguous in
// a row: plain 16-byte fragment reads, and its weight fragments never leave the registers.
// The NEXT frame's c1 and d2 are fetched into registers behind phase (3) and split into the planes after it.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));

#ifndef BWD_UNROLL_KS     // loop unrolling of the three phases (A/B-tested on the device: tools/exp/ablate_encoder_bwd.py)
#define BWD_UNROLL_KS 1
#endif
#ifndef BWD_UNROLL_T
#define BWD_UNROLL_T 1
#endif
#ifndef BWD_UNROLL_KC
#define BWD_UNROLL_KC 1
#endif
constexpr int C1_V = (C1_POS * 4 + 255) / 256;   // f32x4 per thread for one c1 image (7)
constexpr int D2_V = (C2_POS * 8 + 255) / 256;   // f32x4 per thread for one d2 image (3)
constexpr int XROW = 32;                         // bytes per conv1 position in a plane (16 bf16)
constexpr int XPL = C1_POS * XROW;               // 12800
constexpr int X_BYTES = 3 * XPL + 64;            // + 64 zero bytes
constexpr int ZROW = 64;                         // bytes per conv2 position in a plane (32 bf16)
constexpr int ZROWS = 111;                       // 11 x 10 halo grid + row 110 (tap (9,9) of position 99)
constexpr int ZPL = ZROWS * ZROW;                // 7104
constexpr int Z_BYTES = 3 * ZPL;                 // 21312
constexpr int Z_HALO = 30;                       // zero rows: 0..9 (y = -1), 10,20..90 (x = -1), 100..110 (y = 9)
constexpr int FR_CHUNKS = FRAME_BYTES / 16;      // 1323 16-byte pieces of a frame
constexpr int FR_DMA = (FR_CHUNKS + 63) / 64;    // 21 wave-wide LDS-DMA instructions (1 KiB each; the last one overshoots)
constexpr int Z_OFF = FR_LDS + 320;              // 21504: the DMA overshoot (21504 bytes written) stays in the spare
constexpr int X_OFF = Z_OFF + Z_BYTES;           // 42816 = size of Y
constexpr int BWD_LDS = X_OFF + X_BYTES;         // 81280: two workgroups per CU
static_assert(FR_DMA * 1024 <= Z_OFF && 2 * FRAME_BYTES <= X_OFF, "frame images must fit Y");
static_assert(2 * BWD_LDS <= 160 * 1024, "two workgroups must fit one CU's LDS");

// workgroup barrier that does NOT drain the vector-memory counter (an LDS-DMA stays in flight across it): LDS
// writes / reads of this wave are retired first, the compiler may not move memory accesses across it
#define WG_BARRIER()                                      \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory");                        \
  } while (0)

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// six term pairs of one product tile, smallest first
#define SPLIT_MMA(A, B, C)            \
  do {                                \
    C = MFMA_BF16(A[2], B[0], C);     \
    C = MFMA_BF16(A[0], B[2], C);     \
    C = MFMA_BF16(A[1], B[1], C);     \
    C = MFMA_BF16(A[1], B[0], C);     \
    C = MFMA_BF16(A[0], B[1], C);     \
    C = MFMA_BF16(A[0], B[0], C);     \
  } while (0)

// LDS bank spreading (ds_read_b64_tr_b16 serves a wave as two 32-lane halves over 64 banks = one 256-byte window):
//  * the K index of a 32-deep chunk is dealt to the lanes so that one half-wave reads 8 CONSECUTIVE positions:
//    lane (q, qq) of read h takes position 16(q>>1) + 8h + 4(q&1) + qq (both operands use the same deal);
//  * X rows (32 B) are stored at row p ^ ((p>>3)&1): 8 consecutive rows AND 8 rows two apart (the stride-2 conv2 taps)
//    then fall into 8 different 32-byte slots of the window;
//  * Z rows (64 B) swap their 32-byte halves when bit 2 of the row index is set: the 8 consecutive rows of a
//    half-wave's n-tile read fall into 8 different slots.
// Without these every transposed read of phases (1) and (3) was 2-way conflicted (PMC: 57 % of the LDS cycles).
__device__ __forceinline__ int xrow(int p) { return p ^ ((p >> 3) & 1); }
__device__ __forceinline__ int kdeal(int q, int qq) { return 16 * (q >> 1) + 4 * (q & 1) + qq; }

// 4 fp32 -> three planes of 4 bf16 (round to nearest even; x = pl0 + pl1 + pl2 exactly)
__device__ __forceinline__ void split4(const f32x4& v, u32x2v (&pl)[3]) {
  f32x2v x01 = {v[0], v[1]}, x23 = {v[2], v[3]};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const bf16x2v h01 = __builtin_convertvector(x01, bf16x2v), h23 = __builtin_convertvector(x23, bf16x2v);
    pl[t] = (u32x2v){__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
    if (t < 2) {
      x01 = x01 - __builtin_convertvector(h01, f32x2v);
      x23 = x23 - __builtin_convertvector(h23, f32x2v);
    }
  }
}

// two transposed 4-row blocks -> the 8 consecutive-k values of one 16x16x32 operand lane.  Lane 4*qq + pp of a
// 16-lane group passes the address of block row qq (+ 8*pp bytes); it receives column (lane & 15) of the 4 rows.
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* a0, const unsigned char* a1) {
  typedef s16x4v __attribute__((address_space(3))) * lds_p;
  const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
  const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a1));
  const s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void stage_c1_planes(unsigned char* xp, int tid, const f32x4 (&pc1)[C1_V]) {
#pragma unroll
  for (int c = 0; c < C1_V; ++c) {
    const int id = tid + 256 * c;
    if (id < C1_POS * 4) {
      u32x2v pl[3];
      split4(pc1[c], pl);
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x2v*>(xp + t * XPL + xrow(id >> 2) * XROW + (id & 3) * 8) = pl[t];
    }
  }
}

__device__ __forceinline__ void stage_d2_planes(unsigned char* zp, int tid, const f32x4 (&pd2)[D2_V], float (&adb2)[4]) {
  // the halo rows of the three planes are zero (the bf16 frame image of the previous frame lay over them)
  for (int e = tid; e < 3 * Z_HALO * 4; e += 256) {
    const int t = e / (Z_HALO * 4), k = (e >> 2) % Z_HALO;
    const int r = k < 10 ? k : (k < 19 ? (k - 9) * 10 : 81 + k);
    *reinterpret_cast<u32x4*>(zp + t * ZPL + r * ZROW + (e & 3) * 16) = (u32x4){0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int c = 0; c < D2_V; ++c) {
    const int id = tid + 256 * c;
    if (id < C2_POS * 8) {
      u32x2v pl[3];
      split4(pd2[c], pl);
      const int pos = id >> 3, r = (pos / 9 + 1) * 10 + pos % 9 + 1;
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x2v*>(zp + t * ZPL + r * ZROW + ((id & 7) ^ (r & 4)) * 8) = pl[t];
#pragma unroll
      for (int e = 0; e < 4; ++e) adb2[e] += pd2[c][e];
    }
  }
}

// One wave-wide LDS-DMA: 64 x 16 bytes, lane l's source -> LDS byte address lds_dst + 16 l (M0 = wave-uniform base).
// Issued from inline asm ON PURPOSE: hipcc orders every later LDS read of the same __shared__ array behind a DMA it
// can see (s_waitcnt vmcnt(0) at the first ds_read of phase (1)), which would expose the HBM latency the DMA is there
// to hide.  The kernel waits for it itself (s_waitcnt vmcnt(0) + barrier before phase (3)); no compiler-counted
// vector load is in flight while a DMA is (the prefetch loads are issued after that wait and consumed before the next
// DMA), so the compiler's own vmcnt bookkeeping stays exact.
__device__ __forceinline__ void glds16(const uint8_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// ------------------------------------------------------------------------------------------------
// Forward: conv1 8x8 s4 + ReLU -> conv2 4x4 s2 + ReLU, fused per frame, BOTH on v_mfma_f32_16x16x32_bf16 with
// fp32-grade error.  conv1 (above): the uint8 pixel is one exact bf16 term, the weights three -> 3 MFMAs per tile
// and K chunk.  conv2: c1 is split into three bf16 planes as it leaves conv1's accumulators, W2 once per kernel;
// six term-pair MFMAs per tile and K chunk (6/16 of the fp32 MFMA's matrix-pipe time, which round 1 used here).
//
// One frame per 256-thread workgroup at a time, two workgroups per CU (71 KB of LDS each), like the backward:
//   FR  the uint8 frame, by LDS-DMA (issued for frame n+1 as soon as conv1 of frame n has read FR; lands under conv2)
//   X   c1 planes [3][400 pos][16 ch] bf16 (row p stored at p ^ ((p>>3)&1), as in the backward)
//   P   partial conv2 tiles of the waves that own the upper half of K
// conv1 is evaluated TRANSPOSED (D[channel][position] = W1^T x patches): a lane then holds 4 consecutive channels of
// one position, i.e. one 16-byte store of the saved fp32 activation and one 8-byte store per bf16 plane.
// conv2: M = 81 positions (6 tiles), N = 32, K = 256 = 8 chunks of (2 taps x 16 channels).  No LDS is left for
// W2, so its fragments live in registers and the work is cut so that a wave needs few of them: wave gw owns n-tile
// gw & 1 and K half gw >> 1 (4 chunks: 48 registers of W2 fragments) for all 6 position tiles; the two halves of K
// are added in a FIXED order (lower + upper) through P, so the result does not depend on scheduling.
// ------------------------------------------------------------------------------------------------
// Operand format (round 3): ENC_FWD_F16 1 = fp16 hi + lo with one power-of-two scale per tensor, like gemm_split.hip:
// conv1 takes 2 term pairs per tile and K chunk (W1 hi / lo x the exact pixel) instead of 3, conv2 3 (c1 hi / lo x W2
// hi / lo: hh, hl, lh) instead of 6, and c1 is kept in LDS as two planes instead of three.  W1's and W2's maxima are
// reduced once per kernel; c1's scale comes from a BOUND (it must be known before the first c1 value exists):
// |c1[c]| <= 255 * frame_scale * sum_k |W1[k][c]| + |b1[c]| -- loose (by the 255 for the maze's 0 / 1 bytes), which
// shortens the range over which hi + lo carry 22 bits but keeps the absolute error under 2^-32 of the largest c1.
// ENC_FWD_F16 0 = round 2's three bf16 terms.
#ifndef ENC_FWD_F16
#define ENC_FWD_F16 1
#endif
// ENC_FWD_MAGIC 1 (fp16 form only): uint8 pixel -> fp16 with 4 byte permutes + 4 packed adds per 8-deep fragment instead
// of 8 v_cvt_f32_ubyte + 4 v_cvt_pkrtz: the halfword 0x6400 | b IS 1024 + b (ulp 1 in [1024, 2048)), and subtracting 1024
// is exact.  (Feeding the byte as an fp16 SUBNORMAL, b * 2^-24, needs the permutes alone, but the MFMA aligns its 32
// products by their exponent FIELDS: a subnormal's leading zeros are lost bits of the adder -- 5e-4 relative error on 0 / 1
// bytes, tools/exp/mfma_denorm_probe.py, profiles/r03_mfma_subnormal_probe.log.  Rejected.)
#ifndef ENC_FWD_MAGIC
#define ENC_FWD_MAGIC 1
#endif
// conv2's fragment addresses come from a per-lane table built once per kernel (24 sixteen-bit plane offsets in 12
// registers) instead of ~44 VALU of div / mod / swizzle arithmetic per position tile and frame; every wave finishes 3 of
// its n-tile's 6 position tiles (it keeps those partial sums in registers and hands the other 3 to its partner through
// P) instead of the lower-K waves finishing all 6 while the upper-K waves idle at the next barrier; conv2's fragment
// reads run ENC_FWD_PF steps ahead of the MFMAs that consume them.  (Round 3, tools/exp/fwd_ab.py: 1.45 -> 1.21 ms per
// 81,920 frames, outputs bit-identical.)
#ifndef ENC_FWD_PF
#define ENC_FWD_PF 2
#endif
// ENC_FWD_FR2 1: two uint8 frame buffers, the LDS-DMA runs two frames ahead (80,896 B of LDS: still two workgroups per CU).
// Measured neutral (tools/exp/fwd_ab.py, profiles/r03_encoder_fwd_ab.log: 1.20 vs 1.21 ms; with the magic conversion 1.15 vs
// 1.12): the wait before [F2] is not the DMA -- vmcnt also counts the 25.6 KB of conv1 activations this wave has stored, and
// at 4.2 TB/s of HBM traffic (2.6 write + 1.6 read) those drain slowly.  Off.
#ifndef ENC_FWD_FR2
#define ENC_FWD_FR2 0
#endif
constexpr int NPLF = ENC_FWD_F16 ? 2 : 3;
typedef _Float16 fh8 __attribute__((ext_vector_type(8)));
typedef _Float16 fh2 __attribute__((ext_vector_type(2)));
#if ENC_FWD_F16
typedef fh8 fop8;
#define MFMA_FOP(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define SPLIT_MMA_FOP(A, B, C)       \
  do {                               \
    C = MFMA_FOP(A[1], B[0], C);     \
    C = MFMA_FOP(A[0], B[1], C);     \
    C = MFMA_FOP(A[0], B[0], C);     \
  } while (0)
#else
typedef bf16x8 fop8;
#define MFMA_FOP(a, b, c) MFMA_BF16(a, b, c)
#define SPLIT_MMA_FOP(A, B, C) SPLIT_MMA(A, B, C)
#endif
// 4 fp32 -> NPLF planes of 4 sixteen-bit terms
__device__ __forceinline__ void split4_fop(const f32x4& v, float scale, u32x2v (&pl)[3]) {
#if ENC_FWD_F16
  const f32x2v x01 = (f32x2v){v[0], v[1]} * scale, x23 = (f32x2v){v[2], v[3]} * scale;
  const fh2 h01 = __builtin_convertvector(x01, fh2), h23 = __builtin_convertvector(x23, fh2);
  const fh2 l01 = __builtin_convertvector(x01 - __builtin_convertvector(h01, f32x2v), fh2);
  const fh2 l23 = __builtin_convertvector(x23 - __builtin_convertvector(h23, f32x2v), fh2);
  pl[0] = (u32x2v){__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
  pl[1] = (u32x2v){__builtin_bit_cast(unsigned int, l01), __builtin_bit_cast(unsigned int, l23)};
  pl[2] = pl[1];
#else
  (void)scale;
  split4(v, pl);
#endif
}
// 8 uint8 (two dwords) -> 8 sixteen-bit floats (exact in either format)
__device__ __forceinline__ fop8 u8x8_to_fop(uint32_t w0, uint32_t w1) {
#if ENC_FWD_F16 && ENC_FWD_MAGIC
  u32x4v r;        // halfword j = 0x6400 | byte j (selector 4 = byte 0 of the constant) = fp16(1024 + b)
  r[0] = __builtin_amdgcn_perm(0x64646464u, w0, 0x04010400u);
  r[1] = __builtin_amdgcn_perm(0x64646464u, w0, 0x04030402u);
  r[2] = __builtin_amdgcn_perm(0x64646464u, w1, 0x04010400u);
  r[3] = __builtin_amdgcn_perm(0x64646464u, w1, 0x04030402u);
  const fh8 k1024 = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};
  return __builtin_bit_cast(fop8, r) - k1024;
#elif ENC_FWD_F16
  float f[8];
  f[0] = (float)(w0 & 0xffu); f[1] = (float)((w0 >> 8) & 0xffu); f[2] = (float)((w0 >> 16) & 0xffu); f[3] = (float)(w0 >> 24);
  f[4] = (float)(w1 & 0xffu); f[5] = (float)((w1 >> 8) & 0xffu); f[6] = (float)((w1 >> 16) & 0xffu); f[7] = (float)(w1 >> 24);
  u32x4v r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(f[2 * e], f[2 * e + 1]));   // exact: integers
  return __builtin_bit_cast(fop8, r);
#else
  return u8x8_to_bf16(w0, w1);
#endif
}

constexpr int FWD_FR = FR_DMA * 1024;            // 21504: uint8 frame + DMA overshoot
constexpr int FWD_X = (1 + ENC_FWD_FR2) * FWD_FR;   // two frame buffers: the DMA runs two frames ahead
constexpr int FWD_P = FWD_X + NPLF * XPL;
constexpr int FWD_LDS = FWD_P + 2 * 6 * 1024;    // 80896 (fp16x2, two frame buffers)
static_assert(2 * FWD_LDS <= 160 * 1024, "two workgroups must fit one CU's LDS");

// conv1 for TWO (or one) 16-position tiles, transposed: acc[r] = channel 4q + r at position 16 t + i
template <bool TWO>
__device__ __forceinline__ void conv1_tiles(const uint8_t* fr, unsigned char* xp, float* __restrict__ c1_out,
                                            const u32x4v (&w1)[6][3], const int (&koff)[6], const f32x4& bias, float scale,
                                            int ta, int tb, int i, int q, float& c1_max, float c1_scale) {
  const int pa = ta * 16 + i, pb = tb * 16 + i;
  const uint8_t* fa = fr + (4 * (pa / 20)) * FRAME_ROW_BYTES + 12 * (pa % 20);
  const uint8_t* fb = fr + (4 * (pb / 20)) * FRAME_ROW_BYTES + 12 * (pb % 20);
  f32x4 acca = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < 6; ++kc) {
    const uint32_t* pa32 = reinterpret_cast<const uint32_t*>(fa + koff[kc]);
    const fop8 xa = u8x8_to_fop(pa32[0], pa32[1]);
    fop8 xb;
    if (TWO) {
      const uint32_t* pb32 = reinterpret_cast<const uint32_t*>(fb + koff[kc]);
      xb = u8x8_to_fop(pb32[0], pb32[1]);
    }
#pragma unroll
    for (int t = NPLF - 1; t >= 0; --t) {            // smallest weight term first
      const fop8 w = __builtin_bit_cast(fop8, w1[kc][t]);
      acca = MFMA_FOP(w, xa, acca);
      if (TWO) accb = MFMA_FOP(w, xb, accb);
    }
  }
#pragma unroll
  for (int h = 0; h < (TWO ? 2 : 1); ++h) {
    const f32x4& acc = h ? accb : acca;
    const int pos = h ? pb : pa;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(scale * acc[r] + bias[r], 0.f);
    if (c1_out) *reinterpret_cast<f32x4*>(c1_out + pos * C1_CH + 4 * q) = v;
    c1_max = fmaxf(fmaxf(c1_max, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    u32x2v pl[3];
    split4_fop(v, c1_scale, pl);
#pragma unroll
    for (int u = 0; u < NPLF; ++u) *reinterpret_cast<u32x2v*>(xp + u * XPL + xrow(pos) * XROW + 8 * q) = pl[u];
  }
}

#ifdef ENC_FWD_STAMPS   // tools/exp/fwd_ab.py only: where a wave's cycles go (workgroup 3, s_memtime ticks summed over its frames)
__device__ unsigned long long g_fstamp[4][8];
#define FSTAMP(k)                                                  \
  do {                                                             \
    if (blockIdx.x == 3 && lane == 0) {                            \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
      g_fstamp[gw][k] += t_ - t_prev_;                             \
      t_prev_ = t_;                                                \
    }                                                              \
  } while (0)
#else
#define FSTAMP(k)
#endif

// ---- the weights' share of the forward prologue: scales and operand fragments.  Every workgroup of every launch used to
// redo it (two reductions over W1 / W2, 80 strided loads and their splits per thread: ~6 of the ~18 us a launch costs before
// its first frame); unreal_encoder_prepare runs it ONCE per weight update into a 45 KB block that the launches read back
// with 20 coalesced 16-byte loads per thread.  Same device functions on both routes: identical registers, identical outputs.
struct EncFwdScales { float S_W1, S_W2, S_C1; };
constexpr int ENC_PREP_W1 = 1;                                  // u32x4 index: [kc][t][lane]
constexpr int ENC_PREP_W2 = ENC_PREP_W1 + 6 * NPLF * 64;        // [c][t][thread]
constexpr int ENC_PREP_VECS = ENC_PREP_W2 + 4 * NPLF * 256;     // x 16 bytes

// red: 96 floats of LDS; all 256 threads; three barriers
__device__ __forceinline__ EncFwdScales enc_fwd_scales(const float* __restrict__ W1, const float* __restrict__ b1,
                                                       const float* __restrict__ W2, float scale, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, gw = tid >> 6;
  // red: [16] max |W1|, [17] max |W2|, [32..95] the waves' partial L1 norms of W1 per channel
  if (tid < 18) red[tid] = 0.f;
  __syncthreads();
  float l1 = 0.f, m1 = 0.f, m2 = 0.f;
  const int c = tid & 15;
  for (int k = tid >> 4; k < 192; k += 16) { const float w = fabsf(W1[k * 16 + c]); l1 += w; m1 = fmaxf(m1, w); }
  for (int e = tid; e < 8192; e += 256) m2 = fmaxf(m2, fabsf(W2[e]));
  // channel L1 norms in a FIXED order (a float atomicAdd's order is not: workgroups could then land on different sides
  // of a power of two and round their c1 planes differently): the wave's four lanes of channel c by a shuffle tree, the
  // four waves' partials by a fixed sum below.  (The maxima are order-independent: atomicMax.)
  l1 += __shfl_xor(l1, 16, 64);
  l1 += __shfl_xor(l1, 32, 64);
  if (lane < 16) red[32 + gw * 16 + lane] = l1;
  atomicMax(reinterpret_cast<unsigned int*>(red + 16), __float_as_uint(m1));
  atomicMax(reinterpret_cast<unsigned int*>(red + 17), __float_as_uint(m2));
  __syncthreads();
  float bound = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float l1c = (red[32 + k] + red[48 + k]) + (red[64 + k] + red[80 + k]);
    bound = fmaxf(bound, 255.f * fabsf(scale) * l1c + fabsf(b1[k]));
  }
  const EncFwdScales sc = {pow2_scale(red[16]), pow2_scale(red[17]), pow2_scale(bound)};
  __syncthreads();
  return sc;
}

// conv1: A[row = channel i][k = 32kc + 8q + j], NPLF terms
__device__ __forceinline__ void enc_fwd_w1_frags(const float* __restrict__ W1, float S_W1, int q, int i, u32x4v (&w1)[6][3]) {
#pragma unroll
  for (int kc = 0; kc < 6; ++kc) {
    f32x4 lo4, hi4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo4[e] = W1[(32 * kc + 8 * q + e) * 16 + i];
      hi4[e] = W1[(32 * kc + 8 * q + 4 + e) * 16 + i];
    }
    u32x2v lo[3], hi[3];
    split4_fop(lo4, S_W1, lo);
    split4_fop(hi4, S_W1, hi);
#pragma unroll
    for (int t = 0; t < NPLF; ++t) w1[kc][t] = (u32x4v){lo[t][0], lo[t][1], hi[t][0], hi[t][1]};
  }
}

// conv2: B[k = 32kc + 8q + j][col = n = 16nt + i] = W2[(tap = 2kc + (q>>1)) * 16 + 8(q&1) + j][n], kc = 4kh + c
__device__ __forceinline__ void enc_fwd_w2_frags(const float* __restrict__ W2, float S_W2, int q, int i, int nt, int kh,
                                                 fop8 (&w2)[4][NPLF]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k0 = 32 * (4 * kh + c) + 8 * q;
    f32x4 lo4, hi4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo4[j] = W2[(k0 + j) * 32 + 16 * nt + i];
      hi4[j] = W2[(k0 + 4 + j) * 32 + 16 * nt + i];
    }
    u32x2v lo[3], hi[3];
    split4_fop(lo4, S_W2, lo);
    split4_fop(hi4, S_W2, hi);
#pragma unroll
    for (int t = 0; t < NPLF; ++t) {
      const u32x4 w4 = {lo[t][0], lo[t][1], hi[t][0], hi[t][1]};
      w2[c][t] = __builtin_bit_cast(fop8, w4);
    }
  }
}

// one workgroup: the prepared block of encoder_fwd_kernel (header: S_W1, S_W2, S_C1, the frame scale it was made for)
__global__ __launch_bounds__(256) void encoder_prepare_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                              const float* __restrict__ W2, float scale, u32x4* __restrict__ prep) {
  __shared__ float red[96];
  const int tid = threadIdx.x, lane = tid & 63, gw = tid >> 6;
  const int i = lane & 15, q = lane >> 4, nt = gw & 1, kh = gw >> 1;
  const EncFwdScales sc = enc_fwd_scales(W1, b1, W2, scale, red);
  if (tid == 0) prep[0] = (u32x4){__float_as_uint(sc.S_W1), __float_as_uint(sc.S_W2), __float_as_uint(sc.S_C1), __float_as_uint(scale)};
  u32x4v w1[6][3];
  enc_fwd_w1_frags(W1, sc.S_W1, q, i, w1);
  if (gw == 0) {
#pragma unroll
    for (int kc = 0; kc < 6; ++kc)
#pragma unroll
      for (int t = 0; t < NPLF; ++t) prep[ENC_PREP_W1 + (kc * NPLF + t) * 64 + lane] = w1[kc][t];
  }
  fop8 w2[4][NPLF];
  enc_fwd_w2_frags(W2, sc.S_W2, q, i, nt, kh, w2);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < NPLF; ++t) prep[ENC_PREP_W2 + (c * NPLF + t) * 256 + tid] = __builtin_bit_cast(u32x4, w2[c][t]);
}

template <bool BITS>
__global__ __launch_bounds__(256, 2) void encoder_fwd_kernel(int N, const uint8_t* __restrict__ frames,
                                                             const int* __restrict__ frame_idx, float scale,
                                                             const float* __restrict__ W1, const float* __restrict__ b1,
                                                             const float* __restrict__ W2, const float* __restrict__ b2,
                                                             float* __restrict__ c1_out, float* __restrict__ f2_out,
                                                             uint16_t* __restrict__ relu_bits, float* f2_absmax,
                                                             float* c1_absmax, const u32x4* __restrict__ prep) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[FWD_LDS];
  float f2_max = 0.f, c1_max = 0.f;            // max of the outputs this lane has stored (they are >= 0)
  const int tid = threadIdx.x, lane = tid & 63, gw = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  uint8_t* fr = smem;
  unsigned char* xp = smem + FWD_X;
  unsigned char* pp = smem + FWD_P;
  const unsigned lds_fr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const int nt = gw & 1, kh = gw >> 1;         // conv2: this wave's n-tile and K half

  // fp16x2: power-of-two scales of W1, W2 and of the c1 planes (from the bound above) -- read from the prepared block
  // (unreal_encoder_prepare: once per weight update) or reduced here, once per workgroup
  float S_W1 = 1.f, S_W2 = 1.f, S_C1 = 1.f;
  if (ENC_FWD_F16) {
    if (prep) {
      const u32x4 h = prep[0];
      S_W1 = __uint_as_float(h[0]); S_W2 = __uint_as_float(h[1]); S_C1 = __uint_as_float(h[2]);
    } else {
      const EncFwdScales sc = enc_fwd_scales(W1, b1, W2, scale, reinterpret_cast<float*>(smem));
      S_W1 = sc.S_W1; S_W2 = sc.S_W2; S_C1 = sc.S_C1;
    }
  }
  // conv1: un-scales W1 and applies the byte scale in one factor
  const float scale1 = scale * pow2_inv(S_W1);
  const float inv_c2 = pow2_inv(S_C1) * pow2_inv(S_W2);    // conv2: exact (both powers of two; |exponents| <= 100 each
                                                           // cannot meet here: c1's bound and W2's maximum are O(1))
  u32x4v w1[6][3];                             // conv1: A[row = channel i][k = 32kc + 8q + j], NPLF terms
  if (ENC_FWD_F16) {
    if (prep) {
#pragma unroll
      for (int kc = 0; kc < 6; ++kc)
#pragma unroll
        for (int t = 0; t < NPLF; ++t) w1[kc][t] = prep[ENC_PREP_W1 + (kc * NPLF + t) * 64 + lane];
    } else {
      enc_fwd_w1_frags(W1, S_W1, q, i, w1);
    }
  } else {
    load_w1_bf16x3(W1, q, i, w1);
  }
  int koff[6];                                 // byte offset of patch element k = 32kc + 8q inside the frame
#pragma unroll
  for (int kc = 0; kc < 6; ++kc) koff[kc] = ((32 * kc + 8 * q) / 24) * FRAME_ROW_BYTES + (32 * kc + 8 * q) % 24;
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1 + 4 * q);
  // conv2: B[k = 32kc + 8q + j][col = n = 16nt + i] = W2[(tap = 2kc + (q>>1)) * 16 + 8(q&1) + j][n], kc = 4kh + c
  fop8 w2[4][NPLF];
  if (ENC_FWD_F16 && prep) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < NPLF; ++t) w2[c][t] = __builtin_bit_cast(fop8, prep[ENC_PREP_W2 + (c * NPLF + t) * 256 + tid]);
  } else {
    enc_fwd_w2_frags(W2, S_W2, q, i, nt, kh, w2);
  }
  const float bias2 = b2[16 * nt + i];
  // conv2 fragment of (position tile mt, K chunk c): byte offset inside a c1 plane of row (2oy + dy)*20 + 2ox + dx, channel
  // half q & 1, for this lane's position 16mt + i and tap 2(4kh + c) + (q >> 1) = (dy, dx); two offsets per register
  unsigned c2tab[12];
#pragma unroll
  for (int e2 = 0; e2 < 12; ++e2) {
    unsigned pk = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int u = (2 * e2 + h) >> 2, c = (2 * e2 + h) & 3;
      const int mt = (u + 3 * kh) % 6;                      // a wave walks its OWN three tiles first
      const int pos = min(16 * mt + i, C2_POS - 1);
      const int p1 = (2 * (pos / 9)) * 20 + 2 * (pos % 9);
      const int tap = 2 * (4 * kh + c) + (q >> 1);
      pk |= (unsigned)(xrow(p1 + (tap >> 2) * 20 + (tap & 3)) * XROW + 16 * (q & 1)) << (16 * h);
    }
    c2tab[e2] = pk;
  }

  // uint8 frame (pool index fidx) -> FR buffer `buf` (lane-linear 1 KiB pieces; wave gw issues pieces gw, gw + 4, ...)
  auto dma_frame = [&](int fidx, int buf) {
    const uint8_t* src = frames + (size_t)fidx * FRAME_BYTES;
    for (int kk = gw; kk < FR_DMA; kk += 4) {
      const int chunk = min(64 * kk + lane, FR_CHUNKS - 1);
      glds16(src + 16 * chunk, __builtin_amdgcn_readfirstlane(lds_fr + buf * FWD_FR + 1024 * kk));
    }
  };
  const int stride = gridDim.x;
  dma_frame(frame_idx[blockIdx.x], 0);         // the launch guarantees gridDim.x <= N
  if (ENC_FWD_FR2 && blockIdx.x + stride < N) dma_frame(frame_idx[blockIdx.x + stride], 1);
  // index of the frame whose DMA is issued behind the next [F1]: two frames ahead (FR2) / one
  int fidx_next = blockIdx.x + (1 + ENC_FWD_FR2) * stride < N ? frame_idx[blockIdx.x + (1 + ENC_FWD_FR2) * stride] : 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WG_BARRIER();

  // Two barriers per frame.  [F1] conv1 done: X complete, the frame buffer conv1 read is dead -> the DMA of the frame TWO
  // ahead starts into it (FR2: two frame buffers; with one buffer the next frame's DMA had only conv2 -- 3,600 ticks --
  // to land and every wave waited ~1,400 ticks for it).  [F2] conv2 done: P complete, X dead; every wave has first
  // waited until at most its own pieces of the newest DMA are outstanding (loads return in order: the pieces of the
  // frame conv1 reads next were issued a whole frame earlier).  Then every wave finishes its three output tiles and
  // goes on to conv1 of the next frame; P and X are rewritten only behind the next [F1] / [F2].
#ifdef ENC_FWD_STAMPS
  unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
  int buf = 0;
  for (int n = blockIdx.x; n < N; n += stride) {
    int zero;                                  // opaque 0, new every frame: keeps conv1's address sets out of the registers
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    FSTAMP(0);
    {
      float* c1n = c1_out ? c1_out + (size_t)n * (C1_POS * C1_CH) : nullptr;
      const uint8_t* frn = fr + buf * FWD_FR;
      // 25 tiles over 4 waves = 7 + 6 + 6 + 6
      for (int tt = (gw + 2) & 3; tt < 25; tt += 8) {
        if (tt + 4 < 25) conv1_tiles<true>(frn, xp, c1n, w1, koff, bias1, scale1, tt, tt + 4, i + zero, q, c1_max, S_C1);
        else conv1_tiles<false>(frn, xp, c1n, w1, koff, bias1, scale1, tt, tt, i + zero, q, c1_max, S_C1);
      }
    }
    FSTAMP(1);
    WG_BARRIER();     // [F1] c1 planes complete; this frame's buffer is dead
    FSTAMP(2);
    const bool dma_now = n + (1 + ENC_FWD_FR2) * stride < N;
    if (dma_now) dma_frame(fidx_next, ENC_FWD_FR2 ? buf : 0);
    fidx_next = n + (2 + ENC_FWD_FR2) * stride < N ? frame_idx[n + (2 + ENC_FWD_FR2) * stride] : 0;
    // conv2: this wave's n-tile and K half (4 chunks) of all 6 position tiles -- 24 steps s = 4u + c (u: tile in this
    // wave's order, own tiles first; c: K chunk), fragments ENC_FWD_PF steps ahead
    f32x4 acc[3];
    {
      fop8 af[ENC_FWD_PF + 1][NPLF];
      auto frag = [&](int st, fop8 (&dst)[NPLF]) {
        const unsigned off = (st & 1) ? (c2tab[st >> 1] >> 16) : (c2tab[st >> 1] & 0xffffu);
#pragma unroll
        for (int t = 0; t < NPLF; ++t) dst[t] = *reinterpret_cast<const fop8*>(xp + off + t * XPL);
      };
#pragma unroll
      for (int st = 0; st < ENC_FWD_PF; ++st) frag(st, af[st]);
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 24; ++st) {
        const int u = st >> 2, c = st & 3;
        if (st + ENC_FWD_PF < 24) frag(st + ENC_FWD_PF, af[(st + ENC_FWD_PF) % (ENC_FWD_PF + 1)]);
        SPLIT_MMA_FOP(af[st % (ENC_FWD_PF + 1)], w2[c], a);
        // program order inside the step: { MFMA, VALU, LDS read } x 3 -- the fragment requests of step st + PF go out in
        // the shadow of this step's MFMAs -- and nothing crosses the step boundary (left alone, the scheduler sinks every
        // read to just before its MFMA to save registers: read, lgkmcnt(0), MFMA, ...)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c == 3) {
          if (ENC_FWD_F16) a *= inv_c2;            // back to c1 * W2 units before the two K halves meet
          if (u < 3) acc[u] = a;                   // own tile 3kh + u
          else *reinterpret_cast<f32x4*>(pp + ((nt * 6 + (u - 3 * kh)) * 64 + lane) * 16) = a;   // partner's tile u - 3kh
          a = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    FSTAMP(3);
    // the frame conv1 reads next has landed: everything but this wave's pieces of the DMA just issued (6 for wave 0, 5 for
    // the others; vector loads return in order, and no other vector load is in flight) must have returned
    if (ENC_FWD_FR2 && dma_now) {
      if (gw == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FSTAMP(4);
    WG_BARRIER();     // [F2]
    FSTAMP(5);
    {
      // this wave's three tiles mt = 3kh + u: lower + upper K half (the sum of two floats does not depend on which wave
      // held which), bias, ReLU; tiles 0..4 are whole, tile 5 holds position 80 alone (q = 0, r = 0)
      f32x4 part[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) part[u] = *reinterpret_cast<const f32x4*>(pp + ((nt * 6 + 3 * kh + u) * 64 + lane) * 16);
      float* dst = f2_out + (size_t)n * F2_DIM + (48 * kh + 4 * q) * C2_CH + 16 * nt + i;
      // ReLU pattern: ballot bit 16q + i of step r = (position 4q + r of the tile, channel 16nt + i); word (pos, nt) of the
      // frame's bit matrix.  Lane i < 4 of every q-group stores the word of r = i: one store per tile
      uint16_t* bits_q = BITS ? relu_bits + ((size_t)n * C2_POS + 48 * kh + 4 * q + (i & 3)) * 2 + nt : nullptr;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        f32x4 v;
        unsigned long long m[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaxf((acc[u][r] + part[u][r]) + bias2, 0.f);
          if (BITS) m[r] = __ballot(v[r] > 0.f);
        }
        if (u < 2) {                           // tiles 0, 1, 3, 4: whole
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(16 * u + r) * C2_CH] = v[r];
          f2_max = fmaxf(fmaxf(f2_max, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
          if (BITS) {
            const unsigned long long mi = (i & 2) ? ((i & 1) ? m[3] : m[2]) : ((i & 1) ? m[1] : m[0]);
            if (i < 4) bits_q[(16 * u) * 2] = (uint16_t)(mi >> (16 * q));
          }
        } else {                               // tile 2 (whole) or tile 5 (position 80 only)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = !kh || (q | r) == 0;
            if (ok) dst[(32 + r) * C2_CH] = v[r];
            f2_max = fmaxf(f2_max, ok ? v[r] : 0.f);
          }
          if (BITS) {
            const unsigned long long mi = (i & 2) ? ((i & 1) ? m[3] : m[2]) : ((i & 1) ? m[1] : m[0]);
            if (i < 4 && (!kh || (q | i) == 0)) bits_q[32 * 2] = (uint16_t)(mi >> (16 * q));
          }
        }
      }
    }
    FSTAMP(6);
    buf ^= ENC_FWD_FR2;
  }
  // one commit per workgroup and slot (the frame buffer is dead: no DMA was issued behind the last frame's [F1], and the
  // waves still in their epilogue only read P).  The 2,048 waves of a launch end within microseconds of each other: each
  // would find the slots at their old values and issue its own serialised atomics
  {
    float* red = reinterpret_cast<float*>(smem);
    f2_max = wave_max(f2_max);
    c1_max = wave_max(c1_max);
    if (lane == 0) { red[gw] = f2_max; red[4 + gw] = c1_max; }
    __syncthreads();
    if (gw == 0) {
      absmax_commit(f2_absmax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));   // the A scale of the fc GEMM that reads f2
      absmax_commit(c1_absmax, fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));   // the scale of the c1 planes in unreal_encoder_bwd
    }
  }
}

#ifdef UNREAL_EXP_KERNELS     // tools/exp/encoder_ablate.hip only: experiment kernels that need this file's helpers
#include UNREAL_EXP_KERNELS
#endif

#include "encoder_bwd_roles.h"     // round 3: the role-specialised backward kernel (uses the helpers above)

}  // namespace

extern "C" {

int unreal_encoder_fwd(int N, const uint8_t* frames, const int* frame_idx, float frame_scale, const float* W1,
                       const float* b1, const float* W2, const float* b2, float* c1_out, float* f2_out,
                       uint16_t* relu_bits, float* f2_absmax, float* c1_absmax, const void* prepared, void* stream) {
  if (N <= 0 || !frames || !frame_idx || !W1 || !b1 || !W2 || !b2 || !f2_out) return UNREAL_EINVAL;
  if ((((uintptr_t)prepared) & 15) || (prepared && !ENC_FWD_F16)) return UNREAL_EINVAL;
  const u32x4* prep = static_cast<const u32x4*>(prepared);
  int blocks = min(N, 512);             // one frame per workgroup at a time, two workgroups per CU
  if (relu_bits)
    hipLaunchKernelGGL(encoder_fwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, N, frames, frame_idx,
                       frame_scale, W1, b1, W2, b2, c1_out, f2_out, relu_bits, f2_absmax, c1_absmax, prep);
  else
    hipLaunchKernelGGL(encoder_fwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, N, frames, frame_idx,
                       frame_scale, W1, b1, W2, b2, c1_out, f2_out, relu_bits, f2_absmax, c1_absmax, prep);
  return unreal_launch_status();
}

int unreal_encoder_prepare(const float* W1, const float* b1, const float* W2, float frame_scale, void* prepared,
                           long prepared_bytes, void* stream) {
  if (!W1 || !b1 || !W2 || !prepared || (((uintptr_t)prepared) & 15) || !ENC_FWD_F16) return UNREAL_EINVAL;
  if (prepared_bytes < (long)ENC_PREP_VECS * 16) return UNREAL_EINVAL;
  hipLaunchKernelGGL(encoder_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, W1, b1, W2, frame_scale,
                     static_cast<u32x4*>(prepared));
  return unreal_launch_status();
}

int unreal_encoder_bwd(int N, const uint8_t* frames, const int* frame_idx, float frame_scale, const float* W2,
                       const float* c1_saved, const float* c1_absmax, const float* d2, const float* d2_absmax, float* dW1,
                       float* db1, float* dW2, float* db2, void* stream) {
  if (N <= 0 || !frames || !frame_idx || !W2 || !c1_saved || !d2 || !dW1 || !db1 || !dW2 || !db2)
    return UNREAL_EINVAL;
  if (ENC_BWD_F16 && (!c1_absmax || !d2_absmax)) return UNREAL_EINVAL;
  int blocks = min(N, 256);             // one 512-thread workgroup per CU (4 consumer + 4 producer waves), a frame at a time
  hipLaunchKernelGGL((encoder_bwd_roles_kernel<7, true>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, N, frames,
                     frame_idx, frame_scale, W2, c1_saved, d2, dW1, db1, dW2, db2, c1_absmax, d2_absmax);
  return unreal_launch_status();
}

#ifdef ENC_FWD_STAMPS   // tools/exp only
int exp_read_fstamps(unsigned long long* host32, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_fstamp), sizeof(unsigned long long) * 32);
  if (reset) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fstamp), z, sizeof(z));
  }
  return 0;
}
#endif
#ifdef UNREAL_EXP_ENTRIES     // tools/exp/encoder_ablate.hip only
#include UNREAL_EXP_ENTRIES
#endif

}  // extern "C"
