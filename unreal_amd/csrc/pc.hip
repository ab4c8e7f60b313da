// Pixel-control head on gfx950: the two stride-2 VALID 4x4 transposed convolutions (value + advantage)
// of the dueling Q head, the dueling combine, Q-max, the pixel-control loss and its backward pass.
//
// Reference: /root/reference/model/model.py:411-443 (_pc_deconv_layers), 805-820 (_deconv2d =
// tf.nn.conv2d_transpose, filter [kh,kw,out_c,in_c]), 542-557 (_pc_loss = lambda * l2_loss(pc_r - Q[a])).
//
// The transposed conv is evaluated per OUTPUT PARITY as a small dense GEMM (no scatter, no atomics):
//   out[2a+pa][2b+pb][co] = sum_{da,db in {0,1}} sum_ci in[a-da][b-db][ci] * W[pa+2da][pb+2db][co][ci]
// Forward: the four parities read the SAME input rows, so they are packed side by side into the MFMA's N dimension:
// ONE GEMM of M = 100 base positions (a,b), K = 4 taps x 32 channels, N = 4 parities x (1+A) channels = 20 of 32
// columns (per-parity GEMMs would fill 5 of 16).  Operands are split into fp16 hi + lo planes when they are staged into
// LDS (hp per frame, the weights once per workgroup; round 2: three bf16 terms) under one power-of-two scale per tensor --
// hp's from the absmax slot the pc_fc1 GEMM commits, the weights' reduced here, d_dec's from the slot the forward kernel
// commits -- and multiplied as three term-pair v_mfma_f32_16x16x32_f16 (hh, hl, lh; round 2: six), the scheme and error
// level of csrc/gemm_split.hip.  The backward pass uses the same planes; its weight gradient reduces over positions (the
// row index of the LDS images), so its operands come through transposed LDS reads.
// Same group/LDS organisation and MFMA operand convention as encoder.hip: the next frame's inputs are
// fetched into registers behind the current frame's math, outputs leave through LDS in 16 B/lane rows.
#include "common.h"

namespace {

constexpr int HP_LD = 36, HP_ROWS = 84;   // [81][32] + zero rows; row 81 = padding row
constexpr int NPLP = 2;                   // fp16 hi + lo planes
constexpr int WD_ELEMS = 4 * 4 * 2 * 4 * 16 * 4;   // 8192
constexpr int HP_V = (C2_POS * 8 + 255) / 256;     // f32x4 per thread for one [81][32] image (3)
constexpr int DD_V = (PC_CELLS * 8 / 4 + 255) / 256;   // f32x4 per thread for one [400][<=8] image (4)

struct PcFwdArgs {
  int N, A;
  const float* hp;          // [N][2592] relu(pc_fc1)
  const float* hp_absmax;   // absmax slot covering hp (common.h): the scale of its fp16 planes
  const float* Wv; const float* bv; const float* Wa; const float* ba;
  // bootstrap mode
  float* qmax;              // [N][400] or null
  // training mode
  const int* action;        // [N]
  const float* target;      // [N][400]
  const int* mask;          // [N]
  float lambda, grad_scale;
  float* d_dec;             // [N][400][1+A]
  float* ddec_absmax;       // nullable slot: max |d_dec|, the scale of its planes in the backward kernel
  float* loss;              // scalar accumulator
};

__device__ __forceinline__ float deconv_w(const float* __restrict__ Wv, const float* __restrict__ Wa, int A, int ky, int kx,
                                          int co, int ci) {
  if (co == 0) return Wv[(ky * 4 + kx) * 32 + ci];
  if (co <= A) return Wa[((ky * 4 + kx) * A + (co - 1)) * 32 + ci];
  return 0.f;
}

__device__ __forceinline__ void hp_load(const float* __restrict__ src, int gtid, f32x4 (&r)[HP_V]) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
#pragma unroll
  for (int c = 0; c < HP_V; ++c) {
    int id = gtid + 256 * c;
    r[c] = s4[id < C2_POS * 8 ? id : gtid];
  }
}

__device__ __forceinline__ void hp_store(float* hp, int gtid, const f32x4 (&r)[HP_V]) {
#pragma unroll
  for (int c = 0; c < HP_V; ++c) {
    int id = gtid + 256 * c;
    if (id < C2_POS * 8) *reinterpret_cast<f32x4*>(hp + (id >> 3) * HP_LD + (id & 7) * 4) = r[c];
  }
}

typedef _Float16 fh8p __attribute__((ext_vector_type(8)));
typedef _Float16 fh2p __attribute__((ext_vector_type(2)));
typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2p __attribute__((ext_vector_type(2)));

constexpr int HPP_ROW = 80;                        // bytes per hp row in a plane: 32 ci + 16 B pad
constexpr int HPP_PLANE = HP_ROWS * HPP_ROW;       // 6720
constexpr int WDP_ROW = 80;                        // bytes per (tap, column) weight row: 32 ci + pad
constexpr int WDP_PLANE = 4 * 32 * WDP_ROW;        // [dd(4)][n(32)] rows = 10240

// 4 fp32 -> hi / lo planes of 4 fp16 each: x * scale = hi + lo (round to nearest; 22 significant bits + lo's sign)
__device__ __forceinline__ void split4p(const f32x4& v, float scale, u32x2p (&pl)[NPLP]) {
  const f32x2p x01 = (f32x2p){v[0], v[1]} * scale, x23 = (f32x2p){v[2], v[3]} * scale;
  const fh2p h01 = __builtin_convertvector(x01, fh2p), h23 = __builtin_convertvector(x23, fh2p);
  const fh2p l01 = __builtin_convertvector(x01 - __builtin_convertvector(h01, f32x2p), fh2p);
  const fh2p l23 = __builtin_convertvector(x23 - __builtin_convertvector(h23, f32x2p), fh2p);
  pl[0] = (u32x2p){__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
  pl[1] = (u32x2p){__builtin_bit_cast(unsigned int, l01), __builtin_bit_cast(unsigned int, l23)};
}

// hp [81][32] fp32 (registers) -> hi / lo planes in LDS
__device__ __forceinline__ void hp_store_planes(unsigned char* hpp, int gtid, const f32x4 (&r)[HP_V], float scale) {
#pragma unroll
  for (int c = 0; c < HP_V; ++c) {
    const int id = gtid + 256 * c;
    if (id < C2_POS * 8) {
      u32x2p pl[NPLP];
      split4p(r[c], scale, pl);
      // hp = relu(...) >= 0: the backward kernel's ReLU mask reads "hi != 0" (common.h: keep_positive_visible)
      pl[0][0] = keep_positive_visible(pl[0][0], r[c][0], r[c][1]);
      pl[0][1] = keep_positive_visible(pl[0][1], r[c][2], r[c][3]);
#pragma unroll
      for (int t = 0; t < NPLP; ++t) *reinterpret_cast<u32x2p*>(hpp + t * HPP_PLANE + (id >> 3) * HPP_ROW + (id & 7) * 8) = pl[t];
    }
  }
}

#define PC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define PC_SPLIT_MMA(A, B, C)        \
  do {                               \
    C = PC_MFMA(A[1], B[0], C);      \
    C = PC_MFMA(A[0], B[1], C);      \
    C = PC_MFMA(A[0], B[0], C);      \
  } while (0)

// max |Wv|, |Wa| over the whole block (the weights' power-of-two scale): call with all 256 threads
__device__ __forceinline__ float pc_weight_max(const float* __restrict__ Wv, const float* __restrict__ Wa, int A, float* red) {
  if (threadIdx.x == 0) *red = 0.f;
  __syncthreads();
  float m = 0.f;
  for (int e = threadIdx.x; e < 512; e += 256) m = fmaxf(m, fabsf(Wv[e]));
  for (int e = threadIdx.x; e < 512 * A; e += 256) m = fmaxf(m, fabsf(Wa[e]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(red), __float_as_uint(m));
  __syncthreads();
  return *red;
}

// forward weights -> hi / lo planes [plane][dd][n = par*CO + co (32, zero padded)][ci(32)]: tap dd of parity par is
// W[pa + 2da][pb + 2db][co][ci]
__device__ __forceinline__ void pc_fwd_weight_planes(const float* __restrict__ Wv, const float* __restrict__ Wa, int A,
                                                     unsigned char* wdp, float S_W) {
  const int CO = 1 + A;
  for (int e = threadIdx.x; e < 4 * 32 * 8; e += 256) {      // one f32x4 of ci per item
    const int c4 = e & 7, n = (e >> 3) & 31, dd = e >> 8;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < 4 * CO) {
      const int par = n / CO, co = n - par * CO;
      const int ky = (par >> 1) + 2 * (dd >> 1), kx = (par & 1) + 2 * (dd & 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = deconv_w(Wv, Wa, A, ky, kx, co, 4 * c4 + k);
    }
    u32x2p pl[NPLP];
    split4p(v, S_W, pl);
#pragma unroll
    for (int t = 0; t < NPLP; ++t) *reinterpret_cast<u32x2p*>(wdp + t * WDP_PLANE + (dd * 32 + n) * WDP_ROW + c4 * 8) = pl[t];
  }
}

// rows 81..83 of the hp planes (the out-of-image taps and the k padding of the weight gradient) stay zero
__device__ __forceinline__ void pc_zero_hp_pad_rows(unsigned char* hpp) {
  for (int e = threadIdx.x; e < NPLP * 3 * HPP_ROW / 4; e += 256) {
    const int t = e / (3 * HPP_ROW / 4), w = e % (3 * HPP_ROW / 4);
    reinterpret_cast<uint32_t*>(hpp + t * HPP_PLANE + C2_POS * HPP_ROW)[w] = 0u;
  }
}

// LDS of the forward kernel for CO = 1 + A output channels: hp planes | dec | dout | weight planes.  dec holds the
// pre-activations [400][DS] with an ODD row stride DS >= CO (a thread reads one position's CO floats: odd strides are
// conflict-free), dout the frame's d_dec [400][CO] dense (it leaves in 16-byte pieces).  At CO <= 7 the workgroup needs
// <= 53 KB: THREE workgroups per CU (round 2: 76 KB with three bf16 planes and stride-8 rows, two per CU, 55 % of the
// wave cycles waiting).
template <int COT> struct FwdLds {
  static constexpr int DS = COT | 1;
  static constexpr int DEC = NPLP * HPP_PLANE, DOUT = DEC + PC_CELLS * DS * 4, W = DOUT + PC_CELLS * COT * 4,
                       BYTES = W + NPLP * WDP_PLANE;
  static constexpr int WGS = 3 * BYTES <= 160 * 1024 ? 3 : 2;
};

// deconv of the wave's position tiles (tile ids mt0 and mt0 + 4 when < 7) for ALL parities and channels:
// column n = par * CO + co (n < 4 * CO <= 32, two 16-wide column tiles); pre-activations + bias -> dec
template <bool TWO_NT, int DS>
__device__ __forceinline__ void deconv_packed(const unsigned char* hpp, const unsigned char* wdp, float* dec, int mt0, int i,
                                              int q, int CO, const float (&bias)[2], float unscale) {
  constexpr int NTL = TWO_NT ? 2 : 1;
  const int ntiles = (mt0 + 4 < 7) ? 2 : 1;              // wave-uniform
  f32x4 acc[2][NTL];
  int a[2], b[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int m = min((mt0 + 4 * t) * 16 + i, 99);
    a[t] = m / 10;
    b[t] = m % 10;
  }
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const int da = dd >> 1, db = dd & 1;
    fh8p bw[NTL][NPLP];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int pl = 0; pl < NPLP; ++pl)
        bw[nt][pl] = *reinterpret_cast<const fh8p*>(wdp + pl * WDP_PLANE + ((dd * 32 + nt * 16 + i)) * WDP_ROW + 16 * q);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < ntiles) {
        const int y = a[t] - da, x = b[t] - db;
        const int row = (y >= 0 && y < 9 && x >= 0 && x < 9) ? y * 9 + x : C2_POS;
        fh8p av[NPLP];
#pragma unroll
        for (int pl = 0; pl < NPLP; ++pl)
          av[pl] = *reinterpret_cast<const fh8p*>(hpp + pl * HPP_PLANE + row * HPP_ROW + 16 * q);
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) PC_SPLIT_MMA(av, bw[nt], acc[t][nt]);
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int n = nt * 16 + i;
    if (n < 4 * CO) {
      const int par = n / CO, co = n - par * CO;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (t < ntiles) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = (mt0 + 4 * t) * 16 + 4 * q + r;
            if (m < 100)
              dec[((2 * (m / 10) + (par >> 1)) * 20 + 2 * (m % 10) + (par & 1)) * DS + co] = acc[t][nt][r] * unscale + bias[nt];
          }
        }
    }
  }
}

// One frame per 256-thread workgroup at a time, THREE independent workgroups per CU at CO <= 7 (two at CO = 8): one's
// dueling / loss VALU phase and its global traffic overlap the others' MFMAs.  COT = 1 + A (8: any A <= 7).
template <int COT>
__global__ __launch_bounds__(256, FwdLds<COT>::WGS) void pc_deconv_fwd_kernel(PcFwdArgs p) {
  typedef FwdLds<COT> L;
  constexpr int DS = L::DS;
  __shared__ __attribute__((aligned(16))) unsigned char smem[L::BYTES];
  __shared__ float s_red;
  const int gtid = threadIdx.x;
  const int lane = threadIdx.x & 63, gw = gtid >> 6;
  const int i = lane & 15, q = lane >> 4;
  unsigned char* hpp = smem;
  float* dec = reinterpret_cast<float*>(smem + L::DEC);
  float* dout = reinterpret_cast<float*>(smem + L::DOUT);      // staged d_dec of the frame: [400][CO] dense
  unsigned char* wdp = smem + L::W;
  const int A = p.A, CO = 1 + p.A;

  // power-of-two scales (exact): hp from its producer's absmax slot, the weights' maximum reduced here
  const float S_W = pow2_scale(pc_weight_max(p.Wv, p.Wa, A, &s_red));
  const float S_HP = pow2_scale(*p.hp_absmax);
  const float unscale = pow2_inv(S_HP) * pow2_inv(S_W);

  pc_fwd_weight_planes(p.Wv, p.Wa, A, wdp, S_W);
  pc_zero_hp_pad_rows(hpp);
  float bias[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = nt * 16 + i;
    const int co = n % CO;
    bias[nt] = (n < 4 * CO) ? (co == 0 ? p.bv[0] : p.ba[co - 1]) : 0.f;
  }
  const bool two_nt = 4 * CO > 16;
  float loss_acc = 0.f, dd_max = 0.f;

  const int stride = gridDim.x;
  f32x4 pre[HP_V];
  {
    const int n0 = blockIdx.x;
    if (n0 < p.N) {
      hp_load(p.hp + (size_t)n0 * F2_DIM, gtid, pre);
      hp_store_planes(hpp, gtid, pre, S_HP);
    }
  }
  int prev = -1;
  for (int base = blockIdx.x; base < p.N; base += stride) {
    const int n = base;
    const bool valid = n < p.N;
    const int nn = n + stride;
    const bool has_next = nn < p.N;
    __syncthreads();  // [S0] hp planes of frame n staged; dout of the previous frame complete
    if (prev >= 0 && p.d_dec) {
      f32x4* dst = reinterpret_cast<f32x4*>(p.d_dec + (size_t)prev * PC_CELLS * CO);
      for (int id = gtid; id < PC_CELLS * CO / 4; id += 256) dst[id] = reinterpret_cast<const f32x4*>(dout)[id];
    }
    if (has_next) hp_load(p.hp + (size_t)nn * F2_DIM, gtid, pre);
    // this frame's targets / action / mask: fetched now so that their latency sits behind the deconv
    float tg[2] = {0.f, 0.f};
    int act = 0;
    bool on = true;
    if (valid && p.d_dec) {
      tg[0] = p.target[(size_t)n * PC_CELLS + gtid];
      tg[1] = p.target[(size_t)n * PC_CELLS + min(gtid + 256, PC_CELLS - 1)];
      act = p.action[n];
      on = p.mask[n] != 0;
    }
    if (valid) {
      if (two_nt) deconv_packed<true, DS>(hpp, wdp, dec, gw, i, q, CO, bias, unscale);
      else deconv_packed<false, DS>(hpp, wdp, dec, gw, i, q, CO, bias, unscale);
    }
    __syncthreads();  // [S1] dec complete; hp free; dout drained
    if (valid) {
      // dueling combine per output position (pre-activations in dec)
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int pos = gtid + 256 * pi;
        if (pos >= PC_CELLS) break;
        const float* d = dec + pos * DS;
        const float vpre = d[0];
        const float V = fmaxf(vpre, 0.f);
        float adv[8], mean = 0.f;
        for (int k = 0; k < A; ++k) { adv[k] = fmaxf(d[1 + k], 0.f); mean += adv[k]; }
        mean /= (float)A;
        if (p.qmax) {
          float mx = V + adv[0] - mean;
          for (int k = 1; k < A; ++k) mx = fmaxf(mx, V + adv[k] - mean);
          p.qmax[(size_t)n * PC_CELLS + pos] = mx;
        }
        if (p.d_dec) {
          const float qa = V + adv[act] - mean;
          const float diff = tg[pi] - qa;
          const float dq = on ? -p.lambda * diff * p.grad_scale : 0.f;
          if (on) loss_acc += 0.5f * p.lambda * diff * diff;
          float* o = dout + pos * CO;
          o[0] = vpre > 0.f ? dq : 0.f;
          dd_max = fmaxf(dd_max, fabsf(dq));          // |d_dec| <= |dq| (the advantage factors are within [-1, 1])
          for (int k = 0; k < A; ++k)
            o[1 + k] = d[1 + k] > 0.f ? dq * (((k == act) ? 1.f : 0.f) - 1.f / (float)A) : 0.f;
        }
      }
    }
    if (has_next) hp_store_planes(hpp, gtid, pre, S_HP);
    prev = valid ? n : -1;
  }
  __syncthreads();
  if (prev >= 0 && p.d_dec) {
    f32x4* dst = reinterpret_cast<f32x4*>(p.d_dec + (size_t)prev * PC_CELLS * CO);
    for (int id = gtid; id < PC_CELLS * CO / 4; id += 256) dst[id] = reinterpret_cast<const f32x4*>(dout)[id];
  }
  // one atomic per workgroup and output (block-uniform pointers): the workgroups of a launch end within microseconds of
  // each other, and same-address atomics serialise
  {
    __shared__ float wsum[4], wmx[4];
    loss_acc = wave_sum(loss_acc);
    dd_max = wave_max(dd_max);
    if (lane == 0) { wsum[gw] = loss_acc; wmx[gw] = dd_max; }
    __syncthreads();
    if (gw == 0) {
      if (p.loss && lane == 0) atomicAdd(p.loss, (((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]) * p.grad_scale);
      if (p.ddec_absmax) absmax_commit(p.ddec_absmax, fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3])));   // bound of max |d_dec|
    }
  }
}

struct PcBwdArgs {
  int N, A;
  const float* hp;          // [N][2592] relu(pc_fc1) (forward activation)
  const float* hp_absmax;   // absmax slot covering hp
  const float* d_dec;       // [N][400][1+A]
  const float* ddec_absmax; // absmax slot covering d_dec (committed by the forward kernel)
  const float* Wv; const float* Wa;
  float* d_hp;              // [N][2592] gradient wrt pc_fc1 PRE-activation (relu mask applied)
  float* dWv; float* dbv; float* dWa; float* dba;
  float* dhp_absmax;        // nullable absmax slot (common.h): max |d_hp|, the A scale of the pc_fc1 dgrad GEMM
};

// Backward on the split-operand scheme of the forward (fp16 hi + lo planes per operand, three term-pair MFMAs):
//   dgrad  d_hp[pos][ci] = sum_{ky,kx,co} d_dec[2y+ky][2x+kx][co] W[ky][kx][co][ci]
//          per ky one 32-deep step: k = (kx, co) is 64 contiguous bytes of the d_dec planes ([400 pos][8 co] bf16);
//   wgrad  dW[(ky,kx,co)][ci] += sum_pos d_dec[2y+ky][2x+kx][co] hp[pos][ci]
//          the reduction index is the position = the row index of both LDS images: both operands come through
//          ds_read_b64_tr_b16 (4 rows x 16 columns, transposed in flight); a block row of d_dec is the 32 bytes of two
//          neighbouring output positions (kx, kx+1) x 8 co, i.e. exactly one 16-row MFMA tile.
typedef short s16x4p __attribute__((ext_vector_type(4)));
typedef short s16x8p __attribute__((ext_vector_type(8)));
constexpr int DDP_ROW = 16;                          // bytes per output position in a d_dec plane (8 co bf16)
constexpr int DDP_PLANE = (PC_CELLS + 4) * DDP_ROW;  // + 4 zero rows (positions past the 81 of the last k step)
constexpr int WBP_PLANE = 4 * 32 * WDP_ROW;          // [ky][ci(32)] rows of 32 (kx,co) bf16 + pad = 10240
constexpr int BWD_GRP_BYTES = NPLP * HPP_PLANE + NPLP * DDP_PLANE + F2_DIM * 4;   // hp planes | d_dec planes | d_hp staging

__device__ __forceinline__ fh8p tr_pair_p(const unsigned char* a0, const unsigned char* a1) {
  typedef s16x4p __attribute__((address_space(3))) * lds_p;
  const s16x4p lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
  const s16x4p hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a1));
  const s16x8p v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(fh8p, v);
}

// one fp32 value -> its fp16 hi / lo terms under `scale`
__device__ __forceinline__ void split1p(float x, float scale, unsigned short (&t)[NPLP]) {
  x *= scale;
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  t[0] = __builtin_bit_cast(unsigned short, h);
  t[1] = __builtin_bit_cast(unsigned short, l);
}

// backward weights -> hi / lo planes [plane][ky][ci(32)][k = kx*8 + co (32, co >= CO zero)]
__device__ __forceinline__ void pc_bwd_weight_planes(const float* __restrict__ Wv, const float* __restrict__ Wa, int A,
                                                     unsigned char* wbp, float S_W) {
  for (int e = threadIdx.x; e < 4 * 32 * 32; e += 256) {
    const int k = e & 31, ci = (e >> 5) & 31, ky = e >> 10;
    const int kx = k >> 3, co = k & 7;
    float v = 0.f;
    if (co == 0) v = Wv[(ky * 4 + kx) * 32 + ci];
    else if (co <= A) v = Wa[((ky * 4 + kx) * A + (co - 1)) * 32 + ci];
    unsigned short t[NPLP];
    split1p(v, S_W, t);
#pragma unroll
    for (int pl = 0; pl < NPLP; ++pl)
      reinterpret_cast<unsigned short*>(wbp + pl * WBP_PLANE + (ky * 32 + ci) * WDP_ROW)[k] = t[pl];
  }
}

// d_dec of a frame, a thread's two output positions (gtid, gtid + 256; 8 channel slots each, channels >= CO zero), split
// into the planes and stored as ONE 16-byte row per plane; the bias gradient sums ride along
__device__ __forceinline__ void pc_stage_dd(unsigned char* ddp, int gtid, const float (&dd)[2][8], float S_DD, float (&adbk)[8]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pos = gtid + 256 * h;
    if (pos < PC_CELLS) {
      u32x2p lo[NPLP], hi[NPLP];
      split4p((f32x4){dd[h][0], dd[h][1], dd[h][2], dd[h][3]}, S_DD, lo);
      split4p((f32x4){dd[h][4], dd[h][5], dd[h][6], dd[h][7]}, S_DD, hi);
#pragma unroll
      for (int pl = 0; pl < NPLP; ++pl) {
        typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4p*>(ddp + pl * DDP_PLANE + pos * DDP_ROW) = (u32x4p){lo[pl][0], lo[pl][1], hi[pl][0], hi[pl][1]};
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) adbk[k] += dd[h][k];
    }
  }
}

// dgrad of one frame: wave gw owns position tiles (gw>>1) + 2jj (jj = 0..2) and channel half nt = gw & 1;
// d_hp (relu mask of pc_fc1 applied: hp > 0 <=> its hi term > 0) -> dhs [81][32]
__device__ __forceinline__ void pc_dgrad_frame(const unsigned char* hpp, const unsigned char* ddp, const unsigned char* wbp,
                                               float* dhs, int gw, int i, int q, float un_dgrad) {
  const int nt = gw & 1;
  f32x4 acc[3];
  int abase[3];
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int pos = min(((gw >> 1) + 2 * jj) * 16 + i, C2_POS - 1);
    abase[jj] = ((2 * (pos / 9)) * 20 + 2 * (pos % 9) + q) * DDP_ROW;      // k chunk q = kx
  }
#pragma unroll
  for (int ky = 0; ky < 4; ++ky) {
    fh8p bw[NPLP];
#pragma unroll
    for (int pl = 0; pl < NPLP; ++pl)
      bw[pl] = *reinterpret_cast<const fh8p*>(wbp + pl * WBP_PLANE + (ky * 32 + nt * 16 + i) * WDP_ROW + 16 * q);
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      fh8p av[NPLP];
#pragma unroll
      for (int pl = 0; pl < NPLP; ++pl)
        av[pl] = *reinterpret_cast<const fh8p*>(ddp + pl * DDP_PLANE + abase[jj] + ky * 20 * DDP_ROW);
      PC_SPLIT_MMA(av, bw, acc[jj]);
    }
  }
#pragma unroll
  for (int jj = 0; jj < 3; ++jj)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pos = ((gw >> 1) + 2 * jj) * 16 + 4 * q + r;
      if (pos < C2_POS) {
        const int ci = nt * 16 + i;
        const unsigned short h0 = reinterpret_cast<const unsigned short*>(hpp + pos * HPP_ROW)[ci];
        dhs[pos * 32 + ci] = (h0 != 0 && !(h0 & 0x8000)) ? acc[jj][r] * un_dgrad : 0.f;
      }
    }
}

// wgrad of one frame into aw: wave gw = ky; K = 81 positions in 3 steps of 32 (rows past 80 read zero rows);
// dW tiles: kxh = 0..1 (kx = 2kxh + (row>>3), co = row&7), nt = 0..1
__device__ __forceinline__ void pc_wgrad_frame(const unsigned char* hpp, const unsigned char* ddp, int gw, int i, int q,
                                               f32x4 (&aw)[2][2]) {
  const int qq = i >> 2, pp = i & 3;             // transposed reads: lane (4qq + pp) of a 16-lane group addresses block row qq
#pragma unroll 1
  for (int ks = 0; ks < 3; ++ks) {
    const int p0 = 32 * ks + 8 * q + qq, p1 = p0 + 4;
    const unsigned char* b0 = hpp + min(p0, C2_POS) * HPP_ROW + 8 * pp;
    const unsigned char* b1 = hpp + min(p1, C2_POS) * HPP_ROW + 8 * pp;
    fh8p bf[2][NPLP];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < NPLP; ++pl) bf[t][pl] = tr_pair_p(b0 + pl * HPP_PLANE + 32 * t, b1 + pl * HPP_PLANE + 32 * t);
    // d_dec block row of position p: output rows (2y+ky, 2x + 2kxh .. +1) x 8 co = 32 contiguous bytes
    const int r0 = p0 < C2_POS ? (2 * (p0 / 9) + gw) * 20 + 2 * (p0 % 9) : PC_CELLS;
    const int r1 = p1 < C2_POS ? (2 * (p1 / 9) + gw) * 20 + 2 * (p1 % 9) : PC_CELLS;
#pragma unroll
    for (int kxh = 0; kxh < 2; ++kxh) {
      fh8p af[NPLP];
#pragma unroll
      for (int pl = 0; pl < NPLP; ++pl)
        af[pl] = tr_pair_p(ddp + pl * DDP_PLANE + (r0 + 2 * kxh) * DDP_ROW + 8 * pp,
                           ddp + pl * DDP_PLANE + (r1 + 2 * kxh) * DDP_ROW + 8 * pp);
      PC_SPLIT_MMA(af, bf[0], aw[kxh][0]);
      PC_SPLIT_MMA(af, bf[1], aw[kxh][1]);
    }
  }
}

// d_hp of the frame leaves in full 128 B lines; returns the running max |d_hp|
__device__ __forceinline__ float pc_drain_dhp(const float* dhs, float* __restrict__ dst_frame, int gtid, float dhp_max) {
  f32x4* dst = reinterpret_cast<f32x4*>(dst_frame);
  for (int id = gtid; id < F2_DIM / 4; id += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(dhs)[id];
    dst[id] = v;
    dhp_max = fmaxf(fmaxf(dhp_max, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  return dhp_max;
}

// end of a workgroup: its dW tiles (one atomic per element), and ONE atomic / commit per workgroup for the bias gradients
// and max |d_hp|.  red: 36 floats of LDS nobody else touches any more.
__device__ __forceinline__ void pc_commit_grads(const f32x4 (&aw)[2][2], float un_wgrad, const float (&adbk)[8], float dhp_max,
                                                int A, float* dWv, float* dWa, float* dbv, float* dba, float* dhp_absmax,
                                                float* red, int gw, int lane, int i, int q) {
  const int gtid = threadIdx.x, CO = 1 + A;
#pragma unroll
  for (int kxh = 0; kxh < 2; ++kxh)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 4 * q + r;                 // row in the 16-row tile: kx = 2kxh + (row>>3), co = row&7
        int kx = 2 * kxh + (row >> 3), co = row & 7, ci = t * 16 + i;
        float v = aw[kxh][t][r] * un_wgrad;
        if (co == 0) atomicAdd(dWv + (gw * 4 + kx) * 32 + ci, v);
        else if (co <= A) atomicAdd(dWa + ((gw * 4 + kx) * A + (co - 1)) * 32 + ci, v);
      }
  float* wsum = red;         // [4][8]
  float* wmx = red + 32;     // [4]
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = wave_sum(adbk[k]);
    if (lane == 0) wsum[gw * 8 + k] = v;
  }
  dhp_max = wave_max(dhp_max);
  if (lane == 0) wmx[gw] = dhp_max;
  __syncthreads();
  if (gtid < CO) {
    const float v = ((wsum[gtid] + wsum[8 + gtid]) + wsum[16 + gtid]) + wsum[24 + gtid];
    if (gtid == 0) atomicAdd(dbv, v);
    else atomicAdd(dba + (gtid - 1), v);
  }
  if (gw == 0) absmax_commit(dhp_absmax, fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3])));
}

// one frame per 256-thread workgroup at a time, two independent workgroups per CU (57 KB of LDS each)
__global__ __launch_bounds__(256, 2) void pc_deconv_bwd_kernel(PcBwdArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[BWD_GRP_BYTES + NPLP * WBP_PLANE];
  __shared__ float s_red[36];
  static_assert(2 * (BWD_GRP_BYTES + NPLP * WBP_PLANE + 144) <= 160 * 1024, "two workgroups must fit one CU's LDS");
  const int gtid = threadIdx.x;
  const int lane = threadIdx.x & 63, gw = gtid >> 6;
  const int i = lane & 15, q = lane >> 4;
  unsigned char* hpp = smem;
  unsigned char* ddp = hpp + NPLP * HPP_PLANE;
  float* dhs = reinterpret_cast<float*>(ddp + NPLP * DDP_PLANE);      // staged d_hp of the frame: [81][32] dense
  unsigned char* wbp = smem + BWD_GRP_BYTES;
  const int A = p.A, CO = 1 + p.A;

  // power-of-two scales (exact): hp / d_dec from their absmax slots, the weights' maximum reduced here
  const float S_W = pow2_scale(pc_weight_max(p.Wv, p.Wa, A, s_red));
  const float S_HP = pow2_scale(*p.hp_absmax), S_DD = pow2_scale(*p.ddec_absmax);
  const float un_dgrad = pow2_inv(S_DD) * pow2_inv(S_W);       // d_hp = d_dec . W
  const float un_wgrad = pow2_inv(S_DD) * pow2_inv(S_HP);      // dW   = d_dec^T . hp

  pc_bwd_weight_planes(p.Wv, p.Wa, A, wbp, S_W);
  pc_zero_hp_pad_rows(hpp);
  // d_dec planes: the co >= CO padding columns and the 4 extra rows stay zero (never rewritten)
  for (int e = gtid; e < NPLP * DDP_PLANE / 4; e += 256) reinterpret_cast<uint32_t*>(ddp)[e] = 0u;

  f32x4 aw[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) aw[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float adbk[8];           // bias gradient: one accumulator per channel slot
#pragma unroll
  for (int k = 0; k < 8; ++k) adbk[k] = 0.f;
  float dhp_max = 0.f;     // max |d_hp| this thread has stored

  const int stride = gridDim.x;
  f32x4 pre_hp[HP_V];
  // d_dec of a frame ([400][CO] fp32, dense): a thread owns output positions gtid and gtid + 256 -- CO consecutive floats
  // each (a wave's loads cover one contiguous 64 * 4 * CO byte run).  The first form of this staging dealt f32x4 pieces to
  // threads: a runtime division per element to find its position, three 2-byte LDS writes per element and eight
  // compare-selects for the bias sums -- about as much VALU as the rest of the kernel.
  float pre_dd[2][8];
  auto load_dd = [&](int frame) {
    const float* src = p.d_dec + (size_t)frame * PC_CELLS * CO;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pos = min(gtid + 256 * h, PC_CELLS - 1);
#pragma unroll
      for (int co = 0; co < 8; ++co) pre_dd[h][co] = (co < CO) ? src[pos * CO + co] : 0.f;
    }
  };

  __syncthreads();   // zero fills visible before the first scatter
  {
    const int n0 = blockIdx.x;
    if (n0 < p.N) {
      hp_load(p.hp + (size_t)n0 * F2_DIM, gtid, pre_hp);
      hp_store_planes(hpp, gtid, pre_hp, S_HP);
      load_dd(n0);
      pc_stage_dd(ddp, gtid, pre_dd, S_DD, adbk);
    }
  }
  for (int n = blockIdx.x; n < p.N; n += stride) {
    const int nn = n + stride;
    const bool has_next = nn < p.N;
    __syncthreads();  // [S0] hp / d_dec planes of frame n staged; dhs drained
    if (has_next) {
      hp_load(p.hp + (size_t)nn * F2_DIM, gtid, pre_hp);
      load_dd(nn);
    }
    pc_dgrad_frame(hpp, ddp, wbp, dhs, gw, i, q, un_dgrad);
    pc_wgrad_frame(hpp, ddp, gw, i, q, aw);
    __syncthreads();  // [S1] all reads of the planes done; dhs of frame n complete
    dhp_max = pc_drain_dhp(dhs, p.d_hp + (size_t)n * F2_DIM, gtid, dhp_max);
    if (has_next) {
      hp_store_planes(hpp, gtid, pre_hp, S_HP);
      pc_stage_dd(ddp, gtid, pre_dd, S_DD, adbk);
    }
  }
  pc_commit_grads(aw, un_wgrad, adbk, dhp_max, A, p.dWv, p.dWa, p.dbv, p.dba, p.dhp_absmax, s_red, gw, lane, i, q);
}

// ---- forward (loss) + backward of the training pass in ONE kernel ---------------------------------------------------
// A frame's d_dec depends on that frame alone, and both kernels above stage the same hp image: fused, d_dec never leaves
// the CU (16 KB per frame of HBM traffic: written by the forward, read by the backward) and hp is read and split once
// instead of twice (10 KB per frame) -- 22.3 KB per frame against 48.7.  d_dec's power-of-two scale is then the FRAME's
// (max |dq| over its 400 cells, met in LDS) instead of the launch's: finer, and no slot / second launch needed; the
// weight gradient is accumulated per frame from zero and added to the running tiles under that frame's inverse scale.
struct PcTrainArgs {
  int N, A;
  const float* hp; const float* hp_absmax;
  const float* Wv; const float* bv; const float* Wa; const float* ba;
  const int* action; const float* target; const int* mask;
  float lambda, grad_scale;
  float* loss;
  float* d_hp; float* dhp_absmax;
  float* dWv; float* dbv; float* dWa; float* dba;
  float* d_dec;             // nullable: the frames' d_dec [N][400][1+A] (tests; the trainer does not ask for it)
};

template <int COT> struct TrainLds {
  static constexpr int DS = COT | 1;
  static constexpr int UNI = PC_CELLS * DS * 4 > F2_DIM * 4 ? PC_CELLS * DS * 4 : F2_DIM * 4;   // dec, later dhs
  static constexpr int DDP = NPLP * HPP_PLANE, U = DDP + NPLP * DDP_PLANE, WF = U + UNI, WB = WF + NPLP * WDP_PLANE,
                       RED = WB + NPLP * WBP_PLANE, BYTES = RED + 192;
  static constexpr int WGS = 2 * BYTES <= 160 * 1024 ? 2 : 1;
};

template <int COT>
__global__ __launch_bounds__(256, TrainLds<COT>::WGS) void pc_deconv_train_kernel(PcTrainArgs p) {
  typedef TrainLds<COT> L;
  constexpr int DS = L::DS;
  __shared__ __attribute__((aligned(16))) unsigned char smem[L::BYTES];
  const int gtid = threadIdx.x;
  const int lane = threadIdx.x & 63, gw = gtid >> 6;
  const int i = lane & 15, q = lane >> 4;
  unsigned char* hpp = smem;
  unsigned char* ddp = smem + L::DDP;
  float* uni = reinterpret_cast<float*>(smem + L::U);      // dec [400][DS] until the frame's d_dec is staged, then dhs [81][32]
  unsigned char* wdp = smem + L::WF;
  unsigned char* wbp = smem + L::WB;
  float* red = reinterpret_cast<float*>(smem + L::RED);    // [0..35] weight max / final sums; [40..43] the frame's wave maxima
  const int A = p.A, CO = 1 + p.A;

  const float S_W = pow2_scale(pc_weight_max(p.Wv, p.Wa, A, red));
  const float S_HP = pow2_scale(*p.hp_absmax);
  const float un_fwd = pow2_inv(S_HP) * pow2_inv(S_W);
  pc_fwd_weight_planes(p.Wv, p.Wa, A, wdp, S_W);
  pc_bwd_weight_planes(p.Wv, p.Wa, A, wbp, S_W);
  pc_zero_hp_pad_rows(hpp);
  for (int e = gtid; e < NPLP * DDP_PLANE / 4; e += 256) reinterpret_cast<uint32_t*>(ddp)[e] = 0u;
  float bias[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = nt * 16 + i;
    const int co = n % CO;
    bias[nt] = (n < 4 * CO) ? (co == 0 ? p.bv[0] : p.ba[co - 1]) : 0.f;
  }
  const bool two_nt = 4 * CO > 16;

  f32x4 aw[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) aw[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float adbk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) adbk[k] = 0.f;
  float dhp_max = 0.f, loss_acc = 0.f;

  const int stride = gridDim.x;
  f32x4 pre[HP_V];
  __syncthreads();   // zero fills visible
  if (blockIdx.x < p.N) {
    hp_load(p.hp + (size_t)blockIdx.x * F2_DIM, gtid, pre);
    hp_store_planes(hpp, gtid, pre, S_HP);
  }
  for (int n = blockIdx.x; n < p.N; n += stride) {
    const int nn = n + stride;
    const bool has_next = nn < p.N;
    __syncthreads();  // [S0] hp planes of frame n staged; the previous frame's d_hp drained
    if (has_next) hp_load(p.hp + (size_t)nn * F2_DIM, gtid, pre);
    const float tg[2] = {p.target[(size_t)n * PC_CELLS + gtid], p.target[(size_t)n * PC_CELLS + min(gtid + 256, PC_CELLS - 1)]};
    const int act = p.action[n];
    const bool on = p.mask[n] != 0;
    if (two_nt) deconv_packed<true, DS>(hpp, wdp, uni, gw, i, q, CO, bias, un_fwd);
    else deconv_packed<false, DS>(hpp, wdp, uni, gw, i, q, CO, bias, un_fwd);
    __syncthreads();  // [S1] pre-activations complete
    // dueling combine, loss and d_dec of this thread's two output positions (pc_deconv_fwd_kernel's arithmetic)
    float dd[2][8];
    float fmx = 0.f;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
      for (int k = 0; k < 8; ++k) dd[pi][k] = 0.f;
      const int pos = gtid + 256 * pi;
      if (pos < PC_CELLS) {
        const float* d = uni + pos * DS;
        const float vpre = d[0];
        const float V = fmaxf(vpre, 0.f);
        float apre[COT - 1], mean = 0.f, aact = 0.f;
#pragma unroll
        for (int k = 0; k < COT - 1; ++k) {
          apre[k] = k < A ? d[1 + k] : 0.f;
          const float ad = fmaxf(apre[k], 0.f);
          mean += ad;
          aact = (k == act) ? ad : aact;
        }
        mean /= (float)A;
        const float qa = V + aact - mean;
        const float diff = tg[pi] - qa;
        const float dq = on ? -p.lambda * diff * p.grad_scale : 0.f;
        if (on) loss_acc += 0.5f * p.lambda * diff * diff;
        fmx = fmaxf(fmx, fabsf(dq));          // |d_dec| <= |dq| (the advantage factors are within [-1, 1])
        dd[pi][0] = vpre > 0.f ? dq : 0.f;
#pragma unroll
        for (int k = 0; k < COT - 1; ++k)
          dd[pi][1 + k] = (k < A && apre[k] > 0.f) ? dq * (((k == act) ? 1.f : 0.f) - 1.f / (float)A) : 0.f;
        if (p.d_dec) {
          float* o = p.d_dec + ((size_t)n * PC_CELLS + pos) * CO;
#pragma unroll
          for (int k = 0; k < COT; ++k)
            if (k < CO) o[k] = dd[pi][k];
        }
      }
    }
    fmx = wave_max(fmx);
    if (lane == 0) red[40 + gw] = fmx;
    __syncthreads();  // [S2] the frame's max |dq|
    const float S_DD = pow2_scale(fmaxf(fmaxf(red[40], red[41]), fmaxf(red[42], red[43])));
    const float inv_dd = pow2_inv(S_DD);
    pc_stage_dd(ddp, gtid, dd, S_DD, adbk);
    __syncthreads();  // [S3] d_dec planes staged; every read of dec done
    pc_dgrad_frame(hpp, ddp, wbp, uni, gw, i, q, inv_dd * pow2_inv(S_W));
    {
      f32x4 awf[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) awf[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      pc_wgrad_frame(hpp, ddp, gw, i, q, awf);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) aw[a][b] += awf[a][b] * inv_dd;
    }
    __syncthreads();  // [S4] d_hp of frame n complete; all reads of the planes done
    dhp_max = pc_drain_dhp(uni, p.d_hp + (size_t)n * F2_DIM, gtid, dhp_max);
    if (has_next) hp_store_planes(hpp, gtid, pre, S_HP);
  }
  __syncthreads();
  loss_acc = wave_sum(loss_acc);
  if (lane == 0) red[44 + gw] = loss_acc;
  pc_commit_grads(aw, pow2_inv(S_HP), adbk, dhp_max, A, p.dWv, p.dWa, p.dbv, p.dba, p.dhp_absmax, red, gw, lane, i, q);
  if (gtid == 0) atomicAdd(p.loss, (((red[44] + red[45]) + red[46]) + red[47]) * p.grad_scale);
}

}  // namespace

extern "C" {

int unreal_pc_deconv_fwd(int N, int A, const float* hp, const float* hp_absmax, const float* Wv, const float* bv,
                         const float* Wa, const float* ba, float* qmax, const int* action, const float* target,
                         const int* mask, float lambda, float grad_scale, float* d_dec, float* ddec_absmax, float* loss,
                         void* stream) {
  if (N <= 0 || A <= 0 || A > 7 || !hp || !hp_absmax || !Wv || !bv || !Wa || !ba) return UNREAL_EINVAL;
  if (!qmax && !d_dec) return UNREAL_EINVAL;
  if (d_dec && (!action || !target || !mask || !loss)) return UNREAL_EINVAL;
  if ((((uintptr_t)hp) | ((uintptr_t)d_dec)) & 15) return UNREAL_EINVAL;
  PcFwdArgs p{N, A, hp, hp_absmax, Wv, bv, Wa, ba, qmax, action, target, mask, lambda, grad_scale, d_dec,
              d_dec ? ddec_absmax : nullptr, d_dec ? loss : nullptr};
  hipStream_t st = (hipStream_t)stream;
  // one frame per workgroup at a time; as many workgroups as fit the chip (3 per CU at 1 + A <= 7 channels)
#define PC_FWD(COT) hipLaunchKernelGGL(pc_deconv_fwd_kernel<COT>, dim3(min(N, 256 * FwdLds<COT>::WGS)), dim3(256), 0, st, p)
  switch (1 + A) {
    case 4: PC_FWD(4); break;      // indoor (A = 3)
    case 5: PC_FWD(5); break;      // maze (A = 4)
    case 7: PC_FWD(7); break;      // lab (A = 6)
    default: PC_FWD(8); break;
  }
#undef PC_FWD
  return unreal_launch_status();
}

int unreal_pc_deconv_bwd(int N, int A, const float* hp, const float* hp_absmax, const float* d_dec, const float* ddec_absmax,
                         const float* Wv, const float* Wa, float* d_hp, float* dhp_absmax, float* dWv, float* dbv, float* dWa,
                         float* dba, void* stream) {
  if (N <= 0 || A <= 0 || A > 7 || !hp || !hp_absmax || !d_dec || !ddec_absmax || !Wv || !Wa || !d_hp || !dWv || !dbv ||
      !dWa || !dba)
    return UNREAL_EINVAL;
  if ((((uintptr_t)hp) | ((uintptr_t)d_dec) | ((uintptr_t)d_hp)) & 15) return UNREAL_EINVAL;
  PcBwdArgs p{N, A, hp, hp_absmax, d_dec, ddec_absmax, Wv, Wa, d_hp, dWv, dbv, dWa, dba, dhp_absmax};
  int blocks = min(N, 512);
  hipLaunchKernelGGL(pc_deconv_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  return unreal_launch_status();
}

// Training pass of the pixel-control head in one launch: loss (accumulated into *loss like unreal_pc_deconv_fwd), d_hp,
// the four parameter gradients (accumulated) and max |d_hp|.  d_dec (nullable) receives the frames' d_dec for inspection.
int unreal_pc_deconv_train(int N, int A, const float* hp, const float* hp_absmax, const float* Wv, const float* bv,
                           const float* Wa, const float* ba, const int* action, const float* target, const int* mask,
                           float lambda, float grad_scale, float* loss, float* d_hp, float* dhp_absmax, float* dWv, float* dbv,
                           float* dWa, float* dba, float* d_dec, void* stream) {
  if (N <= 0 || A <= 0 || A > 7 || !hp || !hp_absmax || !Wv || !bv || !Wa || !ba || !action || !target || !mask || !loss ||
      !d_hp || !dWv || !dbv || !dWa || !dba)
    return UNREAL_EINVAL;
  if ((((uintptr_t)hp) | ((uintptr_t)d_hp)) & 15) return UNREAL_EINVAL;
  PcTrainArgs p{N, A, hp, hp_absmax, Wv, bv, Wa, ba, action, target, mask, lambda, grad_scale, loss, d_hp, dhp_absmax,
                dWv, dbv, dWa, dba, d_dec};
  hipStream_t st = (hipStream_t)stream;
#define PC_TRAIN(COT) \
  hipLaunchKernelGGL(pc_deconv_train_kernel<COT>, dim3(min(N, 256 * TrainLds<COT>::WGS)), dim3(256), 0, st, p)
  switch (1 + A) {
    case 4: PC_TRAIN(4); break;      // indoor (A = 3)
    case 5: PC_TRAIN(5); break;      // maze (A = 4)
    case 7: PC_TRAIN(7); break;      // lab (A = 6)
    default: PC_TRAIN(8); break;
  }
#undef PC_TRAIN
  return unreal_launch_status();
}

}  // extern "C"
