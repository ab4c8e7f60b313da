// EXPERIMENT (not built into libunreal_hip.so): encoder backward with conv2 wgrad + dgrad on split bf16 operands.
// Drop-in replacement section for unreal_amd/csrc/encoder.hip (paste before the closing of its anonymous namespace and
// launch encoder_bwd_split_kernel<7> from unreal_encoder_bwd).  Passes tests/test_kernels_gpu.py -k encoder.
//
// Measured on MI355X (tools/exp/ablate_encoder_bwd.py, 81920 random frames, ms):
//                       phase 1   phase 2   phase 3   all
//   fp32 MFMA kernel     1.59      1.82      1.33     4.31
//   this kernel          1.34      1.83      1.46     5.93   (452 B/lane scratch in the combined kernel)
// The 2.7x cut in MFMA cycles does not show: after the conversion the phases are bound by the serial sum of address /
// split VALU, LDS fragment reads and their latencies on 2 waves per SIMD (phase 2: ~8.8k VALU + 5.4k MFMA + 5.4k LDS
// cycles per frame pair ~= the 21.7k measured), and with no LDS left for W2 and 80 accumulator registers per lane the
// combined kernel spills.  Kept for the next attempt: what would have to change is the work split (more waves per
// frame, W2 planes resident in LDS, packed per-lane address tables), not the arithmetic.

// ------------------------------------------------------------------------------------------------
// Backward, phases (1) and (2) on the bf16 matrix cores with split operands (same scheme and error level
// as csrc/gemm_split.hip): c1 and d2 are split into three bf16 planes when they are staged into LDS, W2
// when its fragment is loaded, and every product tile accumulates the six term pairs of weight >= 2^-16.
//   (1) wgrad: the reduction index is the POSITION, which is the row index of both LDS images, so both
//       operands come through ds_read_b64_tr_b16 (a 4-row x 16-column block, transposed in flight): each
//       lane supplies the address of one row, so the stride-2 conv2 taps need no im2col copy.
//   (2) dgrad: the reduction index is d2's channel (contiguous in a row): plain 16-byte fragment reads;
//       W2 fragments come straight from global memory (32 KB, L2-resident) -- there is no LDS left for them.
// LDS per frame group: uint8 frame | c1 planes [3][400][16] bf16 (re-used as d1 fp32 [400][20] once phase
// (1) is done) | d2 planes [3][84][32(+8)] bf16 (rows 81.. zero) | ReLU mask of c1 (4 bits per byte).
// Phase (3) and the register-prefetch pipeline are those of the fp32 kernel above.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));

constexpr int C1P_ROW = 32;                        // bytes per conv1 position (16 bf16)
constexpr int C1P_PLANE = C1_POS * C1P_ROW;        // 12800
constexpr int C1P_BYTES = 3 * C1P_PLANE;           // 38400 >= d1 fp32 [400][C1_LD] (32000)
constexpr int D2P_ROW = 80;                        // 32 bf16 + 16 B pad (conflict-free 16-byte row reads)
constexpr int D2P_ROWS = 84;
constexpr int D2P_PLANE = D2P_ROWS * D2P_ROW;      // 6720
constexpr int D2P_BYTES = 3 * D2P_PLANE;           // 20160
constexpr int MASK_BYTES = C1_POS * 4;             // one byte per (position, 4 channels)
constexpr int GRP2_BYTES = FR_LDS + C1P_BYTES + D2P_BYTES + MASK_BYTES;   // 81344; two groups = 162688 <= 160 KiB
static_assert(C1P_BYTES >= C1_LDS * 4, "d1 must fit in the c1 planes");
static_assert(2 * GRP2_BYTES <= 160 * 1024, "LDS budget");

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

__device__ __forceinline__ void stage_c1_planes(unsigned char* c1p, unsigned char* mask, int gtid, const f32x4 (&pc1)[C1_V]) {
#pragma unroll
  for (int c = 0; c < C1_V; ++c) {
    const int id = gtid + 256 * c;
    if (id < C1_POS * 4) {
      u32x2v pl[3];
      split4(pc1[c], pl);
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x2v*>(c1p + t * C1P_PLANE + (id >> 2) * C1P_ROW + (id & 3) * 8) = pl[t];
      mask[id] = (unsigned char)((pc1[c][0] > 0.f ? 1 : 0) | (pc1[c][1] > 0.f ? 2 : 0) | (pc1[c][2] > 0.f ? 4 : 0) |
                                 (pc1[c][3] > 0.f ? 8 : 0));
    }
  }
}

__device__ __forceinline__ void stage_d2_planes(unsigned char* d2p, int gtid, const f32x4 (&pd2)[D2_V], float (&adb2)[4]) {
#pragma unroll
  for (int c = 0; c < D2_V; ++c) {
    const int id = gtid + 256 * c;
    if (id < C2_POS * 8) {
      u32x2v pl[3];
      split4(pd2[c], pl);
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x2v*>(d2p + t * D2P_PLANE + (id >> 3) * D2P_ROW + (id & 7) * 8) = pl[t];
#pragma unroll
      for (int e = 0; e < 4; ++e) adb2[e] += pd2[c][e];
    }
  }
}

// two transposed 4-row blocks -> the 8 consecutive-k values of one 16x16x32 operand lane
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* a0, const unsigned char* a1) {
  typedef s16x4v __attribute__((address_space(3))) * lds_p;
  const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
  const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a1));
  const s16x8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// six term pairs, smallest first
#define SPLIT_MMA(A, B, C)            \
  do {                                \
    C = MFMA_BF16(A[2], B[0], C);     \
    C = MFMA_BF16(A[0], B[2], C);     \
    C = MFMA_BF16(A[1], B[1], C);     \
    C = MFMA_BF16(A[1], B[0], C);     \
    C = MFMA_BF16(A[0], B[1], C);     \
    C = MFMA_BF16(A[0], B[0], C);     \
  } while (0)

// conv2 dgrad on split operands for NT position tiles (16 positions each) of output parity `par`, starting at tile mt0:
// the W2 fragment of a tap (fp32 in registers, fetched by the caller before phase (1); split here) feeds all NT tiles.
template <int NT>
__device__ __forceinline__ float dgrad_split_tiles(const unsigned char* d2p, float* d1, const unsigned char* mask,
                                                   const f32x4 (&wv)[4][2], int par, int mt0, int i, int q, int zero) {
  // `zero` is an opaque 0 made once per frame: without it every row / position offset below is frame-invariant, gets
  // hoisted out of the frame loop as a table of ~100 registers and spills
  f32x4 acc[NT];
  int ma[NT], mb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int m = min((mt0 + t) * 16 + i + zero, 99);
    ma[t] = m / 10;
    mb[t] = m % 10;
  }
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const int da = dd >> 1, db = dd & 1;
    bf16x8 wb[3];
    {
      u32x2v lo[3], hi[3];
      split4(wv[dd][0], lo);
      split4(wv[dd][1], hi);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const u32x4 w4 = {lo[t][0], lo[t][1], hi[t][0], hi[t][1]};
        wb[t] = __builtin_bit_cast(bf16x8, w4);
      }
    }
    // all 3*NT fragment reads of the tap first, then its 6*NT MFMAs: one exposed LDS latency per tap instead of one
    // per tile (left alone, the compiler interleaves read / wait / MFMA tile by tile to save registers)
    bf16x8 af[NT][3];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int y = ma[t] - da, x = mb[t] - db;
      const int row = (y >= 0 && y < 9 && x >= 0 && x < 9) ? y * 9 + x : C2_POS;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        af[t][pl] = *reinterpret_cast<const bf16x8*>(d2p + pl * D2P_PLANE + row * D2P_ROW + 16 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) SPLIT_MMA(af[t], wb, acc[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // masked store of d1 (branch-free: rows past the 100 positions of the parity go to a scratch row of the d2 image)
  float db1 = 0.f;
  unsigned int mk[NT][4];
  int pos[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = min((mt0 + t) * 16 + 4 * q + r + zero, 99);
      pos[t][r] = (2 * (m / 10) + (par >> 1)) * 20 + 2 * (m % 10) + (par & 1);
      mk[t][r] = mask[pos[t][r] * 4 + (i >> 2)];
    }
  float* dump = reinterpret_cast<float*>(const_cast<unsigned char*>(d2p) + (C2_POS + 1) * D2P_ROW);   // rows 82..83: 40 floats
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = (mt0 + t) * 16 + 4 * q + r < 100;        // compile-time true except in the last tile
      const float g = (ok && ((mk[t][r] >> (i & 3)) & 1)) ? acc[t][r] : 0.f;
      float* dst = ok ? d1 + pos[t][r] * C1_LD + i : dump + i;
      *dst = g;
      db1 += g;
    }
  return db1;
}

template <int PHASES>   // bit 0/1/2 = phase (1)/(2)/(3); 7 in the product, other values only for ablation timing
__global__ __launch_bounds__(512) void encoder_bwd_split_kernel(int N, const uint8_t* __restrict__ frames,
                                                                const int* __restrict__ frame_idx, float scale,
                                                                const float* __restrict__ W2,
                                                                const float* __restrict__ c1_saved,
                                                                const float* __restrict__ d2_in, float* __restrict__ dW1,
                                                                float* __restrict__ db1, float* __restrict__ dW2,
                                                                float* __restrict__ db2) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * GRP2_BYTES];
  const int grp = threadIdx.x >> 8, gtid = threadIdx.x & 255;
  const int lane = threadIdx.x & 63, gw = gtid >> 6;
  const int i = lane & 15, q = lane >> 4;
  uint8_t* fr = smem + grp * GRP2_BYTES;
  unsigned char* c1p = fr + FR_LDS;
  float* c1 = reinterpret_cast<float*>(c1p);            // d1 (fp32, [400][C1_LD]) after phase (1)
  unsigned char* d2p = c1p + C1P_BYTES;
  unsigned char* mask = d2p + D2P_BYTES;
  const unsigned char* zrow = d2p + C2_POS * D2P_ROW;   // 80 zero bytes (row 81 of plane 0)
  for (int e = gtid; e < 3 * 3 * D2P_ROW / 4; e += 256) {     // zero rows 81..83 of the three d2 planes
    const int t = e / (3 * D2P_ROW / 4), w = e % (3 * D2P_ROW / 4);
    reinterpret_cast<uint32_t*>(d2p + t * D2P_PLANE + C2_POS * D2P_ROW)[w] = 0u;
  }

  f32x4 aw2[4][2];      // dW2 tiles: ky = gw, kx = 0..3, nt = 0..1
  f32x4 aw1[3][4];      // dW1 tiles (g,t): patch elements m = 64g + 4*row + t, all 12 tiles, THIS wave's 100 positions
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) aw2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) aw1[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float adb2[4] = {0.f, 0.f, 0.f, 0.f};   // n = (gtid % 8) * 4 + e
  float adb1 = 0.f;                       // channel i

  int off1[3];          // byte offset of patch element m = 64g + 4i in the 8x8x3 patch (4 consecutive m = one dword)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int m = 64 * a + 4 * i;
    off1[a] = (m / 24) * FRAME_ROW_BYTES + (m % 24);
  }

  const int stride = gridDim.x * 2;
  f32x4 pc1[C1_V], pd2[D2_V];
  u32x4 pfr[FR_V];
  {
    const int n0 = blockIdx.x * 2 + grp;
    if (n0 < N) {
      frame_load(frames + (size_t)frame_idx[n0] * FRAME_BYTES, gtid, pfr);
      frame_store(fr, gtid, pfr);
      const f32x4* cs = reinterpret_cast<const f32x4*>(c1_saved + (size_t)n0 * (C1_POS * C1_CH));
#pragma unroll
      for (int c = 0; c < C1_V; ++c) {
        int id = gtid + 256 * c;
        pc1[c] = cs[id < C1_POS * 4 ? id : gtid];
      }
      stage_c1_planes(c1p, mask, gtid, pc1);
      const f32x4* ds = reinterpret_cast<const f32x4*>(d2_in + (size_t)n0 * F2_DIM);
#pragma unroll
      for (int c = 0; c < D2_V; ++c) {
        int id = gtid + 256 * c;
        pd2[c] = ds[id < C2_POS * 8 ? id : gtid];
      }
      stage_d2_planes(d2p, gtid, pd2, adb2);
    }
  }

  const int qq = i >> 2, pp = i & 3;       // transposed reads: lane (4qq + pp) of a 16-lane group addresses block row qq
  for (int base = blockIdx.x * 2; base < N; base += stride) {
    const int n = base + grp;
    const bool valid = n < N;
    const int nn = n + stride;
    const bool has_next = nn < N;
    __syncthreads();  // [S0] planes / mask / frame of frame n staged
    // W2 fragments of phase (2) (this wave's output parity; 8 consecutive n of row (ky,kx,c = i) per tap): fetched
    // here so that phase (1) hides the L2 latency.  The offset goes through an opaque zero: a frame-invariant
    // address lets the compiler hoist the loads AND their bf16 split out of the frame loop, and those 80 registers
    // then live through phase (3) and spill.
    f32x4 wv[4][2];
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    if (valid && (PHASES & 2)) {
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        const int ky = (gw >> 1) + 2 * (dd >> 1), kx = (gw & 1) + 2 * (dd & 1);
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(W2 + ((ky * 4 + kx) * 16 + i) * 32 + 8 * q + zero);
        wv[dd][0] = wsrc[0];
        wv[dd][1] = wsrc[1];
      }
    }
    if (valid && (PHASES & 1)) {
      // (1) conv2 wgrad: dW2[(ky=gw,kx,c)][n] += sum_p c1[2oy+ky][2ox+kx][c] * d2[p][n]; K = 81 positions in 3 steps of 32
#pragma unroll 1
      for (int ks = 0; ks < 3; ++ks) {
        const int p0 = 32 * ks + 8 * q + qq, p1 = p0 + 4;
        const unsigned char* b0 = d2p + min(p0, C2_POS) * D2P_ROW + 8 * pp;
        const unsigned char* b1 = d2p + min(p1, C2_POS) * D2P_ROW + 8 * pp;
        bf16x8 bf[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int t = 0; t < 3; ++t) bf[nt][t] = tr_pair(b0 + t * D2P_PLANE + 32 * nt, b1 + t * D2P_PLANE + 32 * nt);
        const int r0 = (2 * (p0 / 9) + gw) * 20 + 2 * (p0 % 9), r1 = (2 * (p1 / 9) + gw) * 20 + 2 * (p1 % 9);
        const bool z0 = p0 >= C2_POS, z1 = p1 >= C2_POS;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          bf16x8 af[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const unsigned char* a0 = z0 ? zrow + 8 * pp : c1p + t * C1P_PLANE + (r0 + kx) * C1P_ROW + 8 * pp;
            const unsigned char* a1 = z1 ? zrow + 8 * pp : c1p + t * C1P_PLANE + (r1 + kx) * C1P_ROW + 8 * pp;
            af[t] = tr_pair(a0, a1);
          }
          SPLIT_MMA(af, bf[0], aw2[kx][0]);
          SPLIT_MMA(af, bf[1], aw2[kx][1]);
        }
      }
    }
    __syncthreads();  // [S1] all reads of the c1 planes done before d1 overwrites them

    if (valid && (PHASES & 2)) {
      // (2) conv2 dgrad, wave gw = output parity: d1[2a+pa][2b+pb][c] = sum_{dd,n} d2[a-da][b-db][n] W2[pa+2da][pb+2db][c][n]
      adb1 += dgrad_split_tiles<4>(d2p, c1, mask, wv, gw, 0, i, q, zero);
      adb1 += dgrad_split_tiles<3>(d2p, c1, mask, wv, gw, 4, i, q, zero);
    }
    __syncthreads();  // [S2] d1 complete; d2 planes free

    if (has_next) {   // fetch the next frame's uint8 image, c1 and d2 behind phase (3)
      frame_load(frames + (size_t)frame_idx[nn] * FRAME_BYTES, gtid, pfr);
      const f32x4* cs = reinterpret_cast<const f32x4*>(c1_saved + (size_t)nn * (C1_POS * C1_CH));
#pragma unroll
      for (int c = 0; c < C1_V; ++c) {
        int id = gtid + 256 * c;
        pc1[c] = cs[id < C1_POS * 4 ? id : gtid];
      }
      const f32x4* ds = reinterpret_cast<const f32x4*>(d2_in + (size_t)nn * F2_DIM);
#pragma unroll
      for (int c = 0; c < D2_V; ++c) {
        int id = gtid + 256 * c;
        pd2[c] = ds[id < C2_POS * 8 ? id : gtid];
      }
    }
    if (valid && (PHASES & 4)) {
      // (3) conv1 wgrad: identical to the fp32 kernel's phase (3) (exact-product bf16, d1 split hi/mid/lo on the fly)
      const uint8_t* frow = fr + (4 * 5 * gw) * FRAME_ROW_BYTES;
#pragma unroll 1
      for (int kc = 0; kc < 4; ++kc) {
        u32x4 bpl[3];
        {
          uint32_t t0[8], t1[8], t2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int sl = 32 * kc + 8 * q + j;
            const float v = sl < 100 ? c1[(100 * gw + sl) * C1_LD + i] : 0.f;
            t0[j] = __float_as_uint(v) & 0xffff0000u;
            const float r1 = v - __uint_as_float(t0[j]);
            t1[j] = __float_as_uint(r1) & 0xffff0000u;
            t2[j] = __float_as_uint(r1 - __uint_as_float(t1[j]));     // <= 8 significant bits left: exact in bf16
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bpl[0][e] = __builtin_amdgcn_perm(t0[2 * e + 1], t0[2 * e], 0x07060302u);
            bpl[1][e] = __builtin_amdgcn_perm(t1[2 * e + 1], t1[2 * e], 0x07060302u);
            bpl[2][e] = __builtin_amdgcn_perm(t2[2 * e + 1], t2[2 * e], 0x07060302u);
          }
        }
        int pofs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int sl = min(32 * kc + 8 * q + j, 99);                // padding slots: any valid address (B = 0)
          pofs[j] = (4 * (sl / 20)) * FRAME_ROW_BYTES + 12 * (sl % 20);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          uint32_t w[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const uint32_t*>(frow + pofs[j] + off1[g]);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float f0 = (float)((w[2 * e] >> (8 * t)) & 0xffu);
              const float f1 = (float)((w[2 * e + 1] >> (8 * t)) & 0xffu);
              pk[e] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
            }
            const bf16x8 av = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
              aw1[g][t] = MFMA_BF16(av, __builtin_bit_cast(bf16x8, bpl[tm]), aw1[g][t]);
          }
        }
      }
    }
    if (has_next) stage_d2_planes(d2p, gtid, pd2, adb2);
    __syncthreads();  // [S3] phase (3) finished reading d1 and the frame
    if (has_next) {
      stage_c1_planes(c1p, mask, gtid, pc1);
      frame_store(fr, gtid, pfr);
    }
  }

  // flush accumulators (C/D map of the 16x16 MFMAs: col = lane & 15, row = 4 * (lane >> 4) + r)
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(dW2 + ((gw * 4 + kx) * 16 + 4 * q + r) * 32 + nt * 16 + i, aw2[kx][nt][r]);
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(dW1 + (64 * g + 4 * (4 * q + r) + t) * 16 + i, scale * aw1[g][t][r]);
  adb1 += __shfl_xor(adb1, 16, 64);
  adb1 += __shfl_xor(adb1, 32, 64);
  if (q == 0) atomicAdd(db1 + i, adb1);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = adb2[e];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 8) atomicAdd(db2 + lane * 4 + e, v);
  }
}

