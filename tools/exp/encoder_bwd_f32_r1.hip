// ROUND-1 encoder backward (fp32 MFMA phases 1-2, on-the-fly split phase 3), kept for A/B reference only: NOT built
// into libunreal_hip.so.  Drop-in section for unreal_amd/csrc/encoder.hip (inside its anonymous namespace).

// ------------------------------------------------------------------------------------------------
// Backward.  Input d2 = dL/d(conv2 pre-activation) [N][81][32] (ReLU mask already applied by the
// producer), c1 = saved conv1 activation [N][400][16], the uint8 frame.  Produces dW2, dW1 (register
// accumulators across all frames of the workgroup, flushed once with float atomics), db2, db1.
//   (1) dW2[(ky,kx,c)][n] += sum_pos c1[2oy+ky][2ox+kx][c] * d2[pos][n]            M=256 N=32 K=81
//   (2) d1[2a+pa][2b+pb][c] = sum_{da,db,n} d2[a-da][b-db][n] * W2[pa+2da][pb+2db][c][n]
//       per output parity (pa,pb): M=100 N=16 K=128; masked by c1 > 0 and written in place of c1
//   (3) dW1[(ky,kx,cin)][c] += scale * sum_pos u8[4oy+ky][4ox+kx][cin] * d1[pos][c]  M=192 N=16 K=400
// Pipeline per frame: the NEXT frame's uint8 image, c1 and d2 are fetched into registers while phase (3)
// of the current frame runs and are stored to LDS right after it.
// ------------------------------------------------------------------------------------------------
constexpr int C1_V = (C1_POS * 4 + 255) / 256;   // f32x4 per thread for one c1 image (7)
constexpr int D2_V = (C2_POS * 8 + 255) / 256;   // f32x4 per thread for one d2 image (3)

// conv2 dgrad for NT output tiles (16 positions each) of ONE output parity: one weight fragment read
// (w2t, LDS) feeds all NT tiles.
template <int NT>
__device__ __forceinline__ float dgrad_tiles(const float* d2, float* c1, const float* w2t, int par, int mt0, int i,
                                             int q) {
  f32x4 acc[NT];
  int a[NT], b[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int m = min((mt0 + t) * 16 + i, 99);
    a[t] = m / 10;
    b[t] = m % 10;
  }
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const int da = dd >> 1, db = dd & 1;
    int row[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int y = a[t] - da, x = b[t] - db;
      row[t] = ((y >= 0 && y < 9 && x >= 0 && x < 9) ? y * 9 + x : C2_POS) * D2_LD + 4 * q;
    }
#pragma unroll
    for (int cch = 0; cch < 2; ++cch) {
      f32x4 av[NT];
      const f32x4 bw = *reinterpret_cast<const f32x4*>(w2t + ((((par * 4 + dd) * 2 + cch) * 4 + q) * 16 + i) * 4);
#pragma unroll
      for (int t = 0; t < NT; ++t) av[t] = *reinterpret_cast<const f32x4*>(d2 + row[t] + 16 * cch);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t][s], bw[s], acc[t]);
    }
  }
  float db1 = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = (mt0 + t) * 16 + 4 * q + r;
      if (m < 100) {
        const int idx = ((2 * (m / 10) + (par >> 1)) * 20 + 2 * (m % 10) + (par & 1)) * C1_LD + i;
        const float g = c1[idx] > 0.f ? acc[t][r] : 0.f;
        c1[idx] = g;
        db1 += g;
      }
    }
  return db1;
}

#ifdef UNREAL_ABLATE
__device__ unsigned long long g_stamp_sum[16];
#define STAMP(k)                                                                   \
  do {                                                                             \
    if (PHASES == 7 && blockIdx.x == 3 && threadIdx.x == 0) {                      \
      unsigned long long t_ = __builtin_amdgcn_s_memtime();                        \
      g_stamp_sum[k] += t_ - t_prev_;                                              \
      t_prev_ = t_;                                                                \
    }                                                                              \
  } while (0)
#else
#define STAMP(k)
#endif

template <int PHASES>   // bit 0/1/2 = phase (1)/(2)/(3); 7 in the product, other values only for ablation timing
__global__ __launch_bounds__(512) void encoder_bwd_kernel(int N, const uint8_t* __restrict__ frames,
                                                          const int* __restrict__ frame_idx, float scale,
                                                          const float* __restrict__ W2,
                                                          const float* __restrict__ c1_saved,
                                                          const float* __restrict__ d2_in, float* __restrict__ dW1,
                                                          float* __restrict__ db1, float* __restrict__ dW2,
                                                          float* __restrict__ db2) {
  constexpr int GRP_BYTES = FR_LDS + C1_LDS * 4 + D2_ROWS * D2_LD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * GRP_BYTES + W2_ELEMS * 4];
  const int grp = threadIdx.x >> 8, gtid = threadIdx.x & 255;
  const int lane = threadIdx.x & 63, gw = gtid >> 6;
  const int i = lane & 15, q = lane >> 4;
  uint8_t* fr = smem + grp * GRP_BYTES;
  float* c1 = reinterpret_cast<float*>(fr + FR_LDS);
  float* d2 = c1 + C1_LDS;
  float* w2t = reinterpret_cast<float*>(smem + 2 * GRP_BYTES);
  // W2 for dgrad as [par(4)][dd(4)][cch(2)][q(4)][c(16)][s(4)]:
  //   value W2[((pa+2da)*4 + (pb+2db))*16 + c][n], n = 16cch + 4q + s
  for (int e = threadIdx.x; e < W2_ELEMS; e += 512) {
    int s = e & 3, c = (e >> 2) & 15, qq = (e >> 6) & 3, cch = (e >> 8) & 1, dd = (e >> 9) & 3, par = e >> 11;
    int ky = (par >> 1) + 2 * (dd >> 1), kx = (par & 1) + 2 * (dd & 1);
    w2t[e] = W2[((ky * 4 + kx) * 16 + c) * 32 + 16 * cch + 4 * qq + s];
  }
  for (int e = gtid; e < 3 * D2_LD; e += 256) d2[C2_POS * D2_LD + e] = 0.f;   // zero rows 81..83

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
  // prologue: stage c1 / d2 of this group's first frame
  {
    const int n0 = blockIdx.x * 2 + grp;
    if (n0 < N) {
      frame_load(frames + (size_t)frame_idx[n0] * FRAME_BYTES, gtid, pfr);
      frame_store(fr, gtid, pfr);
      const f32x4* cs = reinterpret_cast<const f32x4*>(c1_saved + (size_t)n0 * (C1_POS * C1_CH));
      for (int id = gtid; id < C1_POS * 4; id += 256)
        *reinterpret_cast<f32x4*>(c1 + (id >> 2) * C1_LD + (id & 3) * 4) = cs[id];
      const f32x4* ds = reinterpret_cast<const f32x4*>(d2_in + (size_t)n0 * F2_DIM);
      for (int id = gtid; id < C2_POS * 8; id += 256) {
        f32x4 v = ds[id];
        *reinterpret_cast<f32x4*>(d2 + (id >> 3) * D2_LD + (id & 7) * 4) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) adb2[e] += v[e];
      }
    }
  }

#ifdef UNREAL_ABLATE
  unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
  for (int base = blockIdx.x * 2; base < N; base += stride) {
    const int n = base + grp;
    const bool valid = n < N;
    const int nn = n + stride;
    const bool has_next = nn < N;
    STAMP(8);
    __syncthreads();  // [S0] c1 / d2 of frame n staged
    STAMP(0);
    if (valid && (PHASES & 1)) {
      // (1) conv2 wgrad; operands of step st+1 are read from LDS before the MFMAs of step st issue.
      // The step loop is kept rolled (3 steps per trip): fully unrolled, the 21 lane-dependent address sets are
      // loop-invariant across frames, get hoisted and spill.
      float av[4], bv0, bv1;
      {
        const int ab = (gw * 20) * C1_LD + i + q * 2 * C1_LD;          // kp = q < 9
        bv0 = d2[q * D2_LD + i]; bv1 = d2[q * D2_LD + 16 + i];
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) av[kx] = c1[ab + kx * C1_LD];
      }
#pragma unroll 1
      for (int sb = 0; sb < 21; sb += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int st = sb + u;
          float an[4], bn0, bn1;
          {
            const int kpb = min(4 * (st + 1) + q, D2_ROWS - 1);   // rows 81..83 of d2 are zero; st = 20 reads a dummy
            const int kp = min(kpb, C2_POS - 1);
            const int ab = ((2 * (kp / 9) + gw) * 20 + 2 * (kp % 9)) * C1_LD + i;
            bn0 = d2[kpb * D2_LD + i]; bn1 = d2[kpb * D2_LD + 16 + i];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) an[kx] = c1[ab + kx * C1_LD];
          }
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) {
            aw2[kx][0] = MFMA16(av[kx], bv0, aw2[kx][0]);
            aw2[kx][1] = MFMA16(av[kx], bv1, aw2[kx][1]);
          }
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) av[kx] = an[kx];
          bv0 = bn0; bv1 = bn1;
        }
      }
    }
    STAMP(1);
    __syncthreads();  // [S1] all reads of c1 done before the in-place dgrad overwrite
    STAMP(2);

    if (valid && (PHASES & 2)) {
      // (2) conv2 dgrad: wave gw owns output parity gw; its 7 position tiles as 4 + 3 independent accumulators
      adb1 += dgrad_tiles<4>(d2, c1, w2t, gw, 0, i, q);
      adb1 += dgrad_tiles<3>(d2, c1, w2t, gw, 4, i, q);
    }
    STAMP(3);
    __syncthreads();  // [S2] c1 holds d1; frame staged; d2 free
    STAMP(4);

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
      // (3) conv1 wgrad as EXACT-PRODUCT bf16 MFMAs (same argument as the forward conv1): A = uint8 pixels
      // (exact in bf16), B = d1 split into three bf16 terms (hi/mid/lo by truncation, residuals exact), three
      // v_mfma_f32_16x16x32_bf16 per 32-position chunk, fp32 accumulation.  K (positions) is split over the 4 waves:
      // wave gw owns output rows 5gw..5gw+4 (100 positions = 4 chunks of 32 slots, 28 of them zero padding) and ALL
      // 12 row tiles; tile (g,t) holds patch elements m = 64g + 4*row + t so one aligned dword of the frame per
      // position feeds 4 tiles.  Lane (i, q) supplies slots s = 32kc + 8q + j, j = 0..7, of row i (A) / channel i (B).
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
              aw1[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, bpl[tm]), aw1[g][t],
                                                                  0, 0, 0);
          }
        }
      }
    }
    if (has_next) {
#pragma unroll
      for (int c = 0; c < D2_V; ++c) {
        int id = gtid + 256 * c;
        if (id < C2_POS * 8) {
          *reinterpret_cast<f32x4*>(d2 + (id >> 3) * D2_LD + (id & 7) * 4) = pd2[c];
#pragma unroll
          for (int e = 0; e < 4; ++e) adb2[e] += pd2[c][e];
        }
      }
    }
    STAMP(5);
    __syncthreads();  // [S3] phase (3) finished reading c1 (d1)
    STAMP(6);
    if (has_next) {
#pragma unroll
      for (int c = 0; c < C1_V; ++c) {
        int id = gtid + 256 * c;
        if (id < C1_POS * 4) *reinterpret_cast<f32x4*>(c1 + (id >> 2) * C1_LD + (id & 3) * 4) = pc1[c];
      }
      frame_store(fr, gtid, pfr);
    }
  }

  // flush accumulators
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
  // db1: lanes with equal i (channel) across q
  adb1 += __shfl_xor(adb1, 16, 64);
  adb1 += __shfl_xor(adb1, 32, 64);
  if (q == 0) atomicAdd(db1 + i, adb1);
  // db2: threads with equal (gtid % 8) own the same 4 columns
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = adb2[e];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 8) atomicAdd(db2 + lane * 4 + e, v);
  }
}

