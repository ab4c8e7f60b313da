// Fused conv encoder for 84x84x3 uint8 frames on gfx950: conv1 8x8 s4 (3->16) + ReLU -> conv2 4x4 s2
// (16->32) + ReLU, forward and backward, as implicit-GEMM on v_mfma_f32_16x16x4_f32.
//
// Replaces tf.nn.conv2d + bias + relu of /root/reference/model/model.py:281-289,786-787 and their
// tf.gradients (train/rmsprop_applier.py:100-105).  Weights are in TF HWIO layout:
//   W1[(ky*8+kx)*3+cin][16], W2[(ky*4+kx)*16+cin][32]; outputs NHWC (flatten = model.py:331).
//
// One frame is processed by a GROUP of 4 waves; a 512-thread workgroup holds two groups that share
// the LDS copy of W2.  Per group the uint8 frame (21 KB) and the conv1 activation (400 x 16 fp32,
// row stride 20 floats so that ds_read_b128 im2col reads spread over the banks) live in LDS; the
// frame is read from HBM exactly once, conv1 output never leaves the CU on the inference path.
// MFMA operand convention (16x16x4): lane l = (i = l&15, q = l>>4) supplies A[i][k] and B[k][i]
// for ONE k per instruction; K is walked in a permuted order chosen so that one 32/128-bit LDS
// read yields the operands of 4 consecutive MFMAs (k = f(chunk(q), s), s = 0..3).
#include "common.h"

namespace {

constexpr int FR_LDS = 21184;            // FRAME_BYTES rounded up to 64
constexpr int C1_LD = 20;                // floats per conv1 position in LDS
constexpr int C1_LDS = C1_POS * C1_LD;   // 8000 floats
constexpr int W2_ELEMS = 256 * 32;
constexpr int D2_LD = 36;
constexpr int D2_ROWS = 84;              // 81 + 3 zero rows (row 81 doubles as the "padding" row)

__device__ __forceinline__ void cvt4(uint32_t w, float (&f)[4]) {
  f[0] = (float)(w & 0xffu);
  f[1] = (float)((w >> 8) & 0xffu);
  f[2] = (float)((w >> 16) & 0xffu);
  f[3] = (float)(w >> 24);
}

// conv1 weights into registers: reg (p,c,s) holds W1[k][j], k = (2p + cc/6)*24 + 4*(cc%6) + s, cc = 3q + c
__device__ __forceinline__ void load_w1_regs(const float* __restrict__ W1, int q, int j, float (&w)[4][3][4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int cc = 3 * q + c;
      int kb = (2 * p + cc / 6) * 24 + 4 * (cc % 6);
#pragma unroll
      for (int s = 0; s < 4; ++s) w[p][c][s] = W1[(kb + s) * 16 + j];
    }
}

__device__ __forceinline__ void copy_frame_to_lds(const uint8_t* __restrict__ src, uint8_t* dst, int gtid) {
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  for (int c = gtid; c < FRAME_BYTES / 16; c += 256) d4[c] = s4[c];
}

// conv1 for one group: fr (uint8 LDS) -> c1 (fp32 LDS, post-ReLU)
__device__ __forceinline__ void conv1_tiles(const uint8_t* fr, float* c1, const float (&w1)[4][3][4], float bias_j,
                                            float scale, int gw, int i, int q) {
  for (int tt = gw; tt < 25; tt += 8) {
    const int ta = tt, tb = min(tt + 4, 24);
    const bool vb = (tt + 4) < 25;
    const int pa = ta * 16 + i, pb = tb * 16 + i;
    const int ba = (4 * (pa / 20)) * FRAME_ROW_BYTES + 12 * (pa % 20);
    const int bb = (4 * (pb / 20)) * FRAME_ROW_BYTES + 12 * (pb % 20);
    f32x4 acca = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int cc = 3 * q + c;
        const int off = (2 * p + cc / 6) * FRAME_ROW_BYTES + 4 * (cc % 6);
        float fa[4], fb[4];
        cvt4(*reinterpret_cast<const uint32_t*>(fr + ba + off), fa);
        cvt4(*reinterpret_cast<const uint32_t*>(fr + bb + off), fb);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acca = MFMA16(fa[s], w1[p][c][s], acca);
          accb = MFMA16(fb[s], w1[p][c][s], accb);
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c1[(ta * 16 + 4 * q + r) * C1_LD + i] = fmaxf(scale * acca[r] + bias_j, 0.f);
      if (vb) c1[(tb * 16 + 4 * q + r) * C1_LD + i] = fmaxf(scale * accb[r] + bias_j, 0.f);
    }
  }
}

__global__ __launch_bounds__(512) void encoder_fwd_kernel(int N, const uint8_t* __restrict__ frames,
                                                          const int* __restrict__ frame_idx, float scale,
                                                          const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ W2, const float* __restrict__ b2,
                                                          float* __restrict__ c1_out, float* __restrict__ f2_out) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (FR_LDS + C1_LDS * 4) + W2_ELEMS * 4];
  const int grp = threadIdx.x >> 8, gtid = threadIdx.x & 255;
  const int lane = threadIdx.x & 63, gw = gtid >> 6;
  const int i = lane & 15, q = lane >> 4;
  uint8_t* fr = smem + grp * (FR_LDS + C1_LDS * 4);
  float* c1 = reinterpret_cast<float*>(fr + FR_LDS);
  float* w2s = reinterpret_cast<float*>(smem + 2 * (FR_LDS + C1_LDS * 4));

  // W2 -> LDS as [(ky*4+c)][q][n(32)][s]: k = ky*64 + q*16 + 4c + s
  for (int e = threadIdx.x; e < W2_ELEMS; e += 512) {
    int s = e & 3, n = (e >> 2) & 31, qq = (e >> 7) & 3, kc = e >> 9;
    int k = (kc >> 2) * 64 + qq * 16 + 4 * (kc & 3) + s;
    w2s[e] = W2[k * 32 + n];
  }
  float w1[4][3][4];
  load_w1_regs(W1, q, i, w1);
  const float bias1 = b1[i];
  const int nt = gw & 1;                       // this wave's conv2 N-tile
  const float bias2 = b2[nt * 16 + i];

  const int stride = gridDim.x * 2;
  int n = blockIdx.x * 2 + grp;
  if (n < N) copy_frame_to_lds(frames + (size_t)frame_idx[n] * FRAME_BYTES, fr, gtid);
  __syncthreads();

  for (int base = blockIdx.x * 2; base < N; base += stride) {
    n = base + grp;
    const bool valid = n < N;
    if (valid) conv1_tiles(fr, c1, w1, bias1, scale, gw, i, q);
    __syncthreads();  // c1 complete; fr free

    const int nn = n + stride;
    uint4 pre[6];
    const bool has_next = nn < N;
    if (has_next) {
      const uint4* s4 = reinterpret_cast<const uint4*>(frames + (size_t)frame_idx[nn] * FRAME_BYTES);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        int id = gtid + 256 * c;
        if (id < FRAME_BYTES / 16) pre[c] = s4[id];
      }
    }

    if (valid) {
      // conv2: this wave owns M-tiles mt = (gw>>1) + 2*jj (jj = 0..2) of N-tile nt
      f32x4 acc[3];
      int abase[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int pos2 = min(((gw >> 1) + 2 * jj) * 16 + i, C2_POS - 1);
        abase[jj] = ((2 * (pos2 / 9)) * 20 + 2 * (pos2 % 9) + q) * C1_LD;
      }
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 bw = *reinterpret_cast<const f32x4*>(w2s + ((((ky * 4 + c) * 4 + q) * 32) + nt * 16 + i) * 4);
          f32x4 av[3];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj)
            av[jj] = *reinterpret_cast<const f32x4*>(c1 + abase[jj] + ky * 20 * C1_LD + 4 * c);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) acc[jj] = MFMA16(av[jj][s], bw[s], acc[jj]);
        }
      float* f2 = f2_out + (size_t)n * F2_DIM;
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int pos2 = ((gw >> 1) + 2 * jj) * 16 + 4 * q + r;
          if (pos2 < C2_POS) f2[pos2 * 32 + nt * 16 + i] = fmaxf(acc[jj][r] + bias2, 0.f);
        }
      if (c1_out) {
        f32x4* dst = reinterpret_cast<f32x4*>(c1_out + (size_t)n * (C1_POS * C1_CH));
        for (int id = gtid; id < C1_POS * 4; id += 256)
          dst[id] = *reinterpret_cast<const f32x4*>(c1 + (id >> 2) * C1_LD + (id & 3) * 4);
      }
    }
    if (has_next) {
      uint4* d4 = reinterpret_cast<uint4*>(fr);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        int id = gtid + 256 * c;
        if (id < FRAME_BYTES / 16) d4[id] = pre[c];
      }
    }
    __syncthreads();  // next frame staged; c1 free
  }
}

// ------------------------------------------------------------------------------------------------
// Backward.  Input d2 = dL/d(conv2 pre-activation) [N][81][32] (ReLU mask already applied by the
// producer), c1 = saved conv1 activation [N][400][16], the uint8 frame.  Produces dW2, dW1 (register
// accumulators across all frames of the workgroup, flushed once with float atomics), db2, db1.
//   (1) dW2[(ky,kx,c)][n] += sum_pos c1[2oy+ky][2ox+kx][c] * d2[pos][n]            M=256 N=32 K=81
//   (2) d1[2a+pa][2b+pb][c] = sum_{da,db,n} d2[a-da][b-db][n] * W2[pa+2da][pb+2db][c][n]
//       per output parity (pa,pb): M=100 N=16 K=128; masked by c1 > 0 and written in place of c1
//   (3) dW1[(ky,kx,cin)][c] += scale * sum_pos u8[4oy+ky][4ox+kx][cin] * d1[pos][c]  M=192 N=16 K=400
// ------------------------------------------------------------------------------------------------
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
  f32x4 aw1[3];         // dW1 tiles: mt = 3*gw + 0..2
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) aw2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 3; ++a) aw1[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float adb2[4] = {0.f, 0.f, 0.f, 0.f};   // n = (gtid % 8) * 4 + e
  float adb1 = 0.f;                       // channel i

  int off1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int m = (3 * gw + a) * 16 + i;
    off1[a] = (m / 24) * FRAME_ROW_BYTES + (m % 24);
  }

  const int stride = gridDim.x * 2;
  for (int base = blockIdx.x * 2; base < N; base += stride) {
    const int n = base + grp;
    const bool valid = n < N;
    __syncthreads();  // previous frame fully consumed
    if (valid) {
      copy_frame_to_lds(frames + (size_t)frame_idx[n] * FRAME_BYTES, fr, gtid);
      const f32x4* cs = reinterpret_cast<const f32x4*>(c1_saved + (size_t)n * (C1_POS * C1_CH));
      for (int id = gtid; id < C1_POS * 4; id += 256)
        *reinterpret_cast<f32x4*>(c1 + (id >> 2) * C1_LD + (id & 3) * 4) = cs[id];
      const f32x4* ds = reinterpret_cast<const f32x4*>(d2_in + (size_t)n * F2_DIM);
      for (int id = gtid; id < C2_POS * 8; id += 256) {
        f32x4 v = ds[id];
        *reinterpret_cast<f32x4*>(d2 + (id >> 3) * D2_LD + (id & 7) * 4) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) adb2[e] += v[e];
      }
    }
    __syncthreads();

    if (valid) {
      // (1) conv2 wgrad
      for (int st = 0; st < 21; ++st) {
        const int kp = min(4 * st + q, C2_POS - 1);
        const int kpb = 4 * st + q;                       // rows 81..83 of d2 are zero
        const int ab = ((2 * (kp / 9) + gw) * 20 + 2 * (kp % 9)) * C1_LD + i;
        const float b0 = d2[kpb * D2_LD + i], b1v = d2[kpb * D2_LD + 16 + i];
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const float av = c1[ab + kx * C1_LD];
          aw2[kx][0] = MFMA16(av, b0, aw2[kx][0]);
          aw2[kx][1] = MFMA16(av, b1v, aw2[kx][1]);
        }
      }
    }
    __syncthreads();  // all reads of c1 done before the in-place dgrad overwrite

    if (valid) {
      // (2) conv2 dgrad, 28 jobs (par, mt) over 4 waves, two at a time
      for (int jb = gw; jb < 28; jb += 8) {
        const int j0 = jb, j1 = min(jb + 4, 27);
        const bool v1 = (jb + 4) < 28;
        const int par0 = j0 / 7, mt0 = j0 % 7, par1 = j1 / 7, mt1 = j1 % 7;
        const int m0 = min(mt0 * 16 + i, 99), m1 = min(mt1 * 16 + i, 99);
        const int a0 = m0 / 10, bb0 = m0 % 10, a1 = m1 / 10, bb1 = m1 % 10;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          const int da = dd >> 1, db = dd & 1;
          const int y0 = a0 - da, x0 = bb0 - db, y1 = a1 - da, x1 = bb1 - db;
          const int r0 = (y0 >= 0 && y0 < 9 && x0 >= 0 && x0 < 9) ? y0 * 9 + x0 : C2_POS;
          const int r1 = (y1 >= 0 && y1 < 9 && x1 >= 0 && x1 < 9) ? y1 * 9 + x1 : C2_POS;
#pragma unroll
          for (int cch = 0; cch < 2; ++cch) {
            const f32x4 av0 = *reinterpret_cast<const f32x4*>(d2 + r0 * D2_LD + 16 * cch + 4 * q);
            const f32x4 av1 = *reinterpret_cast<const f32x4*>(d2 + r1 * D2_LD + 16 * cch + 4 * q);
            const f32x4 bw0 = *reinterpret_cast<const f32x4*>(w2t + ((((par0 * 4 + dd) * 2 + cch) * 4 + q) * 16 + i) * 4);
            const f32x4 bw1 = *reinterpret_cast<const f32x4*>(w2t + ((((par1 * 4 + dd) * 2 + cch) * 4 + q) * 16 + i) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acc0 = MFMA16(av0[s], bw0[s], acc0);
              acc1 = MFMA16(av1[s], bw1[s], acc1);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int m = mt0 * 16 + 4 * q + r;
          if (m < 100) {
            int idx = ((2 * (m / 10) + (par0 >> 1)) * 20 + 2 * (m % 10) + (par0 & 1)) * C1_LD + i;
            float g = c1[idx] > 0.f ? acc0[r] : 0.f;
            c1[idx] = g;
            adb1 += g;
          }
          m = mt1 * 16 + 4 * q + r;
          if (v1 && m < 100) {
            int idx = ((2 * (m / 10) + (par1 >> 1)) * 20 + 2 * (m % 10) + (par1 & 1)) * C1_LD + i;
            float g = c1[idx] > 0.f ? acc1[r] : 0.f;
            c1[idx] = g;
            adb1 += g;
          }
        }
      }
    }
    __syncthreads();  // c1 now holds d1

    if (valid) {
      // (3) conv1 wgrad
      for (int st = 0; st < 100; ++st) {
        const int kp = 4 * st + q;
        const int fb = (4 * (kp / 20)) * FRAME_ROW_BYTES + 12 * (kp % 20);
        const float bv = c1[kp * C1_LD + i];
#pragma unroll
        for (int a = 0; a < 3; ++a) aw1[a] = MFMA16((float)fr[fb + off1[a]], bv, aw1[a]);
      }
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
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(dW1 + ((3 * gw + a) * 16 + 4 * q + r) * 16 + i, scale * aw1[a][r]);
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

}  // namespace

extern "C" {

int unreal_encoder_fwd(int N, const uint8_t* frames, const int* frame_idx, float frame_scale, const float* W1,
                       const float* b1, const float* W2, const float* b2, float* c1_out, float* f2_out,
                       void* stream) {
  if (N <= 0 || !frames || !frame_idx || !W1 || !b1 || !W2 || !b2 || !f2_out) return UNREAL_EINVAL;
  int blocks = min((N + 1) / 2, 256);
  hipLaunchKernelGGL(encoder_fwd_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, N, frames, frame_idx,
                     frame_scale, W1, b1, W2, b2, c1_out, f2_out);
  return unreal_launch_status();
}

int unreal_encoder_bwd(int N, const uint8_t* frames, const int* frame_idx, float frame_scale, const float* W2,
                       const float* c1_saved, const float* d2, float* dW1, float* db1, float* dW2, float* db2,
                       void* stream) {
  if (N <= 0 || !frames || !frame_idx || !W2 || !c1_saved || !d2 || !dW1 || !db1 || !dW2 || !db2)
    return UNREAL_EINVAL;
  int blocks = min((N + 1) / 2, 256);
  hipLaunchKernelGGL(encoder_bwd_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, N, frames, frame_idx,
                     frame_scale, W2, c1_saved, d2, dW1, db1, dW2, db2);
  return unreal_launch_status();
}

}  // extern "C"
