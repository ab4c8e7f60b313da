// Pixel-control head on gfx950: the two stride-2 VALID 4x4 transposed convolutions (value + advantage)
// of the dueling Q head, the dueling combine, Q-max, the pixel-control loss and its backward pass.
//
// Reference: /root/reference/model/model.py:411-443 (_pc_deconv_layers), 805-820 (_deconv2d =
// tf.nn.conv2d_transpose, filter [kh,kw,out_c,in_c]), 542-557 (_pc_loss = lambda * l2_loss(pc_r - Q[a])).
//
// The transposed conv is evaluated per OUTPUT PARITY as a small dense GEMM (no scatter, no atomics):
//   out[2a+pa][2b+pb][co] = sum_{da,db in {0,1}} sum_ci in[a-da][b-db][ci] * W[pa+2da][pb+2db][co][ci]
// i.e. 4 GEMMs of M = 100 positions, K = 4*32, N = 1+A channels (padded to the 16-wide MFMA tile).
// Same group/LDS organisation and MFMA operand convention as encoder.hip.
#include "common.h"

namespace {

constexpr int HP_LD = 36, HP_ROWS = 84;   // [81][32] + zero rows; row 81 = padding row
constexpr int DEC_LD_F = 8;               // fwd: floats per output position in LDS
constexpr int DEC_LD_B = 12;              // bwd: floats per position (bank spreading)
constexpr int WD_ELEMS = 4 * 4 * 2 * 4 * 16 * 4;   // 8192

struct PcFwdArgs {
  int N, A;
  const float* hp;          // [N][2592] relu(pc_fc1)
  const float* Wv; const float* bv; const float* Wa; const float* ba;
  // bootstrap mode
  float* qmax;              // [N][400] or null
  // training mode
  const int* action;        // [N]
  const float* target;      // [N][400]
  const int* mask;          // [N]
  float lambda, grad_scale;
  float* d_dec;             // [N][400][1+A]
  float* loss;              // scalar accumulator
};

__device__ __forceinline__ float deconv_w(const PcFwdArgs& p, int ky, int kx, int co, int ci) {
  if (co == 0) return p.Wv[(ky * 4 + kx) * 32 + ci];
  if (co <= p.A) return p.Wa[((ky * 4 + kx) * p.A + (co - 1)) * 32 + ci];
  return 0.f;
}

__global__ __launch_bounds__(512) void pc_deconv_fwd_kernel(PcFwdArgs p) {
  constexpr int GRP_BYTES = HP_ROWS * HP_LD * 4 + PC_CELLS * DEC_LD_F * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * GRP_BYTES + WD_ELEMS * 4];
  const int grp = threadIdx.x >> 8, gtid = threadIdx.x & 255;
  const int lane = threadIdx.x & 63, gw = gtid >> 6;
  const int i = lane & 15, q = lane >> 4;
  float* hp = reinterpret_cast<float*>(smem + grp * GRP_BYTES);
  float* dec = hp + HP_ROWS * HP_LD;
  float* wd = reinterpret_cast<float*>(smem + 2 * GRP_BYTES);
  const int A = p.A, CO = 1 + p.A;

  // weights -> [par][dd][cch][q][co(16)][s], ci = 16cch + 4q + s
  for (int e = threadIdx.x; e < WD_ELEMS; e += 512) {
    int s = e & 3, co = (e >> 2) & 15, qq = (e >> 6) & 3, cch = (e >> 8) & 1, dd = (e >> 9) & 3, par = e >> 11;
    int ky = (par >> 1) + 2 * (dd >> 1), kx = (par & 1) + 2 * (dd & 1);
    wd[e] = deconv_w(p, ky, kx, co, 16 * cch + 4 * qq + s);
  }
  for (int e = gtid; e < 3 * HP_LD; e += 256) hp[C2_POS * HP_LD + e] = 0.f;
  const float bias = (i == 0) ? p.bv[0] : ((i <= A) ? p.ba[i - 1] : 0.f);
  float loss_acc = 0.f;

  const int stride = gridDim.x * 2;
  for (int base = blockIdx.x * 2; base < p.N; base += stride) {
    const int n = base + grp;
    const bool valid = n < p.N;
    __syncthreads();
    if (valid) {
      const f32x4* src = reinterpret_cast<const f32x4*>(p.hp + (size_t)n * F2_DIM);
      for (int id = gtid; id < C2_POS * 8; id += 256)
        *reinterpret_cast<f32x4*>(hp + (id >> 3) * HP_LD + (id & 7) * 4) = src[id];
    }
    __syncthreads();
    if (valid) {
      for (int jb = gw; jb < 28; jb += 8) {
        const int j0 = jb, j1 = min(jb + 4, 27);
        const bool v1 = (jb + 4) < 28;
        const int par0 = j0 / 7, mt0 = j0 % 7, par1 = j1 / 7, mt1 = j1 % 7;
        const int m0 = min(mt0 * 16 + i, 99), m1 = min(mt1 * 16 + i, 99);
        const int a0 = m0 / 10, b0 = m0 % 10, a1 = m1 / 10, b1 = m1 % 10;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          const int da = dd >> 1, db = dd & 1;
          const int y0 = a0 - da, x0 = b0 - db, y1 = a1 - da, x1 = b1 - db;
          const int r0 = (y0 >= 0 && y0 < 9 && x0 >= 0 && x0 < 9) ? y0 * 9 + x0 : C2_POS;
          const int r1 = (y1 >= 0 && y1 < 9 && x1 >= 0 && x1 < 9) ? y1 * 9 + x1 : C2_POS;
#pragma unroll
          for (int cch = 0; cch < 2; ++cch) {
            const f32x4 av0 = *reinterpret_cast<const f32x4*>(hp + r0 * HP_LD + 16 * cch + 4 * q);
            const f32x4 av1 = *reinterpret_cast<const f32x4*>(hp + r1 * HP_LD + 16 * cch + 4 * q);
            const f32x4 bw0 = *reinterpret_cast<const f32x4*>(wd + ((((par0 * 4 + dd) * 2 + cch) * 4 + q) * 16 + i) * 4);
            const f32x4 bw1 = *reinterpret_cast<const f32x4*>(wd + ((((par1 * 4 + dd) * 2 + cch) * 4 + q) * 16 + i) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acc0 = MFMA16(av0[s], bw0[s], acc0);
              acc1 = MFMA16(av1[s], bw1[s], acc1);
            }
          }
        }
        if (i < CO) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int m = mt0 * 16 + 4 * q + r;
            if (m < 100)
              dec[((2 * (m / 10) + (par0 >> 1)) * 20 + 2 * (m % 10) + (par0 & 1)) * DEC_LD_F + i] = acc0[r] + bias;
            m = mt1 * 16 + 4 * q + r;
            if (v1 && m < 100)
              dec[((2 * (m / 10) + (par1 >> 1)) * 20 + 2 * (m % 10) + (par1 & 1)) * DEC_LD_F + i] = acc1[r] + bias;
          }
        }
      }
    }
    __syncthreads();
    if (valid) {
      // dueling combine per output position (pre-activations in dec)
      const int act = p.action ? p.action[n] : 0;
      const bool on = p.mask ? (p.mask[n] != 0) : true;
      for (int pos = gtid; pos < PC_CELLS; pos += 256) {
        const float* d = dec + pos * DEC_LD_F;
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
          const float diff = p.target[(size_t)n * PC_CELLS + pos] - qa;
          const float dq = on ? -p.lambda * diff * p.grad_scale : 0.f;
          if (on) loss_acc += 0.5f * p.lambda * diff * diff;
          float* o = p.d_dec + ((size_t)n * PC_CELLS + pos) * CO;
          o[0] = vpre > 0.f ? dq : 0.f;
          for (int k = 0; k < A; ++k)
            o[1 + k] = d[1 + k] > 0.f ? dq * (((k == act) ? 1.f : 0.f) - 1.f / (float)A) : 0.f;
        }
      }
    }
  }
  if (p.loss) {
    loss_acc = wave_sum(loss_acc);
    if (lane == 0) atomicAdd(p.loss, loss_acc * p.grad_scale);
  }
}

struct PcBwdArgs {
  int N, A;
  const float* hp;          // [N][2592] relu(pc_fc1) (forward activation)
  const float* d_dec;       // [N][400][1+A]
  const float* Wv; const float* Wa;
  float* d_hp;              // [N][2592] gradient wrt pc_fc1 PRE-activation (relu mask applied)
  float* dWv; float* dbv; float* dWa; float* dba;
};

__global__ __launch_bounds__(512) void pc_deconv_bwd_kernel(PcBwdArgs p) {
  constexpr int GRP_BYTES = HP_ROWS * HP_LD * 4 + PC_CELLS * DEC_LD_B * 4;
  constexpr int WB_ELEMS = 4 * 2 * 4 * 32 * 4;   // [ky][c][q][ci(32)][s]; kx = q, co = 4c + s
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * GRP_BYTES + WB_ELEMS * 4];
  const int grp = threadIdx.x >> 8, gtid = threadIdx.x & 255;
  const int lane = threadIdx.x & 63, gw = gtid >> 6;
  const int i = lane & 15, q = lane >> 4;
  float* hp = reinterpret_cast<float*>(smem + grp * GRP_BYTES);
  float* dec = hp + HP_ROWS * HP_LD;
  float* wb = reinterpret_cast<float*>(smem + 2 * GRP_BYTES);
  const int A = p.A, CO = 1 + p.A;

  for (int e = threadIdx.x; e < WB_ELEMS; e += 512) {
    int s = e & 3, ci = (e >> 2) & 31, qq = (e >> 7) & 3, c = (e >> 9) & 1, ky = e >> 10;
    int co = 4 * c + s;
    float v = 0.f;
    if (co == 0) v = p.Wv[(ky * 4 + qq) * 32 + ci];
    else if (co <= A) v = p.Wa[((ky * 4 + qq) * A + (co - 1)) * 32 + ci];
    wb[e] = v;
  }
  for (int e = gtid; e < 3 * HP_LD; e += 256) hp[C2_POS * HP_LD + e] = 0.f;
  // zero the co >= CO padding columns of dec once (never rewritten)
  for (int e = gtid; e < PC_CELLS * DEC_LD_B; e += 256) dec[e] = 0.f;

  f32x4 aw[2][2];          // dW tiles: ky = gw, kxh = 0..1 (kx = 2kxh + (i>>3), co = i&7), nt = 0..1
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) aw[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float adb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) adb[k] = 0.f;
  const int nt = gw & 1;
  const int akx = i >> 3, aco = i & 7;

  const int stride = gridDim.x * 2;
  for (int base = blockIdx.x * 2; base < p.N; base += stride) {
    const int n = base + grp;
    const bool valid = n < p.N;
    __syncthreads();
    if (valid) {
      const f32x4* src = reinterpret_cast<const f32x4*>(p.hp + (size_t)n * F2_DIM);
      for (int id = gtid; id < C2_POS * 8; id += 256)
        *reinterpret_cast<f32x4*>(hp + (id >> 3) * HP_LD + (id & 7) * 4) = src[id];
      const float* dsrc = p.d_dec + (size_t)n * PC_CELLS * CO;
      for (int pos = gtid; pos < PC_CELLS; pos += 256)
        for (int k = 0; k < CO; ++k) {
          float v = dsrc[pos * CO + k];
          dec[pos * DEC_LD_B + k] = v;
          adb[k] += v;
        }
    }
    __syncthreads();
    if (valid) {
      // (a) dgrad: d_hp[pos][ci] = sum_{ky,kx,co} d_dec[2y+ky][2x+kx][co] W[ky][kx][co][ci]
      {
        f32x4 acc[3];
        int abase[3];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
          int pos = min(((gw >> 1) + 2 * jj) * 16 + i, C2_POS - 1);
          abase[jj] = ((2 * (pos / 9)) * 20 + 2 * (pos % 9) + q) * DEC_LD_B;
        }
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const f32x4 bw = *reinterpret_cast<const f32x4*>(wb + ((((ky * 2 + c) * 4 + q) * 32) + nt * 16 + i) * 4);
            f32x4 av[3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
              av[jj] = *reinterpret_cast<const f32x4*>(dec + abase[jj] + ky * 20 * DEC_LD_B + 4 * c);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int jj = 0; jj < 3; ++jj) acc[jj] = MFMA16(av[jj][s], bw[s], acc[jj]);
          }
        float* out = p.d_hp + (size_t)n * F2_DIM;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int pos = ((gw >> 1) + 2 * jj) * 16 + 4 * q + r;
            if (pos < C2_POS) {
              int ci = nt * 16 + i;
              out[pos * 32 + ci] = hp[pos * HP_LD + ci] > 0.f ? acc[jj][r] : 0.f;
            }
          }
      }
      // (b) wgrad: dW[(ky,kx,co)][ci] += sum_pos d_dec[2y+ky][2x+kx][co] * hp[pos][ci]
      for (int st = 0; st < 21; ++st) {
        const int kp = min(4 * st + q, C2_POS - 1);
        const int kpb = 4 * st + q;                      // rows 81..83 of hp are zero
        const float b0 = hp[kpb * HP_LD + i], b1 = hp[kpb * HP_LD + 16 + i];
        const int ab = ((2 * (kp / 9) + gw) * 20 + 2 * (kp % 9) + akx) * DEC_LD_B + aco;
#pragma unroll
        for (int kxh = 0; kxh < 2; ++kxh) {
          const float av = dec[ab + 2 * kxh * DEC_LD_B];
          aw[kxh][0] = MFMA16(av, b0, aw[kxh][0]);
          aw[kxh][1] = MFMA16(av, b1, aw[kxh][1]);
        }
      }
    }
  }

#pragma unroll
  for (int kxh = 0; kxh < 2; ++kxh)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 4 * q + r;                 // row in the 16-row tile: kx = 2kxh + (row>>3), co = row&7
        int kx = 2 * kxh + (row >> 3), co = row & 7, ci = t * 16 + i;
        float v = aw[kxh][t][r];
        if (co == 0) atomicAdd(p.dWv + (gw * 4 + kx) * 32 + ci, v);
        else if (co <= A) atomicAdd(p.dWa + ((gw * 4 + kx) * A + (co - 1)) * 32 + ci, v);
      }
  for (int k = 0; k < CO; ++k) {
    float v = wave_sum(adb[k]);
    if (lane == 0) {
      if (k == 0) atomicAdd(p.dbv, v);
      else atomicAdd(p.dba + (k - 1), v);
    }
  }
}

}  // namespace

extern "C" {

int unreal_pc_deconv_fwd(int N, int A, const float* hp, const float* Wv, const float* bv, const float* Wa,
                         const float* ba, float* qmax, const int* action, const float* target, const int* mask,
                         float lambda, float grad_scale, float* d_dec, float* loss, void* stream) {
  if (N <= 0 || A <= 0 || A > 7 || !hp || !Wv || !bv || !Wa || !ba) return UNREAL_EINVAL;
  if (!qmax && !d_dec) return UNREAL_EINVAL;
  if (d_dec && (!action || !target || !mask || !loss)) return UNREAL_EINVAL;
  PcFwdArgs p{N, A, hp, Wv, bv, Wa, ba, qmax, action, target, mask, lambda, grad_scale, d_dec, d_dec ? loss : nullptr};
  int blocks = min((N + 1) / 2, 256);
  hipLaunchKernelGGL(pc_deconv_fwd_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, p);
  return unreal_launch_status();
}

int unreal_pc_deconv_bwd(int N, int A, const float* hp, const float* d_dec, const float* Wv, const float* Wa,
                         float* d_hp, float* dWv, float* dbv, float* dWa, float* dba, void* stream) {
  if (N <= 0 || A <= 0 || A > 7 || !hp || !d_dec || !Wv || !Wa || !d_hp || !dWv || !dbv || !dWa || !dba)
    return UNREAL_EINVAL;
  PcBwdArgs p{N, A, hp, d_dec, Wv, Wa, d_hp, dWv, dbv, dWa, dba};
  int blocks = min((N + 1) / 2, 256);
  hipLaunchKernelGGL(pc_deconv_bwd_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, p);
  return unreal_launch_status();
}

}  // extern "C"
