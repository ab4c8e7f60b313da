// LSTM cell gate math, policy/value/reward-prediction heads, action sampling, A3C / value-replay /
// reward-prediction losses with their gradients, and small reductions.  All HBM-bound elementwise or
// tiny-N work: coalesced row-major access, wavefront shuffles for the reductions (no MFMA).
//
// Reference semantics restated from /root/reference/model/model.py:
//   BasicLSTMCell (TF 1.x, model.py:110,346-351): gates i,j,f,o; c' = c*sig(f+1) + sig(i)*tanh(j); h' = tanh(c')*sig(o)
//   policy / value heads 358-377, softmax 364; base loss 490-516; vr loss 559-566; rp 473-488, 569-576
//   action choice: /root/reference/train/trainer.py:147-148 (numpy RandomState.choice = inverse CDF in fp64)
#include "common.h"
#include "policy_row.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void lstm_gates_fwd_kernel(int rows, const float* __restrict__ pre, const float* __restrict__ bias,
                                      const float* __restrict__ c_prev, float* __restrict__ gates_act,
                                      float* __restrict__ c_out, float* __restrict__ h_out, int ld_h) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * LSTM_N) return;
  int r = g / LSTM_N, u = g - r * LSTM_N;
  const float* p = pre + (size_t)r * 4 * LSTM_N;
  float i = sigmoidf_(p[u] + bias[u]);
  float j = tanhf(p[LSTM_N + u] + bias[LSTM_N + u]);
  float f = sigmoidf_(p[2 * LSTM_N + u] + bias[2 * LSTM_N + u] + 1.0f);
  float o = sigmoidf_(p[3 * LSTM_N + u] + bias[3 * LSTM_N + u]);
  float c = c_prev[g] * f + i * j;
  float h = tanhf(c) * o;
  if (gates_act) {
    float* ga = gates_act + (size_t)r * 4 * LSTM_N;
    ga[u] = i; ga[LSTM_N + u] = j; ga[2 * LSTM_N + u] = f; ga[3 * LSTM_N + u] = o;
  }
  c_out[g] = c;
  h_out[(size_t)r * ld_h + u] = h;
}

__global__ void lstm_gates_bwd_kernel(int rows, const float* __restrict__ dh_above, const float* __restrict__ dh_rec,
                                      float* __restrict__ dc_io, const float* __restrict__ gates_act,
                                      const float* __restrict__ c_prev, const float* __restrict__ c_new,
                                      float* __restrict__ dpre, float* absmax0, float* absmax1) {
  // grid-stride (rows * 256 is a multiple of the block size: no partial waves): a wave commits its maximum ONCE, after all
  // its elements -- one commit per element-wave cost 16 k dependent L2 reads per 4096-row launch (156 us instead of 16)
  float m = 0.f;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < rows * LSTM_N; g += gridDim.x * blockDim.x) {
  int r = g / LSTM_N, u = g - r * LSTM_N;
  const float* ga = gates_act + (size_t)r * 4 * LSTM_N;
  float i = ga[u], j = ga[LSTM_N + u], f = ga[2 * LSTM_N + u], o = ga[3 * LSTM_N + u];
  float dh = dh_above[g] + (dh_rec ? dh_rec[g] : 0.f);
  float tc = tanhf(c_new[g]);
  float dc = dc_io[g] + dh * o * (1.f - tc * tc);
  float* d = dpre + (size_t)r * 4 * LSTM_N;
  const float d0 = dc * j * i * (1.f - i), d1 = dc * i * (1.f - j * j), d2 = dc * c_prev[g] * f * (1.f - f),
              d3 = dh * tc * o * (1.f - o);
  d[u] = d0;
  d[LSTM_N + u] = d1;
  d[2 * LSTM_N + u] = d2;
  d[3 * LSTM_N + u] = d3;
  dc_io[g] = dc * f;
  m = fmaxf(fmaxf(m, fmaxf(fabsf(d0), fabsf(d1))), fmaxf(fabsf(d2), fabsf(d3)));
  }
  // max |d_gates| of this step: the A scale of the products that consume it (unreal_lstm_bptt_step, the fc dgrad).
  // One commit per workgroup: the launch's waves end together and would each issue their (serialised) atomics
  __shared__ float wmx[4];               // blockDim.x = 256
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) {
    const float mm = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
    absmax_commit(absmax0, mm);
    absmax_commit(absmax1, mm);
  }
}

// out[row][n] = X[row][:] . W[:, n] + b[n], NOUT <= 8; one wave per row
template <int NOUT>
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(int rows, int K, const float* __restrict__ X, int ldx,
                                                               const float* __restrict__ W, const float* __restrict__ b,
                                                               float* __restrict__ out, int ldo) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float acc[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
  const float* x = X + (size_t)row * ldx;
  for (int k = lane; k < K; k += 64) {
    float xv = x[k];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] += xv * W[(size_t)k * NOUT + n];
  }
#pragma unroll
  for (int n = 0; n < NOUT; ++n) acc[n] = wave_sum(acc[n]);
  if (lane == 0)
#pragma unroll
    for (int n = 0; n < NOUT; ++n) out[(size_t)row * ldo + n] = acc[n] + b[n];
}

// dX[row][k] (+)= sum_n dO[row][n] W[k][n];  dW[k][n] += sum_rows X[row][k] dO[row][n];  db[n] += sum_rows dO
// thread <-> column k, block <-> chunk of rows; dO rows come through the scalar cache.
template <int NOUT>
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(int rows, int K, int rows_per_block,
                                                               const float* __restrict__ X, int ldx,
                                                               const float* __restrict__ dO, int ldo,
                                                               const float* __restrict__ W, float* __restrict__ dX,
                                                               int lddx, int accumulate_dx, float* __restrict__ dW,
                                                               int dw_sk, int dw_sn, float* __restrict__ db) {
  // dO[row][n] has the same address in every lane (row comes from blockIdx / the loop counter): the compiler reads it
  // through the scalar cache -- no LDS staging, no barrier, no broadcast ds_read per (row, n) in the loop
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * rows_per_block;
  const int nr = min(rows_per_block, rows - r0);
  if (k >= K) return;
  float w[NOUT], acc[NOUT], sdb[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) { w[n] = W ? W[(size_t)k * NOUT + n] : 0.f; acc[n] = 0.f; sdb[n] = 0.f; }
  // four rows per trip with their loads issued together (one load in flight per thread left the kernel at ~1 TB/s);
  // the sums run over the rows in the same order as a one-row loop
  int r = 0;
  for (; r + 4 <= nr; r += 4) {
    float xv[4], old[4], d[4][NOUT];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = X[(size_t)(r0 + r + u) * ldx + k];
    if (dX && accumulate_dx) {
#pragma unroll
      for (int u = 0; u < 4; ++u) old[u] = dX[(size_t)(r0 + r + u) * lddx + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < NOUT; ++n) d[u][n] = dO[(size_t)(r0 + r + u) * ldo + n];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float dx = 0.f;
#pragma unroll
      for (int n = 0; n < NOUT; ++n) {
        acc[n] += xv[u] * d[u][n];
        dx += d[u][n] * w[n];
        sdb[n] += d[u][n];
      }
      if (dX) dX[(size_t)(r0 + r + u) * lddx + k] = accumulate_dx ? (old[u] + dx) : dx;
    }
  }
  for (; r < nr; ++r) {
    const size_t row = (size_t)(r0 + r);
    float xv = X[row * ldx + k];
    float dx = 0.f;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
      const float d = dO[row * ldo + n];
      acc[n] += xv * d;
      dx += d * w[n];
      sdb[n] += d;
    }
    if (dX) {
      float* p = dX + row * lddx + k;
      *p = accumulate_dx ? (*p + dx) : dx;
    }
  }
#pragma unroll
  for (int n = 0; n < NOUT; ++n) atomicAdd(dW + (size_t)k * dw_sk + (size_t)n * dw_sn, acc[n]);
  if (db && k == 0) {                    // every thread holds the same row sums of dO; one of the grid's column 0 adds them
#pragma unroll
    for (int n = 0; n < NOUT; ++n) atomicAdd(db + n, sdb[n]);
  }
}

// softmax over A logits (in place -> pi) + inverse-CDF action draw in fp64 (numpy choice semantics)
__global__ void softmax_sample_kernel(int rows, int A, float* __restrict__ logits_pi, int ld, const double* __restrict__ u,
                                      int* __restrict__ action) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float* p = logits_pi + (size_t)r * ld;
  float m = p[0];
  for (int a = 1; a < A; ++a) m = fmaxf(m, p[a]);
  float s = 0.f;
  for (int a = 0; a < A; ++a) { float e = expf(p[a] - m); p[a] = e; s += e; }
  for (int a = 0; a < A; ++a) p[a] = p[a] / s;
  if (action && u) {
    double tot = 0.0;
    for (int a = 0; a < A; ++a) tot += (double)p[a];
    double run = 0.0, uu = u[r];
    int act = 0;
    for (int a = 0; a < A; ++a) {
      run += (double)p[a];
      if (run / tot <= uu) act = a + 1;     // searchsorted(cdf, u, side='right')
    }
    action[r] = min(act, A - 1);
  } else if (action) {                      // greedy (np.argmax: first maximum), for evaluation
    int best = 0;
    for (int a = 1; a < A; ++a)
      if (p[a] > p[best]) best = a;
    action[r] = best;
  }
}

// One rollout step of the policy: pi = softmax(X Wp + bp), v = X Wv + bv, action ~ pi (inverse CDF in fp64, or
// arg max when u == null) in ONE launch; one wave per row.  The arithmetic is that of linear_small_fwd_kernel<A>,
// linear_small_fwd_kernel<1> and softmax_sample_kernel, operation for operation, so the results are bit-identical to
// the three-kernel path (model/model.py:358-377, train/trainer.py:147-148).
template <int A>
__global__ __launch_bounds__(256) void policy_step_kernel(int rows, const float* __restrict__ X, int ldx,
                                                          const float* __restrict__ Wp, const float* __restrict__ bp,
                                                          const float* __restrict__ Wv, const float* __restrict__ bv,
                                                          const double* __restrict__ u, float* __restrict__ pi_out,
                                                          float* __restrict__ v_out, int* __restrict__ action) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int act = policy_row<A>(X + (size_t)row * ldx, Wp, bp, Wv, bv, u ? u + row : nullptr, pi_out + (size_t)row * A,
                                v_out + row, lane);
  if (lane == 0) action[row] = act;
}

// A3C loss + gradient wrt logits and value.  losses[0..2] += (policy_loss, value_loss, entropy) * loss_scale
__global__ __launch_bounds__(256) void base_loss_grad_kernel(int rows, int A, const float* __restrict__ pi, int ld_pi,
                                                             const float* __restrict__ v, const int* __restrict__ action,
                                                             const float* __restrict__ adv, const float* __restrict__ R,
                                                             const int* __restrict__ active, float beta,
                                                             float grad_scale, float* __restrict__ dlogits,
                                                             float* __restrict__ dv, float* __restrict__ losses) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  float pl = 0.f, vl = 0.f, ent = 0.f;
  if (r < rows) {
    const bool on = active[r] != 0;
    const float* p = pi + (size_t)r * ld_pi;
    float g[8], lp[8];
    float dot = 0.f, H = 0.f;
    const int a_t = action[r];
    const float ad = adv[r];
    for (int a = 0; a < A; ++a) {
      float pa = p[a];
      float pc = fminf(fmaxf(pa, 1e-20f), 1.0f);
      float un = (pa >= 1e-20f && pa <= 1.0f) ? 1.f : 0.f;   // clip_by_value passes gradient inside the range
      lp[a] = logf(pc);
      H -= pa * lp[a];
      // d/dpi of -(log pi_a * adv + beta * H), H = -sum pi log pi
      g[a] = -((a == a_t) ? ad * un / pc : 0.f) + beta * (lp[a] + un);
      dot += pa * g[a];
    }
    for (int a = 0; a < A; ++a) dlogits[(size_t)r * A + a] = on ? grad_scale * p[a] * (g[a] - dot) : 0.f;
    float diff = R[r] - v[r];
    dv[r] = on ? grad_scale * (-0.5f * diff) : 0.f;
    if (on) {
      pl = -(lp[a_t] * ad + beta * H);
      vl = 0.25f * diff * diff;
      ent = H;
    }
  }
  // one atomic per workgroup and loss: the three sums share a cache line and the launch's 1,280 waves end together
  // (3,840 serialised atomics were most of this kernel's 53 us)
  __shared__ float part[3][4];
  pl = wave_sum(pl); vl = wave_sum(vl); ent = wave_sum(ent);
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = pl; part[1][threadIdx.x >> 6] = vl; part[2][threadIdx.x >> 6] = ent; }
  __syncthreads();
  if (threadIdx.x < 3)
    atomicAdd(losses + threadIdx.x, (((part[threadIdx.x][0] + part[threadIdx.x][1]) + part[threadIdx.x][2]) + part[threadIdx.x][3]) * grad_scale);
}

// value-replay loss l2_loss(R - v) = 0.5 * sum (R-v)^2 ; dv = -(R - v)
__global__ __launch_bounds__(256) void vr_loss_grad_kernel(int rows, const float* __restrict__ v, const float* __restrict__ R,
                                                           const int* __restrict__ mask, float grad_scale,
                                                           float* __restrict__ dv, float* __restrict__ loss) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  float l = 0.f;
  if (r < rows) {
    bool on = mask[r] != 0;
    float diff = R[r] - v[r];
    dv[r] = on ? -grad_scale * diff : 0.f;
    if (on) l = 0.5f * diff * diff;
  }
  __shared__ float part[4];              // one atomic per workgroup (see base_loss_grad_kernel)
  l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (((part[0] + part[1]) + part[2]) + part[3]) * grad_scale);
}

// reward prediction: softmax over 3 logits, cross entropy with clipped probabilities
__global__ __launch_bounds__(256) void rp_loss_grad_kernel(int rows, const float* __restrict__ logits, const int* __restrict__ cls,
                                                           float grad_scale, float* __restrict__ prob, float* __restrict__ dlogits,
                                                           float* __restrict__ loss) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  float l = 0.f;
  if (r < rows) {
    const float* z = logits + (size_t)r * 3;
    float m = fmaxf(z[0], fmaxf(z[1], z[2]));
    float e0 = expf(z[0] - m), e1 = expf(z[1] - m), e2 = expf(z[2] - m);
    float s = e0 + e1 + e2;
    float p[3] = {e0 / s, e1 / s, e2 / s};
    int c = cls[r];
    float pc = fminf(fmaxf(p[c], 1e-20f), 1.0f);
    float un = (p[c] >= 1e-20f && p[c] <= 1.0f) ? 1.f : 0.f;
    l = -logf(pc);
    float gc = -un / pc;                    // dL/dp_c
    float dot = p[c] * gc;
    for (int k = 0; k < 3; ++k) {
      if (prob) prob[(size_t)r * 3 + k] = p[k];
      dlogits[(size_t)r * 3 + k] = grad_scale * p[k] * (((k == c) ? gc : 0.f) - dot);
    }
  }
  l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss, l * grad_scale);
}

// out[c] += sum_r X[r][c]; block = 64 columns x 4 row phases over a chunk of rows, one atomic per column
__global__ __launch_bounds__(256) void colsum_kernel(int rows, int cols, int rows_per_block, const float* __restrict__ X,
                                                     int ld, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      s0 += X[(size_t)r * ld + c];
      s1 += X[(size_t)(r + 4) * ld + c];
      s2 += X[(size_t)(r + 8) * ld + c];
      s3 += X[(size_t)(r + 12) * ld + c];
    }
    for (; r < r1; r += 4) s0 += X[(size_t)r * ld + c];
  }
  part[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && c < cols) atomicAdd(out + c, (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]));
}

// d[r][c] = src[r][c] > 0 ? d[r][c] : 0
__global__ void relu_mask_kernel(int rows, int cols, float* __restrict__ d, int ldd, const float* __restrict__ src, int lds) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * cols) return;
  int r = g / cols, c = g - r * cols;
  float* p = d + (size_t)r * ldd + c;
  if (!(src[(size_t)r * lds + c] > 0.f)) *p = 0.f;
}

template <int NOUT>
int launch_small_fwd(int rows, int K, const float* X, int ldx, const float* W, const float* b, float* out, int ldo,
                     hipStream_t st) {
  hipLaunchKernelGGL((linear_small_fwd_kernel<NOUT>), dim3((rows + 3) / 4), dim3(256), 0, st, rows, K, X, ldx, W, b, out,
                     ldo);
  return unreal_launch_status();
}

template <int NOUT>
int launch_small_bwd(int rows, int K, const float* X, int ldx, const float* dO, int ldo, const float* W, float* dX,
                     int lddx, int acc, float* dW, int dw_sk, int dw_sn, float* db, hipStream_t st) {
  if (dw_sk == 0 && dw_sn == 0) { dw_sk = NOUT; dw_sn = 1; }
  const int rpb = 128;
  dim3 grid((K + 255) / 256, (rows + rpb - 1) / rpb);
  hipLaunchKernelGGL((linear_small_bwd_kernel<NOUT>), grid, dim3(256), 0, st, rows, K, rpb, X,
                     ldx, dO, ldo, W, dX, lddx, acc, dW, dw_sk, dw_sn, db);
  return unreal_launch_status();
}

}  // namespace

#define GRID1(n) dim3(((n) + 255) / 256), dim3(256), 0, (hipStream_t)stream

extern "C" {

int unreal_lstm_gates_fwd(int rows, const float* pre, const float* bias, const float* c_prev, float* gates_act,
                          float* c_out, float* h_out, int ld_h, void* stream) {
  if (rows <= 0 || !pre || !bias || !c_prev || !c_out || !h_out || ld_h < LSTM_N) return UNREAL_EINVAL;
  hipLaunchKernelGGL(lstm_gates_fwd_kernel, GRID1(rows * LSTM_N), rows, pre, bias, c_prev, gates_act, c_out, h_out,
                     ld_h);
  return unreal_launch_status();
}

int unreal_lstm_gates_bwd(int rows, const float* dh_above, const float* dh_rec, float* dc_io, const float* gates_act,
                          const float* c_prev, const float* c_new, float* dpre, float* dpre_absmax0, float* dpre_absmax1,
                          void* stream) {
  if (rows <= 0 || !dh_above || !dc_io || !gates_act || !c_prev || !c_new || !dpre) return UNREAL_EINVAL;
  hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(min(rows, 1024)), dim3(256), 0, (hipStream_t)stream, rows, dh_above, dh_rec,
                     dc_io, gates_act, c_prev, c_new, dpre, dpre_absmax0, dpre_absmax1);
  return unreal_launch_status();
}

int unreal_linear_small_fwd(int rows, int K, int NOUT, const float* X, int ldx, const float* W, const float* b,
                            float* out, int ldo, void* stream) {
  if (rows <= 0 || K <= 0 || !X || !W || !b || !out || ldx < K || ldo < NOUT) return UNREAL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (NOUT) {
    case 1: return launch_small_fwd<1>(rows, K, X, ldx, W, b, out, ldo, st);
    case 3: return launch_small_fwd<3>(rows, K, X, ldx, W, b, out, ldo, st);
    case 4: return launch_small_fwd<4>(rows, K, X, ldx, W, b, out, ldo, st);
    case 6: return launch_small_fwd<6>(rows, K, X, ldx, W, b, out, ldo, st);
    default: return UNREAL_EINVAL;
  }
}

int unreal_linear_small_bwd(int rows, int K, int NOUT, const float* X, int ldx, const float* dO, int ldo,
                            const float* W, float* dX, int lddx, int accumulate_dx, float* dW, int dw_stride_k,
                            int dw_stride_n, float* db, void* stream) {
  if (rows <= 0 || K <= 0 || !X || !dO || !dW || ldx < K || ldo < NOUT) return UNREAL_EINVAL;
  if (dX && !W) return UNREAL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (NOUT) {
    case 1: return launch_small_bwd<1>(rows, K, X, ldx, dO, ldo, W, dX, lddx, accumulate_dx, dW, dw_stride_k, dw_stride_n, db, st);
    case 3: return launch_small_bwd<3>(rows, K, X, ldx, dO, ldo, W, dX, lddx, accumulate_dx, dW, dw_stride_k, dw_stride_n, db, st);
    case 4: return launch_small_bwd<4>(rows, K, X, ldx, dO, ldo, W, dX, lddx, accumulate_dx, dW, dw_stride_k, dw_stride_n, db, st);
    case 5: return launch_small_bwd<5>(rows, K, X, ldx, dO, ldo, W, dX, lddx, accumulate_dx, dW, dw_stride_k, dw_stride_n, db, st);
    case 6: return launch_small_bwd<6>(rows, K, X, ldx, dO, ldo, W, dX, lddx, accumulate_dx, dW, dw_stride_k, dw_stride_n, db, st);
    case 7: return launch_small_bwd<7>(rows, K, X, ldx, dO, ldo, W, dX, lddx, accumulate_dx, dW, dw_stride_k, dw_stride_n, db, st);
    default: return UNREAL_EINVAL;
  }
}

int unreal_softmax_sample(int rows, int A, float* logits_pi, int ld, const double* u, int* action, void* stream) {
  if (rows <= 0 || A <= 0 || A > 8 || !logits_pi || ld < A) return UNREAL_EINVAL;
  hipLaunchKernelGGL(softmax_sample_kernel, GRID1(rows), rows, A, logits_pi, ld, u, action);
  return unreal_launch_status();
}

int unreal_policy_step(int rows, int A, const float* X, int ldx, const float* Wp, const float* bp, const float* Wv,
                       const float* bv, const double* u, float* pi_out, float* v_out, int* action, void* stream) {
  if (rows <= 0 || !X || !Wp || !bp || !Wv || !bv || !pi_out || !v_out || !action || ldx < LSTM_N) return UNREAL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
  switch (A) {
    case 3: hipLaunchKernelGGL((policy_step_kernel<3>), grid, block, 0, st, rows, X, ldx, Wp, bp, Wv, bv, u, pi_out, v_out, action); break;
    case 4: hipLaunchKernelGGL((policy_step_kernel<4>), grid, block, 0, st, rows, X, ldx, Wp, bp, Wv, bv, u, pi_out, v_out, action); break;
    case 6: hipLaunchKernelGGL((policy_step_kernel<6>), grid, block, 0, st, rows, X, ldx, Wp, bp, Wv, bv, u, pi_out, v_out, action); break;
    default: return UNREAL_EINVAL;
  }
  return unreal_launch_status();
}

int unreal_base_loss_grad(int rows, int A, const float* pi, int ld_pi, const float* v, const int* action,
                          const float* adv, const float* R, const int* active, float entropy_beta, float grad_scale,
                          float* dlogits, float* dv, float* losses, void* stream) {
  if (rows <= 0 || A <= 0 || A > 8 || !pi || !v || !action || !adv || !R || !active || !dlogits || !dv || !losses)
    return UNREAL_EINVAL;
  hipLaunchKernelGGL(base_loss_grad_kernel, GRID1(rows), rows, A, pi, ld_pi, v, action, adv, R, active, entropy_beta,
                     grad_scale, dlogits, dv, losses);
  return unreal_launch_status();
}

int unreal_vr_loss_grad(int rows, const float* v, const float* R, const int* mask, float grad_scale, float* dv,
                        float* loss, void* stream) {
  if (rows <= 0 || !v || !R || !mask || !dv || !loss) return UNREAL_EINVAL;
  hipLaunchKernelGGL(vr_loss_grad_kernel, GRID1(rows), rows, v, R, mask, grad_scale, dv, loss);
  return unreal_launch_status();
}

int unreal_rp_loss_grad(int rows, const float* logits, const int* cls, float grad_scale, float* prob, float* dlogits,
                        float* loss, void* stream) {
  if (rows <= 0 || !logits || !cls || !dlogits || !loss) return UNREAL_EINVAL;
  hipLaunchKernelGGL(rp_loss_grad_kernel, GRID1(rows), rows, logits, cls, grad_scale, prob, dlogits, loss);
  return unreal_launch_status();
}

int unreal_colsum(int rows, int cols, const float* X, int ld, float* out, void* stream) {
  if (rows <= 0 || cols <= 0 || !X || !out || ld < cols) return UNREAL_EINVAL;
  const int rpb = 1024;
  dim3 grid((cols + 63) / 64, (rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, rows, cols, rpb, X, ld, out);
  return unreal_launch_status();
}

int unreal_relu_mask(int rows, int cols, float* d, int ldd, const float* src, int lds, void* stream) {
  if (rows <= 0 || cols <= 0 || !d || !src) return UNREAL_EINVAL;
  hipLaunchKernelGGL(relu_mask_kernel, GRID1(rows * cols), rows, cols, d, ldd, src, lds);
  return unreal_launch_status();
}

}  // extern "C"
