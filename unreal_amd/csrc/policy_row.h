// One rollout step of the policy for ONE feature row, on one wave: pi = softmax(x Wp + bp), v = x Wv + bv, action ~ pi by
// inverse CDF in fp64 (numpy RandomState.choice: searchsorted(cumsum(pi) / sum, u, 'right')) or arg max when u == null.
// Shared by unreal_policy_step (heads.hip) and the fused environment step unreal_maze_policy_rollout_step (env.hip): the
// SAME instructions in the same order, so the two paths agree bit for bit (model/model.py:358-377, trainer.py:147-148).
#pragma once
#include "common.h"

// Call with the whole wave; lane 0 stores pi_row[A], *v_row and returns the action (other lanes return 0).
template <int A>
__device__ __forceinline__ int policy_row(const float* __restrict__ x, const float* __restrict__ Wp,
                                          const float* __restrict__ bp, const float* __restrict__ Wv,
                                          const float* __restrict__ bv, const double* __restrict__ u_row,
                                          float* __restrict__ pi_row, float* __restrict__ v_row, int lane) {
  float acc[A], accv = 0.f;
#pragma unroll
  for (int n = 0; n < A; ++n) acc[n] = 0.f;
  for (int k = lane; k < LSTM_N; k += 64) {
    float xv = x[k];
#pragma unroll
    for (int n = 0; n < A; ++n) acc[n] += xv * Wp[(size_t)k * A + n];
    accv += xv * Wv[k];
  }
#pragma unroll
  for (int n = 0; n < A; ++n) acc[n] = wave_sum(acc[n]);
  accv = wave_sum(accv);
  if (lane != 0) return 0;
  *v_row = accv + bv[0];
  float p[A];
#pragma unroll
  for (int n = 0; n < A; ++n) p[n] = acc[n] + bp[n];
  float m = p[0];
#pragma unroll
  for (int a = 1; a < A; ++a) m = fmaxf(m, p[a]);
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < A; ++a) { float e = expf(p[a] - m); p[a] = e; s += e; }
#pragma unroll
  for (int a = 0; a < A; ++a) { p[a] = p[a] / s; pi_row[a] = p[a]; }
  if (u_row) {
    double tot = 0.0;
#pragma unroll
    for (int a = 0; a < A; ++a) tot += (double)p[a];
    double run = 0.0, uu = *u_row;
    int act = 0;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      run += (double)p[a];
      if (run / tot <= uu) act = a + 1;     // searchsorted(cdf, u, side='right')
    }
    return min(act, A - 1);
  }
  int best = 0;                             // greedy (np.argmax: first maximum), for evaluation
#pragma unroll
  for (int a = 1; a < A; ++a)
    if (p[a] > p[best]) best = a;
  return best;
}
