// Experience-replay sampling on the device ring, n-step / pixel-control / value-replay return scans,
// and the small per-actor bookkeeping kernels of the rollout.
//
// Reference behaviour restated from
//   /root/reference/train/experience.py:100-118 (sample_sequence), 121-153 (sample_rp_sequence)
//   /root/reference/train/trainer.py:298-324 (n-step returns), 354-372 (pc returns), 394-406 (vr returns),
//   427-435 (reward class), experience.py:35-46 (concat_action_and_reward)
//
// Ring convention (see env.hip): absolute frame index i of actor b lives in slot i % H1 with
// H1 = H + 1; the deque of the reference is the absolute range [max(0,count-H), count).
#include "common.h"

namespace {

__global__ void sample_seq_kernel(int B, int H, int H1, int L, const int* start_draw, const int* count,
                                  const int* r_terminal, int* seq_idx, int* seq_len) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int cnt = count[b];
  const int top = max(0, cnt - H);
  int a0 = top + start_draw[b];
  const size_t rb = (size_t)b * H1;
  if (r_terminal[rb + a0 % H1]) a0 += 1;   // experience.py:105-107
  int n = 0, last = 0;
  for (int i = 0; i < L; ++i) {
    int a = min(a0 + i, cnt - 1);
    int idx = (int)(rb + a % H1);
    seq_idx[(size_t)i * B + b] = idx;
    last = idx;
    ++n;
    if (r_terminal[idx]) break;            // stop after the first terminal, inclusive (:113-116)
  }
  for (int i = n; i < L; ++i) seq_idx[(size_t)i * B + b] = last;   // padding rows (masked downstream)
  seq_len[b] = n;
}

// one wave per actor
// bucket rule: mode 0 = this fork (reward > 0 | everything else, experience.py:76-80);
//              mode 1 = upstream / Lab replay (reward != 0 | reward == 0, experience_lab_ver.py:76-80)
__global__ __launch_bounds__(64) void sample_rp_kernel(int B, int H, int H1, const int* coin, const double* u,
                                                       const int* count, const float* r_reward, int* rp_idx,
                                                       int* rp_class, int mode) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int cnt = count[b];
  const int top = max(0, cnt - H);
  const int lo = top + 3;
  const int nw = cnt - lo;
  const size_t rb = (size_t)b * H1;
  int npos = 0;
  for (int c = 0; c < nw; c += 64) {
    int a = lo + c + lane;
    const float rv = (a < cnt) ? r_reward[rb + a % H1] : 0.f;
    bool ispos = (a < cnt) && (mode ? (rv != 0.f) : (rv > 0.f));
    npos += __popcll(__ballot(ispos));
  }
  const int nneg = nw - npos;
  bool from_neg = (coin[b] == 0);
  if (npos == 0) from_neg = true;          // experience.py:130-135
  else if (nneg == 0) from_neg = false;
  const int n = from_neg ? nneg : npos;
  int rank = (int)(u[b] * (double)n);
  rank = min(max(rank, 0), n - 1);
  int run = 0, end = lo;
  for (int c = 0; c < nw; c += 64) {
    int a = lo + c + lane;
    bool in = a < cnt;
    const float rv = in ? r_reward[rb + a % H1] : 0.f;
    bool ispos = in && (mode ? (rv != 0.f) : (rv > 0.f));
    bool member = in && (ispos != from_neg);
    unsigned long long m = __ballot(member);
    int pc = __popcll(m);
    if (run + pc > rank) {
      int k = rank - run;
      int before = __popcll(m & ((1ull << lane) - 1ull));
      unsigned long long hit = __ballot(member && before == k);
      end = lo + c + (__ffsll((long long)hit) - 1);
      break;
    }
    run += pc;
  }
  if (lane < 3) rp_idx[b * 3 + lane] = (int)(rb + (end - 3 + lane) % H1);
  if (lane == 0) {
    float r = r_reward[rb + end % H1];
    rp_class[b] = (r > -1e-10f && r < 1e-10f) ? 0 : (r > 0.f ? 1 : 2);   // trainer.py:427-434
  }
}

__global__ void base_returns_kernel(int B, int T, const float* rewards, const float* values, const int* n_steps,
                                    const float* boot_v, const int* terminal_end, double gamma, float* R_out,
                                    float* adv_out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = n_steps[b];
  double R = terminal_end[b] ? 0.0 : (double)boot_v[b];
  for (int t = T - 1; t >= 0; --t) {
    size_t i = (size_t)t * B + b;
    if (t >= n) { R_out[i] = 0.f; adv_out[i] = 0.f; continue; }
    R = (double)rewards[i] + gamma * R;
    R_out[i] = (float)R;
    adv_out[i] = (float)(R - (double)values[i]);
  }
}

__global__ void vr_returns_kernel(int B, int L, const int* seq_idx, const int* seq_len, const float* r_reward,
                                  const int* r_terminal, const float* boot_v, double gamma, float* R_out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = seq_len[b];
  // bootstrap unless the SECOND-to-last sampled frame is terminal (trainer.py:394-398)
  double R = (n >= 2 && r_terminal[seq_idx[(size_t)(n - 2) * B + b]]) ? 0.0 : (double)boot_v[b];
  for (int t = L - 2; t >= 0; --t) {
    size_t i = (size_t)t * B + b;
    if (t >= n - 1) { R_out[i] = 0.f; continue; }
    R = (double)r_reward[seq_idx[i]] + gamma * R;
    R_out[i] = (float)R;
  }
}

__global__ void pc_returns_kernel(int B, int L, const int* seq_idx, const int* seq_len, const float* r_pc,
                                  const int* r_terminal, const float* boot_qmax, double gamma_pc, float* R_out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * PC_CELLS) return;
  const int b = g / PC_CELLS, c = g - b * PC_CELLS;
  const int n = seq_len[b];
  double R = (n >= 2 && r_terminal[seq_idx[(size_t)(n - 2) * B + b]]) ? 0.0 : (double)boot_qmax[g];
  for (int t = L - 2; t >= 0; --t) {
    size_t row = (size_t)t * B + b;
    if (t >= n - 1) { R_out[row * PC_CELLS + c] = 0.f; continue; }
    R = (double)r_pc[(size_t)seq_idx[row] * PC_CELLS + c] + gamma_pc * R;
    R_out[row * PC_CELLS + c] = (float)R;
  }
}

// xcat[row][256 .. 256+A] = one_hot(last_action) || last_reward ; columns up to ld zeroed
__global__ void lar_fill_kernel(int rows, int A, const int* last_action, const float* last_reward, const int* idx,
                                float* xcat, int ld, int col0, int clip) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = ld - col0;
  if (g >= rows * w) return;
  int r = g / w, k = g - r * w;
  int src = idx ? idx[r] : r;
  float v = 0.f;
  if (k < A) v = (last_action[src] == k) ? 1.f : 0.f;
  else if (k == A) v = clip ? fminf(fmaxf(last_reward[src], -1.f), 1.f) : last_reward[src];
  xcat[(size_t)r * ld + col0 + k] = v;
}

// objective vectors ride with the frames, one [obj] row per ring slot (indoor_environment.py:70-73,113: the
// 'objective' entry of a state).  put: the staged objective of every active actor goes to its CURRENT slot.
__global__ void objective_put_kernel(int B, int H1, int obj, const int* count, const int* active, const float* staged,
                                     float* r_objective) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * obj) return;
  int b = g / obj, j = g - b * obj;
  if (active && !active[b]) return;
  r_objective[((size_t)b * H1 + count[b] % H1) * obj + j] = staged[g];
}

// xcat[row][col0 .. col0+obj) = objective of frame idx[row], or of the slot `slot_offset` before/after it in the same
// actor's ring (trainer.py:300: the bootstrap value is fed the objective of the PREVIOUS frame's state)
__global__ void objective_fill_kernel(int rows, int obj, int H1, const float* r_objective, const int* idx, int slot_offset,
                                      float* xcat, int ld, int col0) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * obj) return;
  int r = g / obj, j = g - r * obj;
  int f = idx[r], b = f / H1, s = f - b * H1;
  s = (s + slot_offset % H1 + H1) % H1;
  xcat[(size_t)r * ld + col0 + j] = r_objective[((size_t)b * H1 + s) * obj + j];
}

// gather a per-frame int attribute of the ring into a dense [rows] array
__global__ void gather_i32_kernel(int rows, const int* src, const int* idx, int* out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < rows) out[g] = src[idx[g]];
}

// rollout bookkeeping after env step t (trainer.py:236-296: the loop breaks at the first terminal)
__global__ void rollout_advance_kernel(int B, const int* terminal_t, int* active, int* active_log_t, int* n_steps,
                                       int* terminal_end) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int a = active[b];
  active_log_t[b] = a;
  if (a) {
    n_steps[b] += 1;
    if (terminal_t[b]) {
      active[b] = 0;
      terminal_end[b] = 1;
    }
  }
}

// row mask for sampled sequences: mask[t][b] = t < len[b] - 1
__global__ void seq_mask_kernel(int B, int T, const int* seq_len, int* mask) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * T) return;
  int t = g / B, b = g - t * B;
  mask[g] = (t < seq_len[b] - 1) ? 1 : 0;
}

// zero the LSTM state of actors whose episode ended in this rollout (trainer.py:293)
__global__ void reset_state_kernel(int B, const int* terminal_end, float* c, float* h) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * LSTM_N) return;
  if (terminal_end[g / LSTM_N]) { c[g] = 0.f; h[g] = 0.f; }
}

// frame index of every actor's current observation: b*H1 + count[b] % H1
__global__ void ring_cur_idx_kernel(int B, int H1, int b0, const int* count, int* out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = (b0 + b) * H1 + count[b] % H1;
}

// bootstrap frame of every sampled sequence: the LAST sampled frame (trainer.py:356-358, 396-398)
__global__ void seq_last_idx_kernel(int B, const int* seq_idx, const int* seq_len, int* out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = seq_idx[(size_t)(seq_len[b] - 1) * B + b];
}

// stats[0] += env steps taken, stats[1] += finished episodes, stats[2] += sum of their scores;
// clears score_valid (one launch per process(): trainer.py:635-636 return value)
__global__ void rollout_stats_kernel(int B, const int* n_steps, int* score_valid, const float* score_out,
                                     double* stats) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0, e = 0.0, sc = 0.0;
  if (b < B) {
    s = (double)n_steps[b];
    if (score_valid[b]) { e = 1.0; sc = (double)score_out[b]; score_valid[b] = 0; }
  }
  s = wave_sum_d(s); e = wave_sum_d(e); sc = wave_sum_d(sc);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(stats + 0, s);
    atomicAdd(stats + 1, e);
    atomicAdd(stats + 2, sc);
  }
}

}  // namespace

#define GRID1(n) dim3(((n) + 255) / 256), dim3(256), 0, (hipStream_t)stream

extern "C" {

int unreal_replay_sample_seq(int B, int H, int H1, int L, const int* start_draw, const int* count,
                             const int* r_terminal, int* seq_idx, int* seq_len, void* stream) {
  if (B <= 0 || L < 2 || H1 != H + 1 || H < L + 2 || !start_draw || !count || !r_terminal || !seq_idx || !seq_len)
    return UNREAL_EINVAL;
  hipLaunchKernelGGL(sample_seq_kernel, GRID1(B), B, H, H1, L, start_draw, count, r_terminal, seq_idx, seq_len);
  return unreal_launch_status();
}

int unreal_replay_sample_rp(int B, int H, int H1, const int* coin, const double* u, const int* count,
                            const float* r_reward, int* rp_idx, int* rp_class, int mode, void* stream) {
  if (B <= 0 || H1 != H + 1 || H < 4 || !coin || !u || !count || !r_reward || !rp_idx || !rp_class)
    return UNREAL_EINVAL;
  hipLaunchKernelGGL(sample_rp_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, B, H, H1, coin, u, count,
                     r_reward, rp_idx, rp_class, mode);
  return unreal_launch_status();
}

int unreal_base_returns(int B, int T, const float* rewards, const float* values, const int* n_steps,
                        const float* boot_v, const int* terminal_end, double gamma, float* R_out, float* adv_out,
                        void* stream) {
  if (B <= 0 || T <= 0) return UNREAL_EINVAL;
  hipLaunchKernelGGL(base_returns_kernel, GRID1(B), B, T, rewards, values, n_steps, boot_v, terminal_end, gamma,
                     R_out, adv_out);
  return unreal_launch_status();
}

int unreal_vr_returns(int B, int L, const int* seq_idx, const int* seq_len, const float* r_reward,
                      const int* r_terminal, const float* boot_v, double gamma, float* R_out, void* stream) {
  if (B <= 0 || L < 2) return UNREAL_EINVAL;
  hipLaunchKernelGGL(vr_returns_kernel, GRID1(B), B, L, seq_idx, seq_len, r_reward, r_terminal, boot_v, gamma,
                     R_out);
  return unreal_launch_status();
}

int unreal_pc_returns(int B, int L, const int* seq_idx, const int* seq_len, const float* r_pc,
                      const int* r_terminal, const float* boot_qmax, double gamma_pc, float* R_out, void* stream) {
  if (B <= 0 || L < 2) return UNREAL_EINVAL;
  hipLaunchKernelGGL(pc_returns_kernel, GRID1(B * PC_CELLS), B, L, seq_idx, seq_len, r_pc, r_terminal, boot_qmax,
                     gamma_pc, R_out);
  return unreal_launch_status();
}

int unreal_lar_fill(int rows, int A, const int* last_action, const float* last_reward, const int* idx, float* xcat,
                    int ld, int col0, int clip_reward, void* stream) {
  if (rows <= 0 || A <= 0 || ld < col0 + A + 1) return UNREAL_EINVAL;
  hipLaunchKernelGGL(lar_fill_kernel, GRID1(rows * (ld - col0)), rows, A, last_action, last_reward, idx, xcat, ld,
                     col0, clip_reward);
  return unreal_launch_status();
}

int unreal_objective_put(int B, int H1, int obj, const int* count, const int* active, const float* staged,
                         float* r_objective, void* stream) {
  if (B <= 0 || H1 < 2 || obj <= 0 || !count || !staged || !r_objective) return UNREAL_EINVAL;
  hipLaunchKernelGGL(objective_put_kernel, GRID1(B * obj), B, H1, obj, count, active, staged, r_objective);
  return unreal_launch_status();
}

int unreal_objective_fill(int rows, int obj, int H1, const float* r_objective, const int* idx, int slot_offset,
                          float* xcat, int ld, int col0, void* stream) {
  if (rows <= 0 || obj <= 0 || H1 < 2 || !r_objective || !idx || !xcat || ld < col0 + obj) return UNREAL_EINVAL;
  hipLaunchKernelGGL(objective_fill_kernel, GRID1(rows * obj), rows, obj, H1, r_objective, idx, slot_offset, xcat, ld,
                     col0);
  return unreal_launch_status();
}

int unreal_gather_i32(int rows, const int* src, const int* idx, int* out, void* stream) {
  if (rows <= 0) return UNREAL_EINVAL;
  hipLaunchKernelGGL(gather_i32_kernel, GRID1(rows), rows, src, idx, out);
  return unreal_launch_status();
}

int unreal_rollout_advance(int B, const int* terminal_t, int* active, int* active_log_t, int* n_steps,
                           int* terminal_end, void* stream) {
  if (B <= 0) return UNREAL_EINVAL;
  hipLaunchKernelGGL(rollout_advance_kernel, GRID1(B), B, terminal_t, active, active_log_t, n_steps, terminal_end);
  return unreal_launch_status();
}

int unreal_seq_mask(int B, int T, const int* seq_len, int* mask, void* stream) {
  if (B <= 0 || T <= 0) return UNREAL_EINVAL;
  hipLaunchKernelGGL(seq_mask_kernel, GRID1(B * T), B, T, seq_len, mask);
  return unreal_launch_status();
}

int unreal_ring_cur_idx(int B, int H1, int b0, const int* count, int* out, void* stream) {
  if (B <= 0 || H1 < 2 || !count || !out) return UNREAL_EINVAL;
  hipLaunchKernelGGL(ring_cur_idx_kernel, GRID1(B), B, H1, b0, count, out);
  return unreal_launch_status();
}

int unreal_seq_last_idx(int B, const int* seq_idx, const int* seq_len, int* out, void* stream) {
  if (B <= 0 || !seq_idx || !seq_len || !out) return UNREAL_EINVAL;
  hipLaunchKernelGGL(seq_last_idx_kernel, GRID1(B), B, seq_idx, seq_len, out);
  return unreal_launch_status();
}

int unreal_rollout_stats(int B, const int* n_steps, int* score_valid, const float* score_out, double* stats,
                         void* stream) {
  if (B <= 0 || !n_steps || !score_valid || !score_out || !stats) return UNREAL_EINVAL;
  hipLaunchKernelGGL(rollout_stats_kernel, GRID1(B), B, n_steps, score_valid, score_out, stats);
  return unreal_launch_status();
}

int unreal_reset_state(int B, const int* terminal_end, float* c, float* h, void* stream) {
  if (B <= 0) return UNREAL_EINVAL;
  hipLaunchKernelGGL(reset_state_kernel, GRID1(B * LSTM_N), B, terminal_end, c, h);
  return unreal_launch_status();
}

}  // extern "C"
