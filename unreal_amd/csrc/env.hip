// Batched maze environment (K1), pixel change (K2) and the replay-ring write, for gfx950.
//
// Reference behaviour restated (never copied) from
//   /root/reference/environment/maze_environment.py:18-128  (map, _move, _get_current_image, process)
//   /root/reference/environment/environment.py:88-102       (_calc_pixel_change)
//   /root/reference/train/experience.py:63-93                (add_frame; successive-terminal discard)
//   /root/reference/train/trainer.py:194-205,264-296         (who resets what, and when)
//
// Layout: every actor owns H1 = history_size + 1 physical ring slots.  The observation the policy
// is about to act on already lives in slot (count % H1) -- the env renders s_{t+1} straight into
// the slot that the NEXT add_frame will commit, so a frame is written to HBM exactly once and is
// never copied.  The extra slot keeps the oldest committed frame intact while it is still
// sample-able.  One workgroup (256 threads) per actor: 21,168 B of frame are written with
// 16 B/lane coalesced stores; the kernel is a pure HBM-write stream.
#include "common.h"
#include "policy_row.h"

namespace {

constexpr const char* kMap =
    "--+---G"
    "--+-+++"
    "S-+---+"
    "--+++--"
    "--+-+--"
    "--+----"
    "-----++";

constexpr uint64_t wall_mask() {
  uint64_t m = 0;
  for (int i = 0; i < 49; ++i)
    if (kMap[i] == '+') m |= (1ull << i);
  return m;
}
constexpr int find_cell(char c) {
  for (int i = 0; i < 49; ++i)
    if (kMap[i] == c) return i;
  return -1;
}
constexpr uint64_t kWalls = wall_mask();
constexpr int kStartX = find_cell('S') % 7, kStartY = find_cell('S') / 7;
constexpr int kGoalX = find_cell('G') % 7, kGoalY = find_cell('G') / 7;
static_assert(kStartX == 0 && kStartY == 2 && kGoalX == 6 && kGoalY == 0, "maze constants");

__device__ __forceinline__ bool is_wall(int x, int y) { return (kWalls >> (y * 7 + x)) & 1ull; }

// one byte of the rendered frame: ch0 = wall block, ch1 = agent block, ch2 = 0
__device__ __forceinline__ uint32_t render_byte(int idx, int ax, int ay) {
  int row = idx / FRAME_ROW_BYTES;
  int c3 = idx - row * FRAME_ROW_BYTES;
  int col = c3 / 3;
  int ch = c3 - col * 3;
  int cy = row / 12, cx = col / 12;
  uint32_t wall = (uint32_t)((kWalls >> (cy * 7 + cx)) & 1ull);
  uint32_t agent = (cx == ax && cy == ay) ? 1u : 0u;
  return ch == 0 ? wall : (ch == 1 ? agent : 0u);
}

// The frame is the constant wall image plus the 12x12 agent block (channel 1).  A workgroup builds the wall image
// ONCE in LDS (the per-byte index arithmetic below is ~250 VALU per 16 bytes: rendering every frame from scratch made
// the step kernel VALU-bound at 1.6 TB/s) and streams it out for each of its actors; the agent block -- 12 rows of 36
// contiguous bytes (0,1,0)x12, never on a wall cell -- is patched in afterwards as 9 dwords per row.
constexpr int kActorsPerGroup = 8;
#ifndef MAZE_APG_BIG
#define MAZE_APG_BIG 8          // actors per workgroup of the step kernel at > 1024 actors (A/B: tools/exp/maze_apg_ab.py)
#endif
constexpr int kStepActorsBig = MAZE_APG_BIG;
#ifndef MAZE_APG_TINY
#define MAZE_APG_TINY 1         // actors per workgroup at <= 64 actors (a small update's rollout step: one actor per workgroup)
#endif
constexpr int kStepActorsTiny = MAZE_APG_TINY;

__device__ __forceinline__ void build_wall_image(uint4* img) {
  for (int c = threadIdx.x; c < FRAME_BYTES / 16; c += blockDim.x) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int base = c * 16 + k * 4;
      w[k] = render_byte(base, -1, -1) | (render_byte(base + 1, -1, -1) << 8) | (render_byte(base + 2, -1, -1) << 16) |
             (render_byte(base + 3, -1, -1) << 24);
    }
    img[c] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// caller: __syncthreads() between the two (same workgroup, same addresses: the barrier orders the stores)
__device__ __forceinline__ void render_walls(uint8_t* dst, const uint4* img) {
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  for (int c = threadIdx.x; c < FRAME_BYTES / 16; c += blockDim.x) d4[c] = img[c];
}
__device__ __forceinline__ void render_agent(uint8_t* dst, int ax, int ay) {
  if (threadIdx.x < 12 * 9) {
    const int r = threadIdx.x / 9, w = threadIdx.x % 9;            // dword w of the 36-byte run: bytes 4w..4w+3
    // (0,1,0) repeated; byte 4w is channel w mod 3.  Selects, not a table: the compiler put `pat[3]` into constant memory, and
    // the global load it then needed in here came with s_waitcnt vmcnt(0) -- which also waits for the wave's wall stores
    // of this frame: a memory round trip per actor, and the next actor's barrier made the whole workgroup wait for it
    const int ph = w % 3;
    const uint32_t pat = ph == 0 ? 0x00000100u : (ph == 1 ? 0x01000001u : 0x00010000u);
    reinterpret_cast<uint32_t*>(dst + (12 * ay + r) * FRAME_ROW_BYTES + 36 * ax)[w] = pat;
  }
}

// pixels of the 12x12 agent block at cell (cx,cy) inside pixel-change cell (i,j):
// rows 4i+2..4i+5, cols 4j+2..4j+5 of the full frame (the [2:-2] crop, then 4x4 blocks)
__device__ __forceinline__ int overlap1(int cell, int k) {
  int lo = max(12 * cell, 4 * k + 2), hi = min(12 * cell + 11, 4 * k + 5);
  return max(0, hi - lo + 1);
}

struct StepArgs {
  int B, H1;
  const int* actions;
  const int* active;
  int* pos;
  int* last_action;
  float* last_reward;
  int* count;
  uint8_t* frames;
  float* r_reward;
  int* r_action;
  int* r_terminal;
  int* r_last_action;
  float* r_last_reward;
  float* r_pc;
  float* out_reward;
  int* out_terminal;
  float* episode_reward;
  float* score_out;
  int* score_valid;
  int reset_on_terminal;
  int track_score;
  // rollout bookkeeping fused into the step (unreal_maze_rollout_step; all null / 0 for the plain step):
  int* active_rw;        // in: actor still inside its rollout; out: cleared at its terminal (trainer.py:279-296 `break`)
  int* active_log_t;     // active flag of this step (row mask of the losses)
  int* n_steps;          // += 1 per step taken
  int* terminal_end;     // set at the terminal
  int* next_idx;         // nullable: ring index of the NEXT observation ((idx_base + b) * H1 + slot), also for idle actors
  float* next_lar;       // nullable: [B][lar_ld] rows of the next step's LSTM input: one-hot last action | last reward
  int lar_ld, lar_col0, A;
  int idx_base;          // index of this launch's first actor in the ring next_idx is meant for (a half-batch of a ring)
  // fused policy step (unreal_maze_policy_rollout_step; pol_x null: the actions are given): the actors' feature rows ->
  // pi, V and the drawn action, computed by the workgroup that then steps those actors (one launch less per rollout step)
  const float* pol_x; int pol_ldx;
  const float* Wp; const float* bp; const float* Wv; const float* bv;
  const double* pol_u;
  float* pi_out; float* v_out; int* act_out;
};

// APG actors per workgroup: 8 when the batch fills the chip (the wall image is built once per workgroup: ~2.5 us of VALU),
// 2 for small batches (grouped updates: 512 actors per launch), where 8 actors in a row per workgroup were 20 of the
// launch's 23 us and most CUs had no workgroup at all, 1 at <= 64 actors (an 8-actor update: 8 workgroups instead of 4)
template <int APG>
__global__ __launch_bounds__(256) void maze_step_kernel(StepArgs p) {
  __shared__ uint4 wall_img[FRAME_BYTES / 16];
  // the workgroup's actors' scalar state, fetched by one thread per actor while the wall image is built: read inside the
  // per-actor loop, each actor would start with two dependent global round trips (state, then the previous slot's terminal
  // flag) that nothing overlaps -- 8 actors x ~2 us of a 38 us launch
  __shared__ int s_flag[APG], s_x[APG], s_y[APG], s_a[APG],
      s_cnt[APG], s_la[APG], s_prev[APG], s_ns[APG];
  __shared__ float s_lr[APG], s_ep[APG];
  if (p.pol_x) {       // (workgroup-uniform) policy of this workgroup's actors: wave w takes actors w, w + 4, ...
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wave; k < APG; k += 4) {
      const int b = blockIdx.x * APG + k;
      if (b >= p.B) break;
      const int act = policy_row<4>(p.pol_x + (size_t)b * p.pol_ldx, p.Wp, p.bp, p.Wv, p.bv, p.pol_u + b,
                                    p.pi_out + (size_t)b * 4, p.v_out + b, lane);
      if (lane == 0) { s_a[k] = act; p.act_out[b] = act; }
    }
  }
  if (threadIdx.x < APG) {
    const int k = threadIdx.x, b = blockIdx.x * APG + k;
    if (b < p.B) {
      const int cnt = p.count[b];
      s_flag[k] = p.active_rw ? p.active_rw[b] : (p.active ? p.active[b] : 1);
      s_x[k] = p.pos[2 * b]; s_y[k] = p.pos[2 * b + 1];
      if (!p.pol_x) s_a[k] = p.actions[b];
      s_cnt[k] = cnt;
      s_la[k] = p.last_action[b];
      s_lr[k] = p.last_reward[b];
      s_ep[k] = p.track_score ? p.episode_reward[b] : 0.f;
      s_ns[k] = p.active_rw ? p.n_steps[b] : 0;      // (read here: a load inside the actor loop stalls thread 0's wave -- and,
                                                     // through the loop's barrier, the workgroup -- for a memory round trip per actor)
      s_prev[k] = cnt > 0 ? p.r_terminal[(size_t)b * p.H1 + (cnt - 1) % p.H1] : 0;
    }
  }
  build_wall_image(wall_img);
  __syncthreads();
  const int H1 = p.H1;
  for (int k = 0; k < APG; ++k) {
    const int b = blockIdx.x * APG + k;
    if (b >= p.B) break;
    const int act_flag = s_flag[k];
    if (p.active_rw && threadIdx.x == 0) p.active_log_t[b] = act_flag;
    if (!act_flag) {
      // idle for the rest of the rollout: its observation and last action / reward stay what they are
      if (threadIdx.x == 0) {
        if (p.next_idx) p.next_idx[b] = (p.idx_base + b) * H1 + s_cnt[k] % H1;
        if (p.next_lar) {
          float* row = p.next_lar + (size_t)b * p.lar_ld + p.lar_col0;
          const int la0 = s_la[k];
          for (int e = 0; e < p.A; ++e) row[e] = (e == la0) ? 1.f : 0.f;
          row[p.A] = s_lr[k];
        }
      }
      continue;
    }
    const int x = s_x[k], y = s_y[k];
    const int a = s_a[k];
    const int cnt = s_cnt[k];
    const int la = s_la[k];
    const float lr = s_lr[k];
    const int slot = cnt % H1;
    const int prev_term = s_prev[k];
    float ep = s_ep[k];

    // _move (maze_environment.py:76-91)
    int dx = (a == 3) - (a == 2), dy = (a == 1) - (a == 0);
    int nx = x + dx, ny = y + dy;
    bool clamped = nx < 0 || nx > 6 || ny < 0 || ny > 6;
    nx = min(max(nx, 0), 6);
    ny = min(max(ny, 0), 6);
    bool hit_wall = is_wall(nx, ny);
    if (hit_wall) { nx = x; ny = y; }
    const bool hit = clamped || hit_wall;
    const bool terminal = (nx == kGoalX && ny == kGoalY);
    const float reward = terminal ? 1.f : (hit ? -1.f : 0.f);

    const size_t base = (size_t)b * H1 + slot;
    // pixel change between render(nx,ny) and render(x,y): only the two agent blocks differ (ch 1)
    const bool moved = (nx != x) || (ny != y);
    for (int c = threadIdx.x; c < PC_CELLS; c += blockDim.x) {
      int i = c / 20, j = c - i * 20;
      int s = 0;
      if (moved) s = overlap1(y, i) * overlap1(x, j) + overlap1(ny, i) * overlap1(nx, j);
      p.r_pc[base * PC_CELLS + c] = (float)s / 48.0f;
    }

    const bool discard = terminal && cnt > 0 && prev_term;  // experience.py:64-67
    const int ncnt = discard ? cnt : cnt + 1;
    const bool reset = terminal && p.reset_on_terminal;
    const int rx = reset ? kStartX : nx, ry = reset ? kStartY : ny;
    const int nslot = ncnt % H1;
    uint8_t* dst = p.frames + ((size_t)b * H1 + nslot) * FRAME_BYTES;
    render_walls(dst, wall_img);
    __syncthreads();  // every thread has read the actor's state; wall stores precede the agent patch
    render_agent(dst, rx, ry);

    if (threadIdx.x == 0) {
      p.r_reward[base] = reward;
      p.r_action[base] = a;
      p.r_terminal[base] = terminal ? 1 : 0;
      p.r_last_action[base] = la;
      p.r_last_reward[base] = lr;
      p.pos[2 * b] = rx;
      p.pos[2 * b + 1] = ry;
      p.count[b] = ncnt;
      p.last_action[b] = reset ? 0 : a;
      p.last_reward[b] = reset ? 0.f : reward;
      if (p.out_reward) p.out_reward[b] = reward;
      if (p.out_terminal) p.out_terminal[b] = terminal ? 1 : 0;
      if (p.track_score) {
        ep += reward;
        if (terminal) {
          p.score_out[b] = ep;
          p.score_valid[b] = 1;
          ep = 0.f;
        }
        p.episode_reward[b] = ep;
      }
      if (p.active_rw) {
        p.n_steps[b] = s_ns[k] + 1;
        if (terminal) {
          p.active_rw[b] = 0;
          p.terminal_end[b] = 1;
        }
      }
      if (p.next_idx) p.next_idx[b] = (p.idx_base + b) * H1 + nslot;
      if (p.next_lar) {
        float* row = p.next_lar + (size_t)b * p.lar_ld + p.lar_col0;
        const int la1 = reset ? 0 : a;
        for (int e = 0; e < p.A; ++e) row[e] = (e == la1) ? 1.f : 0.f;
        row[p.A] = reset ? 0.f : reward;
      }
    }
  }
}

__global__ __launch_bounds__(256) void maze_reset_kernel(int B, int H1, const int* mask, int* pos,
                                                         int* last_action, float* last_reward,
                                                         const int* count, uint8_t* frames) {
  __shared__ uint4 wall_img[FRAME_BYTES / 16];
  build_wall_image(wall_img);
  __syncthreads();
  for (int k = 0; k < kActorsPerGroup; ++k) {
    const int b = blockIdx.x * kActorsPerGroup + k;
    if (b >= B) break;
    if (mask && !mask[b]) continue;
    const int slot = count[b] % H1;
    uint8_t* dst = frames + ((size_t)b * H1 + slot) * FRAME_BYTES;
    render_walls(dst, wall_img);
    __syncthreads();
    render_agent(dst, kStartX, kStartY);
    if (threadIdx.x == 0) {
      pos[2 * b] = kStartX;
      pos[2 * b + 1] = kStartY;
      last_action[b] = 0;
      last_reward[b] = 0.f;
    }
  }
}

// Generic pixel change between two stored uint8 frames (host-fed environments; also the
// cross-check of the analytic maze form): out = sum_{4x4x3} |new - old| / denom.
__global__ __launch_bounds__(256) void pixel_change_u8_kernel(int N, const uint8_t* frames,
                                                              const int* idx_new, const int* idx_old,
                                                              float denom, float* out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N * PC_CELLS) return;
  int n = g / PC_CELLS, c = g - n * PC_CELLS;
  int i = c / 20, j = c - i * 20;
  const uint8_t* fa = frames + (size_t)idx_new[n] * FRAME_BYTES;
  const uint8_t* fb = frames + (size_t)idx_old[n] * FRAME_BYTES;
  int s = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int off = (4 * i + 2 + r) * FRAME_ROW_BYTES + (4 * j + 2) * 3;
#pragma unroll
    for (int k = 0; k < 12; ++k) s += abs((int)fa[off + k] - (int)fb[off + k]);
  }
  out[g] = (float)s / denom;
}

// ---- host-fed environments (the DeepMind-Lab frame/reward contract) -----------------------
// The simulator runs on host cores; one uint8 frame per actor is staged in HBM (`staged`, after an H2D copy
// from pinned memory) and this kernel does what MazeEnvironment's kernel does for the maze: pixel change
// against the stored previous frame, commit of the ring slot, copy of the new observation into the next slot.
// Contract restated from /root/reference/environment/lab_environment.py:104-119: on a terminal step the
// state is the PREVIOUS state (pixel change 0) and the staged frame is the post-reset observation the
// trainer's env.reset() obtains (train/trainer.py:201-202, 292).  `clip_reward` applies the upstream replay's
// np.clip(reward, -1, 1) to the STORED reward / last_reward (train/experience_lab_ver.py:14,18); the
// environment's own last_reward stays raw.
struct HostFedArgs {
  int B, H1;
  const uint8_t* staged;
  const int* actions;
  const float* rewards;
  const int* terminals;
  const int* active;
  int* last_action;
  float* last_reward;
  int* count;
  uint8_t* frames;
  float* r_reward;
  int* r_action;
  int* r_terminal;
  int* r_last_action;
  float* r_last_reward;
  float* r_pc;
  float* out_reward;
  int* out_terminal;
  float* episode_reward;
  float* score_out;
  int* score_valid;
  int reset_on_terminal, track_score, clip_reward;
  float pc_denom;
};

__device__ __forceinline__ float clip1(float r, int on) { return on ? fminf(fmaxf(r, -1.f), 1.f) : r; }

__global__ __launch_bounds__(256) void hostfed_step_kernel(HostFedArgs p) {
  const int b = blockIdx.x;
  if (p.active && !p.active[b]) return;
  const int H1 = p.H1;
  const int a = p.actions[b];
  const float reward = p.rewards[b];
  const bool terminal = p.terminals[b] != 0;
  const int cnt = p.count[b];
  const int la = p.last_action[b];
  const float lr = p.last_reward[b];
  const int slot = cnt % H1;
  const int prev_term = cnt > 0 ? p.r_terminal[(size_t)b * H1 + (cnt - 1) % H1] : 0;
  float ep = p.track_score ? p.episode_reward[b] : 0.f;
  __syncthreads();
  const size_t base = (size_t)b * H1 + slot;
  const uint8_t* fnew = p.staged + (size_t)b * FRAME_BYTES;
  const uint8_t* fold = p.frames + base * FRAME_BYTES;
  for (int c = threadIdx.x; c < PC_CELLS; c += blockDim.x) {
    int s = 0;
    if (!terminal) {
      const int i = c / 20, j = c - i * 20;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = (4 * i + 2 + r) * FRAME_ROW_BYTES + (4 * j + 2) * 3;
#pragma unroll
        for (int k = 0; k < 12; ++k) s += abs((int)fnew[off + k] - (int)fold[off + k]);
      }
    }
    p.r_pc[base * PC_CELLS + c] = (float)s / p.pc_denom;
  }
  const bool discard = terminal && cnt > 0 && prev_term;
  const int ncnt = discard ? cnt : cnt + 1;
  const bool reset = terminal && p.reset_on_terminal;
  const int nslot = ncnt % H1;
  __syncthreads();   // pixel change has read the old frame before a discard could overwrite the same slot
  {
    const uint4* s4 = reinterpret_cast<const uint4*>(fnew);
    uint4* d4 = reinterpret_cast<uint4*>(p.frames + ((size_t)b * H1 + nslot) * FRAME_BYTES);
    for (int c = threadIdx.x; c < FRAME_BYTES / 16; c += blockDim.x) d4[c] = s4[c];
  }
  if (threadIdx.x == 0) {
    p.r_reward[base] = clip1(reward, p.clip_reward);
    p.r_action[base] = a;
    p.r_terminal[base] = terminal ? 1 : 0;
    p.r_last_action[base] = la;
    p.r_last_reward[base] = clip1(lr, p.clip_reward);
    p.count[b] = ncnt;
    p.last_action[b] = reset ? 0 : a;
    p.last_reward[b] = reset ? 0.f : reward;
    if (p.out_reward) p.out_reward[b] = reward;
    if (p.out_terminal) p.out_terminal[b] = terminal ? 1 : 0;
    if (p.track_score) {
      ep += reward;
      if (terminal) {
        p.score_out[b] = ep;
        p.score_valid[b] = 1;
        ep = 0.f;
      }
      p.episode_reward[b] = ep;
    }
  }
}

// env.reset() for host-fed actors: the staged post-reset observation becomes the current observation
__global__ __launch_bounds__(256) void hostfed_reset_kernel(int B, int H1, const int* mask, const uint8_t* staged,
                                                            int* last_action, float* last_reward, const int* count,
                                                            uint8_t* frames) {
  const int b = blockIdx.x;
  if (mask && !mask[b]) return;
  const int slot = count[b] % H1;
  const uint4* s4 = reinterpret_cast<const uint4*>(staged + (size_t)b * FRAME_BYTES);
  uint4* d4 = reinterpret_cast<uint4*>(frames + ((size_t)b * H1 + slot) * FRAME_BYTES);
  for (int c = threadIdx.x; c < FRAME_BYTES / 16; c += blockDim.x) d4[c] = s4[c];
  if (threadIdx.x == 0) {
    last_action[b] = 0;
    last_reward[b] = 0.f;
  }
}

// ---- Philox4x32-10 counter RNG: key = seed, counter = (index, stream) ------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t index, uint64_t stream, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// Element i of a rank's draw is element (i / row_len) * row_stride + col0 + i % row_len of the GLOBAL draw: with
// row_len = this rank's actors, row_stride = all actors and col0 = the rank's first actor, a job sharded over W ranks
// draws exactly what one process holding every actor would (row_len = row_stride = n, col0 = 0: a plain stream).
__device__ __forceinline__ uint64_t philox_index(int i, int row_len, int row_stride, int col0) {
  return (uint64_t)(i / row_len) * (uint64_t)row_stride + (uint64_t)col0 + (uint64_t)(i % row_len);
}

__global__ void philox_uniform_kernel(uint64_t seed, uint64_t stream, int n, int row_len, int row_stride, int col0,
                                      double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t r[4];
  philox4x32_10(seed, philox_index(i, row_len, row_stride, col0), stream, r);
  // 53-bit uniform in [0,1), same construction as numpy's random_sample
  out[i] = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) / 9007199254740992.0;
}

__global__ void philox_randint_kernel(uint64_t seed, uint64_t stream, int n, int row_len, int row_stride, int col0,
                                      int high, int* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t r[4];
  philox4x32_10(seed, philox_index(i, row_len, row_stride, col0), stream, r);
  out[i] = (int)(((uint64_t)r[0] * (uint64_t)high) >> 32);
}

}  // namespace

extern "C" {

int unreal_maze_step(int B, int H1, const int* actions, const int* active, int* pos, int* last_action,
                     float* last_reward, int* count, uint8_t* frames, float* r_reward, int* r_action,
                     int* r_terminal, int* r_last_action, float* r_last_reward, float* r_pc,
                     float* out_reward, int* out_terminal, float* episode_reward, float* score_out,
                     int* score_valid, int reset_on_terminal, int track_score, void* stream) {
  if (B <= 0 || H1 < 2 || !actions || !pos || !count || !frames) return UNREAL_EINVAL;
  if (track_score && (!episode_reward || !score_out || !score_valid)) return UNREAL_EINVAL;
  StepArgs p{B, H1, actions, active, pos, last_action, last_reward, count, frames, r_reward, r_action,
             r_terminal, r_last_action, r_last_reward, r_pc, out_reward, out_terminal, episode_reward,
             score_out, score_valid, reset_on_terminal, track_score, nullptr, nullptr, nullptr, nullptr, nullptr,
             nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (B <= 64) hipLaunchKernelGGL(maze_step_kernel<kStepActorsTiny>, dim3((B + kStepActorsTiny - 1) / kStepActorsTiny), dim3(256), 0, (hipStream_t)stream, p);
  else if (B <= 1024) hipLaunchKernelGGL(maze_step_kernel<2>, dim3((B + 1) / 2), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(maze_step_kernel<kStepActorsBig>, dim3((B + kStepActorsBig - 1) / kStepActorsBig), dim3(256), 0, (hipStream_t)stream, p);
  return unreal_launch_status();
}

int unreal_maze_rollout_step(int B, int H1, const int* actions, int* pos, int* last_action, float* last_reward, int* count,
                             uint8_t* frames, float* r_reward, int* r_action, int* r_terminal, int* r_last_action,
                             float* r_last_reward, float* r_pc, float* out_reward, int* out_terminal,
                             float* episode_reward, float* score_out, int* score_valid, int* active,
                             int* active_log_t, int* n_steps, int* terminal_end, int* next_idx, float* next_lar,
                             int lar_ld, int lar_col0, int A, int idx_base_actor, void* stream) {
  if (B <= 0 || H1 < 2 || !actions || !pos || !count || !frames || !last_action || !last_reward) return UNREAL_EINVAL;
  if (!episode_reward || !score_out || !score_valid || !active || !active_log_t || !n_steps || !terminal_end)
    return UNREAL_EINVAL;
  if (next_lar && (A <= 0 || lar_col0 < 0 || lar_ld < lar_col0 + A + 1)) return UNREAL_EINVAL;
  if (idx_base_actor < 0) return UNREAL_EINVAL;
  StepArgs p{B, H1, actions, nullptr, pos, last_action, last_reward, count, frames, r_reward, r_action,
             r_terminal, r_last_action, r_last_reward, r_pc, out_reward, out_terminal, episode_reward,
             score_out, score_valid, 1, 1, active, active_log_t, n_steps, terminal_end, next_idx, next_lar, lar_ld,
             lar_col0, A, idx_base_actor, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (B <= 64) hipLaunchKernelGGL(maze_step_kernel<kStepActorsTiny>, dim3((B + kStepActorsTiny - 1) / kStepActorsTiny), dim3(256), 0, (hipStream_t)stream, p);
  else if (B <= 1024) hipLaunchKernelGGL(maze_step_kernel<2>, dim3((B + 1) / 2), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(maze_step_kernel<kStepActorsBig>, dim3((B + kStepActorsBig - 1) / kStepActorsBig), dim3(256), 0, (hipStream_t)stream, p);
  return unreal_launch_status();
}

int unreal_maze_policy_rollout_step(int B, int H1, const float* X, int ldx, const float* Wp, const float* bp, const float* Wv,
                                    const float* bv, const double* u, float* pi_out, float* v_out, int* actions_out, int* pos,
                                    int* last_action, float* last_reward, int* count, uint8_t* frames, float* r_reward,
                                    int* r_action, int* r_terminal, int* r_last_action, float* r_last_reward, float* r_pc,
                                    float* out_reward, int* out_terminal, float* episode_reward, float* score_out,
                                    int* score_valid, int* active, int* active_log_t, int* n_steps, int* terminal_end,
                                    int* next_idx, float* next_lar, int lar_ld, int lar_col0, int A, int idx_base_actor,
                                    void* stream) {
  if (B <= 0 || H1 < 2 || !pos || !count || !frames || !last_action || !last_reward) return UNREAL_EINVAL;
  if (!X || ldx < LSTM_N || !Wp || !bp || !Wv || !bv || !u || !pi_out || !v_out || !actions_out) return UNREAL_EINVAL;
  if (A != 4) return UNREAL_EINVAL;                  // the maze has four actions (maze_environment.py:98-112)
  if (!episode_reward || !score_out || !score_valid || !active || !active_log_t || !n_steps || !terminal_end)
    return UNREAL_EINVAL;
  if (next_lar && (lar_col0 < 0 || lar_ld < lar_col0 + A + 1)) return UNREAL_EINVAL;
  if (idx_base_actor < 0) return UNREAL_EINVAL;
  StepArgs p{B, H1, nullptr, nullptr, pos, last_action, last_reward, count, frames, r_reward, r_action,
             r_terminal, r_last_action, r_last_reward, r_pc, out_reward, out_terminal, episode_reward,
             score_out, score_valid, 1, 1, active, active_log_t, n_steps, terminal_end, next_idx, next_lar, lar_ld,
             lar_col0, A, idx_base_actor, X, ldx, Wp, bp, Wv, bv, u, pi_out, v_out, actions_out};
  if (B <= 64) hipLaunchKernelGGL(maze_step_kernel<kStepActorsTiny>, dim3((B + kStepActorsTiny - 1) / kStepActorsTiny), dim3(256), 0, (hipStream_t)stream, p);
  else if (B <= 1024) hipLaunchKernelGGL(maze_step_kernel<2>, dim3((B + 1) / 2), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(maze_step_kernel<kStepActorsBig>, dim3((B + kStepActorsBig - 1) / kStepActorsBig), dim3(256), 0, (hipStream_t)stream, p);
  return unreal_launch_status();
}

int unreal_maze_reset(int B, int H1, const int* mask, int* pos, int* last_action, float* last_reward,
                      const int* count, uint8_t* frames, void* stream) {
  if (B <= 0 || H1 < 2 || !pos || !count || !frames) return UNREAL_EINVAL;
  hipLaunchKernelGGL(maze_reset_kernel, dim3((B + kActorsPerGroup - 1) / kActorsPerGroup), dim3(256), 0, (hipStream_t)stream, B, H1, mask, pos,
                     last_action, last_reward, count, frames);
  return unreal_launch_status();
}

int unreal_hostfed_step(int B, int H1, const uint8_t* staged, const int* actions, const float* rewards,
                        const int* terminals, const int* active, int* last_action, float* last_reward, int* count,
                        uint8_t* frames, float* r_reward, int* r_action, int* r_terminal, int* r_last_action,
                        float* r_last_reward, float* r_pc, float* out_reward, int* out_terminal,
                        float* episode_reward, float* score_out, int* score_valid, int reset_on_terminal,
                        int track_score, int clip_reward, float pc_denom, void* stream) {
  if (B <= 0 || H1 < 2 || !staged || !actions || !rewards || !terminals || !count || !frames || pc_denom <= 0.f)
    return UNREAL_EINVAL;
  if (track_score && (!episode_reward || !score_out || !score_valid)) return UNREAL_EINVAL;
  if ((((uintptr_t)staged) | ((uintptr_t)frames)) & 15) return UNREAL_EINVAL;
  HostFedArgs p{B, H1, staged, actions, rewards, terminals, active, last_action, last_reward, count, frames, r_reward,
                r_action, r_terminal, r_last_action, r_last_reward, r_pc, out_reward, out_terminal, episode_reward,
                score_out, score_valid, reset_on_terminal, track_score, clip_reward, pc_denom};
  hipLaunchKernelGGL(hostfed_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p);
  return unreal_launch_status();
}

int unreal_hostfed_reset(int B, int H1, const int* mask, const uint8_t* staged, int* last_action, float* last_reward,
                         const int* count, uint8_t* frames, void* stream) {
  if (B <= 0 || H1 < 2 || !staged || !count || !frames || !last_action || !last_reward) return UNREAL_EINVAL;
  if ((((uintptr_t)staged) | ((uintptr_t)frames)) & 15) return UNREAL_EINVAL;
  hipLaunchKernelGGL(hostfed_reset_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, B, H1, mask, staged,
                     last_action, last_reward, count, frames);
  return unreal_launch_status();
}

int unreal_pixel_change_u8(int N, const uint8_t* frames, const int* idx_new, const int* idx_old,
                           float denom, float* out, void* stream) {
  if (N <= 0 || !frames || !idx_new || !idx_old || !out || denom <= 0.f) return UNREAL_EINVAL;
  int total = N * PC_CELLS;
  hipLaunchKernelGGL(pixel_change_u8_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     N, frames, idx_new, idx_old, denom, out);
  return unreal_launch_status();
}

int unreal_philox_uniform(uint64_t seed, uint64_t stream_id, int n, int row_len, int row_stride, int col0,
                          double* out, void* stream) {
  if (n <= 0 || !out || row_len <= 0 || row_stride < row_len || col0 < 0 || col0 + row_len > row_stride)
    return UNREAL_EINVAL;
  hipLaunchKernelGGL(philox_uniform_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed,
                     stream_id, n, row_len, row_stride, col0, out);
  return unreal_launch_status();
}

int unreal_philox_randint(uint64_t seed, uint64_t stream_id, int n, int row_len, int row_stride, int col0, int high,
                          int* out, void* stream) {
  if (n <= 0 || high <= 0 || !out || row_len <= 0 || row_stride < row_len || col0 < 0 || col0 + row_len > row_stride)
    return UNREAL_EINVAL;
  hipLaunchKernelGGL(philox_randint_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed,
                     stream_id, n, row_len, row_stride, col0, high, out);
  return unreal_launch_status();
}

}  // extern "C"
