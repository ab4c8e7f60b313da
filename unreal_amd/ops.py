"""Host-side launch wrappers: torch tensors in, libunreal_hip.so kernels out.

PyTorch is used for device memory and streams only; every computation below is a hand-written
gfx950 kernel.  Each wrapper validates dtypes / sizes on the host before launching (a kernel that
indexes out of bounds can take the whole GPU host down).
"""
import torch

from ._lib import lib, ptr, stream

FRAME_BYTES = 21168
PC_CELLS = 400
F2_DIM = 2592
C1_DIM = 6400
GEMM_RELU, GEMM_ACCUM, GEMM_ATOMIC, GEMM_RELU_MASK, GEMM_RELU_BITS = 1, 2, 4, 8, 16
RELU_WORDS = 162                      # uint16 words of ReLU bits per frame (2592 / 16)

_DT = {"f32": torch.float32, "i32": torch.int32, "u8": torch.uint8, "f64": torch.float64, "i16": torch.int16}


def _chk(t, dt, n=None, name="tensor", optional=False):
    if t is None:
        if optional:
            return
        raise ValueError("%s is required" % name)
    if not t.is_cuda:
        raise ValueError("%s must be a CUDA/HIP tensor (no CPU fallback)" % name)
    if t.dtype != _DT[dt]:
        raise ValueError("%s: dtype %s, expected %s" % (name, t.dtype, dt))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if n is not None and t.numel() < n:
        raise ValueError("%s: %d elements, need >= %d" % (name, t.numel(), n))


_TIMER = None     # {"name", "events": [(start, end, units)]} while bench.py times one kernel with HIP events


def _call(name, *a):
    if _TIMER is not None and name == _TIMER["name"]:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        lib().call(name, *a, stream())
        e1.record()
        _TIMER["events"].append((e0, e1, a[0]))
        return
    lib().call(name, *a, stream())


def kernel_timer_start(name):
    """Bracket every launch of kernel entry point `name` with HIP events on the launch stream."""
    global _TIMER
    _TIMER = {"name": name, "events": []}


def kernel_timer_stop():
    """-> {"launches", "ms" (sum of launch durations), "units" (sum of each launch's first size argument)}."""
    global _TIMER
    t, _TIMER = _TIMER, None
    torch.cuda.synchronize()
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in t["events"])
    return {"name": t["name"], "launches": len(t["events"]), "ms": ms, "units": sum(u for _, _, u in t["events"])}


# ---- environment ---------------------------------------------------------------------------------
class Ring(object):
    """Device replay ring + per-actor environment state (layout: include/unreal_hip.h)."""

    def __init__(self, B, H, device, objective_size=0):
        self.B, self.H, self.H1 = B, H, H + 1
        self.objective_size = objective_size
        n = B * self.H1
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
        self.frames = torch.empty(n * FRAME_BYTES, dtype=torch.uint8, device=device)
        self.r_reward = z(n)
        self.r_action = z(n, dt=torch.int32)
        self.r_terminal = z(n, dt=torch.int32)
        self.r_last_action = z(n, dt=torch.int32)
        self.r_last_reward = z(n)
        self.r_pc = torch.empty(n * PC_CELLS, dtype=torch.float32, device=device)
        self.r_objective = z(n * objective_size) if objective_size else None
        self.pos = z(B * 2, dt=torch.int32)
        self.last_action = z(B, dt=torch.int32)
        self.last_reward = z(B)
        self.count = z(B, dt=torch.int32)
        self.episode_reward = z(B)
        self.score_out = z(B)
        self.score_valid = z(B, dt=torch.int32)

        self._cur = z(B, dt=torch.int32)

    def cur_idx(self, out=None, base_actor=0):
        """Frame index of every actor's current observation (slot count % H1).  `base_actor` = index of this (view's)
        first actor in the ring the indices are meant for: a view then yields indices into its parent ring."""
        out = self._cur if out is None else out
        _chk(out, "i32", self.B, "cur_idx out")
        _call("unreal_ring_cur_idx", self.B, self.H1, int(base_actor), ptr(self.count), ptr(out))
        return out


def ring_view(ring, b0, b1):
    """The actors [b0, b1) of a ring as a Ring of their own (no copy: frame indices are relative to the view)."""
    v = object.__new__(Ring)
    v.B, v.H, v.H1, v.objective_size = b1 - b0, ring.H, ring.H1, ring.objective_size
    H1 = ring.H1
    v.frames = ring.frames[b0 * H1 * FRAME_BYTES:b1 * H1 * FRAME_BYTES]
    for name in ("r_reward", "r_action", "r_terminal", "r_last_action", "r_last_reward"):
        setattr(v, name, getattr(ring, name)[b0 * H1:b1 * H1])
    v.r_pc = ring.r_pc[b0 * H1 * PC_CELLS:b1 * H1 * PC_CELLS]
    obj = ring.objective_size
    v.r_objective = ring.r_objective[b0 * H1 * obj:b1 * H1 * obj] if obj else None
    v.pos = ring.pos[2 * b0:2 * b1]
    for name in ("last_action", "last_reward", "count", "episode_reward", "score_out", "score_valid", "_cur"):
        setattr(v, name, getattr(ring, name)[b0:b1])
    return v


def maze_reset(ring, mask=None):
    _chk(mask, "i32", ring.B, "mask", optional=True)
    _call("unreal_maze_reset", ring.B, ring.H1, ptr(mask), ptr(ring.pos), ptr(ring.last_action),
          ptr(ring.last_reward), ptr(ring.count), ptr(ring.frames))


def maze_step(ring, actions, active=None, out_reward=None, out_terminal=None, reset_on_terminal=True,
              track_score=False):
    B = ring.B
    _chk(actions, "i32", B, "actions")
    _chk(active, "i32", B, "active", optional=True)
    _chk(out_reward, "f32", B, "out_reward", optional=True)
    _chk(out_terminal, "i32", B, "out_terminal", optional=True)
    _call("unreal_maze_step", B, ring.H1, ptr(actions), ptr(active), ptr(ring.pos), ptr(ring.last_action),
          ptr(ring.last_reward), ptr(ring.count), ptr(ring.frames), ptr(ring.r_reward), ptr(ring.r_action),
          ptr(ring.r_terminal), ptr(ring.r_last_action), ptr(ring.r_last_reward), ptr(ring.r_pc),
          ptr(out_reward), ptr(out_terminal), ptr(ring.episode_reward), ptr(ring.score_out),
          ptr(ring.score_valid), int(reset_on_terminal), int(track_score))


def maze_rollout_step(ring, actions, out_reward, out_terminal, active, active_log_t, n_steps, terminal_end,
                      next_idx=None, next_lar=None, lar_ld=0, lar_col0=0, A=0, base_actor=0):
    """maze_step + rollout_advance (+ cur_idx and lar_fill for the NEXT step's rows) in one launch.  `base_actor`: index
    of this (view's) first actor in the ring the next_idx values are meant for (see Ring.cur_idx)."""
    B = ring.B
    _chk(actions, "i32", B, "actions"); _chk(out_reward, "f32", B, "out_reward"); _chk(out_terminal, "i32", B, "out_terminal")
    for t in (active, active_log_t, n_steps, terminal_end):
        _chk(t, "i32", B)
    _chk(next_idx, "i32", B, "next_idx", optional=True)
    _chk(next_lar, "f32", (B - 1) * lar_ld + lar_col0 + A + 1 if next_lar is not None else None, "next_lar", optional=True)
    _call("unreal_maze_rollout_step", B, ring.H1, ptr(actions), ptr(ring.pos), ptr(ring.last_action), ptr(ring.last_reward),
          ptr(ring.count), ptr(ring.frames), ptr(ring.r_reward), ptr(ring.r_action), ptr(ring.r_terminal),
          ptr(ring.r_last_action), ptr(ring.r_last_reward), ptr(ring.r_pc), ptr(out_reward), ptr(out_terminal),
          ptr(ring.episode_reward), ptr(ring.score_out), ptr(ring.score_valid), ptr(active), ptr(active_log_t),
          ptr(n_steps), ptr(terminal_end), ptr(next_idx), ptr(next_lar), int(lar_ld), int(lar_col0), int(A), int(base_actor))


def maze_policy_rollout_step(ring, X, ldx, Wp, bp, Wv, bv, u, pi_out, v_out, actions, out_reward, out_terminal, active,
                             active_log_t, n_steps, terminal_end, next_idx=None, next_lar=None, lar_ld=0, lar_col0=0, A=4,
                             base_actor=0):
    """policy_step + maze_rollout_step in one launch: the workgroup that steps an actor computes its pi / V / action first
    (bit-identical to the two launches)."""
    B = ring.B
    if A != 4:
        raise ValueError("the maze has 4 actions")
    _chk(X, "f32", (B - 1) * ldx + 256, "X"); _chk(Wp, "f32", 256 * A); _chk(bp, "f32", A); _chk(Wv, "f32", 256)
    _chk(bv, "f32", 1); _chk(u, "f64", B, "u"); _chk(pi_out, "f32", B * A); _chk(v_out, "f32", B)
    _chk(actions, "i32", B, "actions"); _chk(out_reward, "f32", B, "out_reward"); _chk(out_terminal, "i32", B, "out_terminal")
    for t in (active, active_log_t, n_steps, terminal_end):
        _chk(t, "i32", B)
    _chk(next_idx, "i32", B, "next_idx", optional=True)
    _chk(next_lar, "f32", (B - 1) * lar_ld + lar_col0 + A + 1 if next_lar is not None else None, "next_lar", optional=True)
    _call("unreal_maze_policy_rollout_step", B, ring.H1, ptr(X), int(ldx), ptr(Wp), ptr(bp), ptr(Wv), ptr(bv), ptr(u),
          ptr(pi_out), ptr(v_out), ptr(actions), ptr(ring.pos), ptr(ring.last_action), ptr(ring.last_reward), ptr(ring.count),
          ptr(ring.frames), ptr(ring.r_reward), ptr(ring.r_action), ptr(ring.r_terminal), ptr(ring.r_last_action),
          ptr(ring.r_last_reward), ptr(ring.r_pc), ptr(out_reward), ptr(out_terminal), ptr(ring.episode_reward),
          ptr(ring.score_out), ptr(ring.score_valid), ptr(active), ptr(active_log_t), ptr(n_steps), ptr(terminal_end),
          ptr(next_idx), ptr(next_lar), int(lar_ld), int(lar_col0), int(A), int(base_actor))


def pixel_change_u8(frames, idx_new, idx_old, denom, out):
    N = idx_new.numel()
    _chk(frames, "u8"); _chk(idx_new, "i32", N); _chk(idx_old, "i32", N); _chk(out, "f32", N * PC_CELLS)
    _call("unreal_pixel_change_u8", N, ptr(frames), ptr(idx_new), ptr(idx_old), float(denom), ptr(out))


def _philox_shape(n, row_len, row_stride, col0):
    row_len = n if row_len is None else int(row_len)
    row_stride = row_len if row_stride is None else int(row_stride)
    if row_len <= 0 or n % row_len or col0 < 0 or col0 + row_len > row_stride:
        raise ValueError("philox: n=%d row_len=%d row_stride=%d col0=%d" % (n, row_len, row_stride, col0))
    return row_len, row_stride, int(col0)


def philox_uniform(seed, stream_id, out, row_len=None, row_stride=None, col0=0):
    """out[i] = draw (i // row_len) * row_stride + col0 + i % row_len of stream (seed, stream_id)."""
    _chk(out, "f64")
    rl, rs, c0 = _philox_shape(out.numel(), row_len, row_stride, col0)
    _call("unreal_philox_uniform", int(seed), int(stream_id), out.numel(), rl, rs, c0, ptr(out))


def philox_randint(seed, stream_id, high, out, row_len=None, row_stride=None, col0=0):
    _chk(out, "i32")
    rl, rs, c0 = _philox_shape(out.numel(), row_len, row_stride, col0)
    _call("unreal_philox_randint", int(seed), int(stream_id), out.numel(), rl, rs, c0, int(high), ptr(out))


# ---- replay --------------------------------------------------------------------------------------
def replay_sample_seq(ring, L, start_draw, seq_idx, seq_len):
    B = ring.B
    _chk(start_draw, "i32", B); _chk(seq_idx, "i32", L * B); _chk(seq_len, "i32", B)
    _call("unreal_replay_sample_seq", B, ring.H, ring.H1, L, ptr(start_draw), ptr(ring.count),
          ptr(ring.r_terminal), ptr(seq_idx), ptr(seq_len))


def replay_sample_rp(ring, coin, u, rp_idx, rp_class, mode=0):
    B = ring.B
    _chk(coin, "i32", B); _chk(u, "f64", B); _chk(rp_idx, "i32", 3 * B); _chk(rp_class, "i32", B)
    _call("unreal_replay_sample_rp", B, ring.H, ring.H1, ptr(coin), ptr(u), ptr(ring.count),
          ptr(ring.r_reward), ptr(rp_idx), ptr(rp_class), int(mode))


def hostfed_step(ring, staged, actions, rewards, terminals, active=None, out_reward=None, out_terminal=None,
                 reset_on_terminal=True, track_score=False, clip_reward=True, pc_denom=48.0 * 255.0):
    B = ring.B
    _chk(staged, "u8", B * FRAME_BYTES, "staged"); _chk(actions, "i32", B); _chk(rewards, "f32", B)
    _chk(terminals, "i32", B); _chk(active, "i32", B, optional=True)
    _chk(out_reward, "f32", B, optional=True); _chk(out_terminal, "i32", B, optional=True)
    _call("unreal_hostfed_step", B, ring.H1, ptr(staged), ptr(actions), ptr(rewards), ptr(terminals), ptr(active),
          ptr(ring.last_action), ptr(ring.last_reward), ptr(ring.count), ptr(ring.frames), ptr(ring.r_reward),
          ptr(ring.r_action), ptr(ring.r_terminal), ptr(ring.r_last_action), ptr(ring.r_last_reward), ptr(ring.r_pc),
          ptr(out_reward), ptr(out_terminal), ptr(ring.episode_reward), ptr(ring.score_out), ptr(ring.score_valid),
          int(reset_on_terminal), int(track_score), int(clip_reward), float(pc_denom))


def hostfed_reset(ring, staged, mask=None):
    _chk(staged, "u8", ring.B * FRAME_BYTES, "staged"); _chk(mask, "i32", ring.B, optional=True)
    _call("unreal_hostfed_reset", ring.B, ring.H1, ptr(mask), ptr(staged), ptr(ring.last_action),
          ptr(ring.last_reward), ptr(ring.count), ptr(ring.frames))


def base_returns(B, T, rewards, values, n_steps, boot_v, terminal_end, gamma, R_out, adv_out):
    for t in (rewards, values, R_out, adv_out):
        _chk(t, "f32", B * T)
    _chk(n_steps, "i32", B); _chk(boot_v, "f32", B); _chk(terminal_end, "i32", B)
    _call("unreal_base_returns", B, T, ptr(rewards), ptr(values), ptr(n_steps), ptr(boot_v),
          ptr(terminal_end), float(gamma), ptr(R_out), ptr(adv_out))


def vr_returns(ring, L, seq_idx, seq_len, boot_v, gamma, R_out):
    B = ring.B
    _chk(seq_idx, "i32", L * B); _chk(seq_len, "i32", B); _chk(boot_v, "f32", B); _chk(R_out, "f32", (L - 1) * B)
    _call("unreal_vr_returns", B, L, ptr(seq_idx), ptr(seq_len), ptr(ring.r_reward), ptr(ring.r_terminal),
          ptr(boot_v), float(gamma), ptr(R_out))


def pc_returns(ring, L, seq_idx, seq_len, boot_qmax, gamma_pc, R_out):
    B = ring.B
    _chk(seq_idx, "i32", L * B); _chk(seq_len, "i32", B); _chk(boot_qmax, "f32", B * PC_CELLS)
    _chk(R_out, "f32", (L - 1) * B * PC_CELLS)
    _call("unreal_pc_returns", B, L, ptr(seq_idx), ptr(seq_len), ptr(ring.r_pc), ptr(ring.r_terminal),
          ptr(boot_qmax), float(gamma_pc), ptr(R_out))


def lar_fill(rows, A, last_action, last_reward, idx, xcat, ld, col0=256, clip=False):
    _chk(last_action, "i32"); _chk(last_reward, "f32"); _chk(idx, "i32", rows, optional=True)
    _chk(xcat, "f32", rows * ld)
    if idx is None and (last_action.numel() < rows or last_reward.numel() < rows):
        raise ValueError("lar_fill: per-row sources too short")
    _call("unreal_lar_fill", rows, A, ptr(last_action), ptr(last_reward), ptr(idx), ptr(xcat), ld, col0, int(clip))


def objective_put(ring, staged, active=None):
    """staged [B][obj] -> the current slot of every (active) actor."""
    obj = ring.objective_size
    if not obj:
        raise ValueError("this ring stores no objective vectors")
    _chk(staged, "f32", ring.B * obj, "staged objective"); _chk(active, "i32", ring.B, "active", optional=True)
    _call("unreal_objective_put", ring.B, ring.H1, obj, ptr(ring.count), ptr(active), ptr(staged), ptr(ring.r_objective))


def objective_fill(ring, rows, idx, xcat, ld, col0, slot_offset=0):
    obj = ring.objective_size
    if not obj:
        raise ValueError("this ring stores no objective vectors")
    _chk(idx, "i32", rows, "idx"); _chk(xcat, "f32", (rows - 1) * ld + col0 + obj, "xcat")
    _call("unreal_objective_fill", rows, obj, ring.H1, ptr(ring.r_objective), ptr(idx), slot_offset, ptr(xcat), ld, col0)


def gather_i32(src, idx, out):
    _chk(src, "i32"); _chk(idx, "i32"); _chk(out, "i32", idx.numel())
    _call("unreal_gather_i32", idx.numel(), ptr(src), ptr(idx), ptr(out))


def rollout_advance(B, terminal_t, active, active_log_t, n_steps, terminal_end):
    for t in (terminal_t, active, active_log_t, n_steps, terminal_end):
        _chk(t, "i32", B)
    _call("unreal_rollout_advance", B, ptr(terminal_t), ptr(active), ptr(active_log_t), ptr(n_steps),
          ptr(terminal_end))


def seq_mask(B, T, seq_len, mask):
    _chk(seq_len, "i32", B); _chk(mask, "i32", B * T)
    _call("unreal_seq_mask", B, T, ptr(seq_len), ptr(mask))


def seq_last_idx(B, seq_idx, seq_len, out):
    _chk(seq_idx, "i32", 2 * B); _chk(seq_len, "i32", B); _chk(out, "i32", B)
    _call("unreal_seq_last_idx", B, ptr(seq_idx), ptr(seq_len), ptr(out))


def rollout_stats(B, n_steps, score_valid, score_out, stats):
    _chk(n_steps, "i32", B); _chk(score_valid, "i32", B); _chk(score_out, "f32", B); _chk(stats, "f64", 3)
    _call("unreal_rollout_stats", B, ptr(n_steps), ptr(score_valid), ptr(score_out), ptr(stats))


def reset_state(B, terminal_end, c, h):
    _chk(terminal_end, "i32", B); _chk(c, "f32", B * 256); _chk(h, "f32", B * 256)
    _call("unreal_reset_state", B, ptr(terminal_end), ptr(c), ptr(h))


# ---- network -------------------------------------------------------------------------------------
ENC_PREPARED_BYTES = 45072          # include/unreal_hip.h: UNREAL_ENCODER_PREPARED_BYTES


def encoder_prepare(W1, b1, W2, scale, prepared=None):
    """The weights' share of encoder_fwd's prologue (scales + MFMA operand fragments) -> `prepared` (uint8
    [ENC_PREPARED_BYTES]); valid for these W1 / b1 / W2 values and this frame scale.  Returns the block."""
    _chk(W1, "f32", 3072); _chk(b1, "f32", 16); _chk(W2, "f32", 8192)
    if prepared is None:
        prepared = torch.empty(ENC_PREPARED_BYTES, dtype=torch.uint8, device=W1.device)
    _chk(prepared, "u8", ENC_PREPARED_BYTES, "prepared")
    _call("unreal_encoder_prepare", ptr(W1), ptr(b1), ptr(W2), float(scale), ptr(prepared), prepared.numel())
    return prepared


def encoder_fwd(frames, frame_idx, scale, W1, b1, W2, b2, f2_out, c1_out=None, n_frames_pool=None, relu_bits=None,
                f2_max=None, c1_max=None, prepared=None):
    """relu_bits: optional int16 [N * RELU_WORDS], bit (j % 16) of word j / 16 of a frame = f2[j] > 0.
    f2_max: optional absmax slot that receives max f2 (the A scale of the fc GEMM that follows).
    prepared: optional block from encoder_prepare(W1, b1, W2, scale) -- same results, shorter prologue per launch."""
    N = frame_idx.numel()
    _chk(frames, "u8"); _chk(frame_idx, "i32", N)
    _chk(W1, "f32", 3072); _chk(b1, "f32", 16); _chk(W2, "f32", 8192); _chk(b2, "f32", 32)
    _chk(f2_out, "f32", N * F2_DIM); _chk(c1_out, "f32", N * C1_DIM, optional=True)
    _chk(relu_bits, "i16", N * RELU_WORDS, "relu_bits", optional=True)
    _chk(f2_max, "f32", 1, "f2_max", optional=True); _chk(c1_max, "f32", 1, "c1_max", optional=True)
    _chk(prepared, "u8", ENC_PREPARED_BYTES, "prepared", optional=True)
    _call("unreal_encoder_fwd", N, ptr(frames), ptr(frame_idx), float(scale), ptr(W1), ptr(b1), ptr(W2), ptr(b2),
          ptr(c1_out), ptr(f2_out), ptr(relu_bits), ptr(f2_max), ptr(c1_max), ptr(prepared))


def encoder_bwd(frames, frame_idx, scale, W2, c1_saved, d2, dW1, db1, dW2, db2, c1_max=None, d2_max=None):
    """c1_max / d2_max: absmax slots covering c1_saved / d2 (None: reduced here with one extra launch each)."""
    N = frame_idx.numel()
    _chk(c1_max, "f32", 1, "c1_max", optional=True); _chk(d2_max, "f32", 1, "d2_max", optional=True)
    _chk(frames, "u8"); _chk(frame_idx, "i32", N); _chk(W2, "f32", 8192)
    _chk(c1_saved, "f32", N * C1_DIM); _chk(d2, "f32", N * F2_DIM)
    _chk(dW1, "f32", 3072); _chk(db1, "f32", 16); _chk(dW2, "f32", 8192); _chk(db2, "f32", 32)
    c1_max = _absmax_of(c1_saved, 1, N * C1_DIM, N * C1_DIM, c1_max)
    d2_max = _absmax_of(d2, 1, N * F2_DIM, N * F2_DIM, d2_max)
    _call("unreal_encoder_bwd", N, ptr(frames), ptr(frame_idx), float(scale), ptr(W2), ptr(c1_saved), ptr(c1_max), ptr(d2),
          ptr(d2_max), ptr(dW1), ptr(db1), ptr(dW2), ptr(db2))


def gemm(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias=None, mask=None, ldm=0, flags=0, splitk=1):
    _chk(A, "f32", ((K - 1) * lda + M) if transA else ((M - 1) * lda + K), "A")
    _chk(B, "f32", ((N - 1) * ldb + K) if transB else ((K - 1) * ldb + N), "B")
    _chk(C, "f32", (M - 1) * ldc + N, "C")
    _chk(bias, "f32", N, "bias", optional=True)
    _chk(mask, "f32", (M - 1) * ldm + N if ldm else None, "mask", optional=True)
    _call("unreal_gemm_f32", int(transA), int(transB), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(C), ldc, ptr(bias),
          ptr(mask), ldm, flags, splitk)


class AbsmaxPool(object):
    """Absmax slots (include/unreal_hip.h: one float on the device holding max |x| of a tensor) for one pass of the
    trainer: `reset()` zeroes them all with one fill, `new()` hands out the next one.  A producer kernel maxes its
    outputs into a slot, the fp16x2 GEMM that consumes the tensor derives the tensor's power-of-two scale from it."""

    def __init__(self, device, n=1024):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.n, self.next = n, 0

    def reset(self):
        self.buf.zero_()
        self.next = 0

    def new(self):
        if self.next >= self.n:
            raise RuntimeError("AbsmaxPool exhausted (%d slots): reset() it once per pass" % self.n)
        self.next += 1
        return self.buf[self.next - 1:self.next]


_SCRATCH = {}


def _scratch_slot(like):
    """A zeroed slot for callers that pass no absmax of their own (tests, one-off launches): ring of 64, stream-ordered."""
    dev = like.device
    st = _SCRATCH.get(dev)
    if st is None:
        st = _SCRATCH[dev] = [torch.zeros(64, dtype=torch.float32, device=dev), 0]
    st[1] = (st[1] + 1) % 64
    slot = st[0][st[1]:st[1] + 1]
    slot.zero_()
    return slot


def absmax(rows, cols, x, ld, slot):
    """slot = max(slot, max |x[r][c]|) over the rows x cols view of x (row stride ld)."""
    _chk(x, "f32", (rows - 1) * ld + cols, "x"); _chk(slot, "f32", 1, "slot")
    _call("unreal_absmax_f32", rows, cols, ptr(x), ld, ptr(slot))
    return slot


def _absmax_of(x, rows, cols, ld, slot):
    return slot if slot is not None else absmax(rows, cols, x, ld, _scratch_slot(x))


class SplitWeights:
    """fp16x2 shadow of one weight matrix as unreal_gemm_f32_split_nt wants its W operand: planes[t][n][k] (t = hi, lo of
    w * 2^k, k from the matrix's absmax slot `wmax`), rows padded with zeros to a multiple of 32 k.  `refresh()` re-reduces
    the maximum and re-splits from the live fp32 weights (after an optimiser step)."""

    def __init__(self, src, rows, cols, ld_src, transpose, offset=0, row_perm=0, wmax=None, defer=False):
        """wmax: a one-float view the caller owns (ShadowSet keeps the slots of all its shadows in one tensor, zeroed with one
        fill); defer: do not split now (a ShadowSet refreshes all its members together)."""
        self.src, self.rows, self.cols, self.ld_src, self.transpose, self.offset = src, rows, cols, ld_src, transpose, offset
        self.row_perm = row_perm
        self.N, self.K = (cols, rows) if transpose else (rows, cols)
        self.ldw = (self.K + 31) // 32 * 32
        self.plane = self.N * self.ldw
        self.planes = torch.zeros(2 * self.plane, dtype=torch.int16, device=src.device)
        self.wmax = torch.zeros(1, dtype=torch.float32, device=src.device) if wmax is None else wmax
        if not defer:
            self.refresh()

    def multi_descs(self):
        """-> (abs record or None, [split records]) for ShadowSet: pointers and sizes as Python ints."""
        if self.ld_src != self.cols:
            return None, None                         # a column window: not contiguous, refreshed by itself
        src = self.src.data_ptr() + 4 * self.offset
        return ((src, self.wmax.data_ptr(), self.rows * self.cols),
                [(src, self.planes.data_ptr(), self.wmax.data_ptr(), self.rows, self.cols, self.ld_src, int(bool(self.transpose)),
                  int(self.row_perm), self.ldw, self.plane)])

    def refresh(self):
        self.wmax.zero_()
        absmax(self.rows, self.cols, self.src[self.offset:], self.ld_src, self.wmax)
        split_f16x2(self.rows, self.cols, self.src[self.offset:], self.ld_src, self.transpose, self.planes, self.ldw,
                    self.plane, self.wmax, self.row_perm)


def split_f16x2(rows, cols, src, ld_src, transpose, dst, ld_dst, plane, wmax, row_perm=0):
    _chk(src, "f32", (rows - 1) * ld_src + cols, "src"); _chk(wmax, "f32", 1, "wmax")
    orows, ocols = (cols, rows) if transpose else (rows, cols)
    _chk(dst, "i16", plane + (orows - 1) * ld_dst + ocols, "dst")
    _call("unreal_split_f16x2", rows, cols, ptr(src), ld_src, int(bool(transpose)), int(row_perm), ptr(dst), ld_dst,
          plane, ptr(wmax))


def split_bf16x3(rows, cols, src, ld_src, transpose, dst, ld_dst, plane, row_perm=0):
    _chk(src, "f32", (rows - 1) * ld_src + cols, "src")
    orows, ocols = (cols, rows) if transpose else (rows, cols)
    _chk(dst, "i16", 2 * plane + (orows - 1) * ld_dst + ocols, "dst")
    _call("unreal_split_bf16x3", rows, cols, ptr(src), ld_src, int(bool(transpose)), int(row_perm), ptr(dst), ld_dst,
          plane)


class LstmKernelShadow:
    """Gate-interleaved fp16x2 shadow of the WHOLE BasicLSTMCell kernel [K_x + 256, 1024] as unreal_lstm_step_fwd(x=...)
    multiplies it: planes[t][1024][pad32(K_x) + 256] -- input rows, zero padding to a K tile, recurrent rows; one scale
    (absmax slot) for the whole kernel."""

    def __init__(self, kernel, K_x, wmax=None, defer=False):
        self.src, self.K_x, self.row_perm = kernel, K_x, 1
        self.kxpad = (K_x + 31) // 32 * 32
        self.N, self.K = 1024, self.kxpad + 256
        self.ldw = self.K
        self.plane = self.N * self.ldw
        self.planes = torch.zeros(2 * self.plane, dtype=torch.int16, device=kernel.device)
        self.wmax = torch.zeros(1, dtype=torch.float32, device=kernel.device) if wmax is None else wmax
        if not defer:
            self.refresh()

    def multi_descs(self):
        src, dst, wm = self.src.data_ptr(), self.planes.data_ptr(), self.wmax.data_ptr()
        return ((src, wm, (self.K_x + 256) * 1024),
                [(src, dst, wm, self.K_x, 1024, 1024, 1, 1, self.ldw, self.plane),
                 (src + 4 * self.K_x * 1024, dst + 2 * self.kxpad, wm, 256, 1024, 1024, 1, 1, self.ldw, self.plane)])

    def refresh(self):
        self.wmax.zero_()
        absmax(self.K_x + 256, 1024, self.src, 1024, self.wmax)
        split_f16x2(self.K_x, 1024, self.src, 1024, True, self.planes, self.ldw, self.plane, self.wmax, 1)
        split_f16x2(256, 1024, self.src[self.K_x * 1024:], 1024, True, self.planes[self.kxpad:], self.ldw, self.plane,
                    self.wmax, 1)


class ShadowSet(object):
    """All weight shadows of a network refreshed together: one fill of their absmax slots, one launch that reduces every
    matrix's maximum, one that writes every matrix's planes (unreal_shadow_refresh_multi) -- instead of fill + maximum +
    split per matrix.  `make` is called with a factory: wmax_view = slots[i:i + 1] for shadow i."""

    def __init__(self, device, n_shadows):
        self.device = device
        self.slots = torch.zeros(n_shadows, dtype=torch.float32, device=device)
        self.members, self._descs = [], None

    def slot(self):
        i = len(self.members)
        if i >= self.slots.numel():
            raise RuntimeError("ShadowSet: more shadows than slots")
        return self.slots[i:i + 1]

    def add(self, shadow):
        self.members.append(shadow)
        self._descs = None
        return shadow

    def _build(self):
        abs_rows, split_rows, singles = [], [], []
        ab, sb = 0, 0
        for m in self.members:
            a, sp = m.multi_descs()
            if a is None:
                singles.append(m)
                continue
            abs_rows.append([a[0], a[1], a[2], ab])
            ab += (a[2] + 8191) // 8192
            for r in sp:
                tiles_x = (r[4] + 31) // 32
                split_rows.append(list(r) + [tiles_x, sb])
                sb += tiles_x * ((r[3] + 31) // 32)
        mk = lambda rows: torch.tensor(rows, dtype=torch.int64).to(self.device).contiguous()
        self._descs = (mk(abs_rows), len(abs_rows), ab, mk(split_rows), len(split_rows), sb, singles)

    def refresh(self):
        if self._descs is None:
            self._build()
        a, na, ab, sp, ns, sb, singles = self._descs
        self.slots.zero_()
        if na:
            _call("unreal_shadow_refresh_multi", ptr(a), na, ab, ptr(sp), ns, sb)
        for m in singles:                    # (zeroed above; their own refresh would zero the slot again: harmless)
            m.refresh()


def gemm_split_nt(M, N, K, A, lda, W, C, ldc, bias=None, mask=None, ldm=0, flags=0, splitk=1, a_max=None, c_max=None):
    """C = A[M,K] @ W[N,K]^T (fp32-grade: fp16 hi + lo operands with per-tensor scales, 3 term pairs on the fp16 matrix
    cores); W is a SplitWeights.  a_max: absmax slot covering A (None: reduced here with one extra launch); c_max: slot
    that receives max |C|.
    mask: fp32 [M, ldm] with GEMM_RELU_MASK, or int16 bit words [M, ldm] with GEMM_RELU_BITS (encoder_fwd relu_bits)."""
    if W.N != N or W.K != K:
        raise ValueError("split weights are [%d,%d], GEMM wants [%d,%d]" % (W.N, W.K, N, K))
    _chk(A, "f32", (M - 1) * lda + K, "A"); _chk(C, "f32", (M - 1) * ldc + N, "C")
    _chk(bias, "f32", N, "bias", optional=True)
    if flags & GEMM_RELU_BITS:
        _chk(mask, "i16", (M - 1) * ldm + (N + 15) // 16, "mask bits")
    else:
        _chk(mask, "f32", (M - 1) * ldm + N if ldm else None, "mask", optional=True)
    _chk(a_max, "f32", 1, "a_max", optional=True); _chk(c_max, "f32", 1, "c_max", optional=True)
    if c_max is not None and (flags & GEMM_ATOMIC):
        raise ValueError("gemm_split_nt: the split-K / atomic epilogue cannot commit max |C| (c_max); reduce it with "
                         "ops.absmax after the launch")
    if K < 64 and not (flags & GEMM_RELU_BITS):
        # one or two K tiles: nothing to split for -- hi + lo carries 22 bits per element, which only pays off against the
        # rounding of a long accumulation (at K = 3 the split result is ~2.5x the fp32 chain's error).  Such products (none
        # on the trainer's path) go to the exact fp32-MFMA kernel with the weights the shadow was made from.
        if W.transpose:          # src is [K][N]
            gemm(False, False, M, N, K, A, lda, W.src[W.offset:], W.ld_src, C, ldc, bias, mask, ldm, flags, splitk)
        else:                    # src is [N][K]
            gemm(False, True, M, N, K, A, lda, W.src[W.offset:], W.ld_src, C, ldc, bias, mask, ldm, flags, splitk)
        if c_max is not None:
            absmax(M, N, C, ldc, c_max)
        return
    a_max = _absmax_of(A, M, K, lda, a_max)
    _call("unreal_gemm_f32_split_nt", M, N, K, ptr(A), lda, ptr(a_max), ptr(W.planes), W.ldw, W.plane, ptr(W.wmax), ptr(C), ldc,
          ptr(c_max), ptr(bias), ptr(mask), ldm, flags, splitk)


FEW_ROWS = 2048        # at most this many rows: a long-K product has too few tiles for the chip (gemm_split_nt_slabs)


def slab_count(M, N, K):
    """K slabs for gemm_split_nt_slabs, by the same-process A/B at N = 256, K = 2592 (tools/exp/fc_slabs_ab.py,
    profiles/r04_ab_summary.md 3e): 8 up to 32 tiles of 64 x 64 (<= 512 rows), 4 up to 64, 2 up to 128 (2048 rows); from 4096
    rows on the one-launch kernel wins.  0 = use the one-launch kernel."""
    if M > FEW_ROWS or K < 1024:
        return 0
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    return 8 if tiles <= 32 else 4 if tiles <= 64 else 2 if tiles <= 128 else 0


def gemm_split_nt_slabs(M, N, K, A, lda, W, C, ldc, partials, splitk, bias=None, flags=0, a_max=None, c_max=None):
    """gemm_split_nt for few rows and a long K: `splitk` K slabs in separate workgroups -> partials (>= splitk * M * pad4(N)
    floats), added in slab order by a second launch that applies bias / ReLU and commits max |C| (deterministic)."""
    if W.N != N or W.K != K:
        raise ValueError("split weights are [%d,%d], GEMM wants [%d,%d]" % (W.N, W.K, N, K))
    _chk(A, "f32", (M - 1) * lda + K, "A"); _chk(C, "f32", (M - 1) * ldc + N, "C")
    _chk(bias, "f32", N, "bias", optional=True); _chk(partials, "f32", splitk * M * ((N + 3) // 4 * 4), "partials")
    _chk(a_max, "f32", 1, "a_max", optional=True); _chk(c_max, "f32", 1, "c_max", optional=True)
    a_max = _absmax_of(A, M, K, lda, a_max)
    _call("unreal_gemm_f32_split_nt_slabs", M, N, K, ptr(A), lda, ptr(a_max), ptr(W.planes), W.ldw, W.plane, ptr(W.wmax), ptr(C),
          ldc, ptr(c_max), ptr(bias), flags, splitk, ptr(partials), partials.numel())


def gemm_split_tn(M, N, K, A, lda, B, ldb, C, ldc, splitk=1, colsum=None, a_max=None, b_max=None):
    """C[M,N] += A[K,M]^T @ B[K,N] (wgrad; fp32-grade on the fp16 matrix cores, split-K atomics into C);
    colsum[N] += column sums of B (the bias gradient of the same layer) when given.
    a_max / b_max: absmax slots covering A / B (None: reduced here with one extra launch each)."""
    _chk(A, "f32", (K - 1) * lda + M, "A"); _chk(B, "f32", (K - 1) * ldb + N, "B"); _chk(C, "f32", (M - 1) * ldc + N, "C")
    _chk(colsum, "f32", N, "colsum", optional=True)
    _chk(a_max, "f32", 1, "a_max", optional=True); _chk(b_max, "f32", 1, "b_max", optional=True)
    a_max = _absmax_of(A, K, M, lda, a_max)
    b_max = _absmax_of(B, K, N, ldb, b_max)
    _call("unreal_gemm_f32_split_tn", M, N, K, ptr(A), lda, ptr(a_max), ptr(B), ldb, ptr(b_max), ptr(C), ldc, ptr(colsum),
          splitk)


def lstm_step_fwd(rows, h_prev, Wh, gates, bias, c_prev, c_out, h_out, ld_hprev=256, ld_h=256, x=None, ldx=0, Kx=0,
                  x_max=None):
    """One BasicLSTMCell step with the gate math fused into the product.
    x None : gates (in: x-half pre-activations, out: activated gates) += h_prev @ Wh;
             Wh = SplitWeights(kernel recurrent rows, transpose=True, row_perm=1).
    x given: gates (out) = act([x | h_prev] @ kernel); Wh = LstmKernelShadow (gate-interleaved [1024, pad32(Kx) + 256])."""
    kxpad = (Kx + 31) // 32 * 32 if x is not None else 0
    if Wh.N != 1024 or Wh.K != kxpad + 256 or Wh.row_perm != 1:
        raise ValueError("lstm_step_fwd needs the gate-interleaved [1024,%d] shadow of the LSTM kernel" % (kxpad + 256))
    _chk(h_prev, "f32", (rows - 1) * ld_hprev + 256, "h_prev"); _chk(gates, "f32", rows * 1024, "gates")
    _chk(bias, "f32", 1024, "bias"); _chk(c_prev, "f32", rows * 256, "c_prev"); _chk(c_out, "f32", rows * 256, "c_out")
    _chk(h_out, "f32", (rows - 1) * ld_h + 256, "h_out")
    _chk(x, "f32", (rows - 1) * ldx + Kx if x is not None else None, "x", optional=True)
    _chk(x_max, "f32", 1, "x_max", optional=True)
    if x is not None:        # |h_prev| < 1 is covered by the kernel; x needs its slot
        x_max = _absmax_of(x, rows, Kx, ldx, x_max)
    _call("unreal_lstm_step_fwd", rows, ptr(x), ldx, Kx, ptr(x_max), ptr(h_prev), ld_hprev, ptr(Wh.planes), Wh.ldw, Wh.plane,
          ptr(Wh.wmax), ptr(gates), ptr(bias), ptr(c_prev), ptr(c_out), ptr(h_out), ld_h)


def lstm_gates_fwd(rows, pre, bias, c_prev, gates_act, c_out, h_out, ld_h=256):
    _chk(pre, "f32", rows * 1024); _chk(bias, "f32", 1024); _chk(c_prev, "f32", rows * 256)
    _chk(gates_act, "f32", rows * 1024, optional=True); _chk(c_out, "f32", rows * 256)
    _chk(h_out, "f32", (rows - 1) * ld_h + 256)
    _call("unreal_lstm_gates_fwd", rows, ptr(pre), ptr(bias), ptr(c_prev), ptr(gates_act), ptr(c_out), ptr(h_out),
          ld_h)


def lstm_gates_bwd(rows, dh_above, dh_rec, dc_io, gates_act, c_prev, c_new, dpre, c_max0=None, c_max1=None):
    _chk(dh_above, "f32", rows * 256); _chk(dh_rec, "f32", rows * 256, optional=True)
    _chk(dc_io, "f32", rows * 256); _chk(gates_act, "f32", rows * 1024); _chk(c_prev, "f32", rows * 256)
    _chk(c_new, "f32", rows * 256); _chk(dpre, "f32", rows * 1024)
    _chk(c_max0, "f32", 1, "c_max0", optional=True); _chk(c_max1, "f32", 1, "c_max1", optional=True)
    _call("unreal_lstm_gates_bwd", rows, ptr(dh_above), ptr(dh_rec), ptr(dc_io), ptr(gates_act), ptr(c_prev),
          ptr(c_new), ptr(dpre), ptr(c_max0), ptr(c_max1))


def lstm_bptt_step(rows, d_gates, Wh, dh_above, dc_io, gates_act, c_prev, c_new, dpre, a_max=None, c_max0=None,
                   c_max1=None):
    """dh_rec = d_gates @ Wh^T and, in the same launch, the gate backward of the earlier step (dh = dh_above + dh_rec).
    Wh: SplitWeights(kernel recurrent rows, transpose=False) -- [256, 1024]."""
    if Wh.N != 256 or Wh.K != 1024 or getattr(Wh, "row_perm", 0):
        raise ValueError("lstm_bptt_step needs the natural [256,1024] shadow of the recurrent kernel rows")
    _chk(d_gates, "f32", rows * 1024, "d_gates"); _chk(dh_above, "f32", rows * 256); _chk(dc_io, "f32", rows * 256)
    _chk(gates_act, "f32", rows * 1024); _chk(c_prev, "f32", rows * 256); _chk(c_new, "f32", rows * 256)
    _chk(dpre, "f32", rows * 1024)
    _chk(a_max, "f32", 1, "a_max", optional=True); _chk(c_max0, "f32", 1, "c_max0", optional=True)
    _chk(c_max1, "f32", 1, "c_max1", optional=True)
    a_max = _absmax_of(d_gates, rows, 1024, 1024, a_max)
    _call("unreal_lstm_bptt_step", rows, ptr(d_gates), ptr(a_max), ptr(Wh.planes), Wh.ldw, Wh.plane, ptr(Wh.wmax),
          ptr(dh_above), ptr(dc_io), ptr(gates_act), ptr(c_prev), ptr(c_new), ptr(dpre), ptr(c_max0), ptr(c_max1))


def linear_small_fwd(rows, K, NOUT, X, ldx, W, b, out, ldo):
    _chk(X, "f32", (rows - 1) * ldx + K); _chk(W, "f32", K * NOUT); _chk(b, "f32", NOUT)
    _chk(out, "f32", (rows - 1) * ldo + NOUT)
    _call("unreal_linear_small_fwd", rows, K, NOUT, ptr(X), ldx, ptr(W), ptr(b), ptr(out), ldo)


def linear_small_bwd(rows, K, NOUT, X, ldx, dO, ldo, W, dX, lddx, accumulate_dx, dW, db, dw_stride_k=0, dw_stride_n=0):
    _chk(X, "f32", (rows - 1) * ldx + K); _chk(dO, "f32", (rows - 1) * ldo + NOUT)
    _chk(W, "f32", K * NOUT, optional=dX is None)
    _chk(dX, "f32", (rows - 1) * lddx + K, optional=True)
    sk, sn = (dw_stride_k, dw_stride_n) if (dw_stride_k or dw_stride_n) else (NOUT, 1)
    _chk(dW, "f32", (K - 1) * sk + (NOUT - 1) * sn + 1)
    _chk(db, "f32", NOUT, optional=True)
    _call("unreal_linear_small_bwd", rows, K, NOUT, ptr(X), ldx, ptr(dO), ldo, ptr(W), ptr(dX), lddx,
          int(accumulate_dx), ptr(dW), int(dw_stride_k), int(dw_stride_n), ptr(db))


def policy_step(rows, A, X, ldx, Wp, bp, Wv, bv, u, pi_out, v_out, action):
    """pi, v and the sampled (u) or greedy (u None) action of `rows` feature rows in one launch."""
    _chk(X, "f32", (rows - 1) * ldx + 256, "X"); _chk(Wp, "f32", 256 * A); _chk(bp, "f32", A); _chk(Wv, "f32", 256)
    _chk(bv, "f32", 1); _chk(u, "f64", rows, "u", optional=True); _chk(pi_out, "f32", rows * A); _chk(v_out, "f32", rows)
    _chk(action, "i32", rows)
    _call("unreal_policy_step", rows, A, ptr(X), ldx, ptr(Wp), ptr(bp), ptr(Wv), ptr(bv), ptr(u), ptr(pi_out), ptr(v_out),
          ptr(action))


def softmax_sample(rows, A, logits_pi, ld, u=None, action=None):
    _chk(logits_pi, "f32", (rows - 1) * ld + A); _chk(u, "f64", rows, optional=True)
    _chk(action, "i32", rows, optional=True)
    _call("unreal_softmax_sample", rows, A, ptr(logits_pi), ld, ptr(u), ptr(action))


def base_loss_grad(rows, A, pi, ld_pi, v, action, adv, R, active, beta, grad_scale, dlogits, dv, losses):
    _chk(pi, "f32", (rows - 1) * ld_pi + A); _chk(v, "f32", rows); _chk(action, "i32", rows)
    _chk(adv, "f32", rows); _chk(R, "f32", rows); _chk(active, "i32", rows)
    _chk(dlogits, "f32", rows * A); _chk(dv, "f32", rows); _chk(losses, "f32", 3)
    _call("unreal_base_loss_grad", rows, A, ptr(pi), ld_pi, ptr(v), ptr(action), ptr(adv), ptr(R), ptr(active),
          float(beta), float(grad_scale), ptr(dlogits), ptr(dv), ptr(losses))


def vr_loss_grad(rows, v, R, mask, grad_scale, dv, loss):
    _chk(v, "f32", rows); _chk(R, "f32", rows); _chk(mask, "i32", rows); _chk(dv, "f32", rows); _chk(loss, "f32", 1)
    _call("unreal_vr_loss_grad", rows, ptr(v), ptr(R), ptr(mask), float(grad_scale), ptr(dv), ptr(loss))


def rp_loss_grad(rows, logits, cls, grad_scale, prob, dlogits, loss):
    _chk(logits, "f32", rows * 3); _chk(cls, "i32", rows); _chk(prob, "f32", rows * 3, optional=True)
    _chk(dlogits, "f32", rows * 3); _chk(loss, "f32", 1)
    _call("unreal_rp_loss_grad", rows, ptr(logits), ptr(cls), float(grad_scale), ptr(prob), ptr(dlogits), ptr(loss))


def colsum(rows, cols, X, ld, out):
    _chk(X, "f32", (rows - 1) * ld + cols); _chk(out, "f32", cols)
    _call("unreal_colsum", rows, cols, ptr(X), ld, ptr(out))


def relu_mask(rows, cols, d, ldd, src, lds):
    _chk(d, "f32", (rows - 1) * ldd + cols); _chk(src, "f32", (rows - 1) * lds + cols)
    _call("unreal_relu_mask", rows, cols, ptr(d), ldd, ptr(src), lds)


def pc_deconv_fwd(N, A, hp, Wv, bv, Wa, ba, qmax=None, action=None, target=None, mask=None, lam=0.0,
                  grad_scale=1.0, d_dec=None, loss=None, hp_max=None, ddec_max=None):
    """hp_max: absmax slot covering hp (None: reduced here with one extra launch); ddec_max: slot that receives an upper
    bound of max |d_dec| (training mode; what pc_deconv_bwd scales its d_dec planes by)."""
    _chk(hp, "f32", N * F2_DIM); _chk(Wv, "f32", 512); _chk(bv, "f32", 1); _chk(Wa, "f32", 512 * A); _chk(ba, "f32", A)
    _chk(qmax, "f32", N * PC_CELLS, optional=True); _chk(action, "i32", N, optional=True)
    _chk(target, "f32", N * PC_CELLS, optional=True); _chk(mask, "i32", N, optional=True)
    _chk(d_dec, "f32", N * PC_CELLS * (1 + A), optional=True); _chk(loss, "f32", 1, optional=True)
    _chk(hp_max, "f32", 1, "hp_max", optional=True); _chk(ddec_max, "f32", 1, "ddec_max", optional=True)
    hp_max = _absmax_of(hp, N, F2_DIM, F2_DIM, hp_max)
    _call("unreal_pc_deconv_fwd", N, A, ptr(hp), ptr(hp_max), ptr(Wv), ptr(bv), ptr(Wa), ptr(ba), ptr(qmax), ptr(action),
          ptr(target), ptr(mask), float(lam), float(grad_scale), ptr(d_dec), ptr(ddec_max), ptr(loss))


def pc_deconv_bwd(N, A, hp, d_dec, Wv, Wa, d_hp, dWv, dbv, dWa, dba, dhp_max=None, hp_max=None, ddec_max=None):
    """hp_max / ddec_max: absmax slots covering hp / d_dec (None: reduced here with one extra launch each)."""
    _chk(hp, "f32", N * F2_DIM); _chk(d_dec, "f32", N * PC_CELLS * (1 + A)); _chk(Wv, "f32", 512)
    _chk(Wa, "f32", 512 * A); _chk(d_hp, "f32", N * F2_DIM); _chk(dWv, "f32", 512); _chk(dbv, "f32", 1)
    _chk(dWa, "f32", 512 * A); _chk(dba, "f32", A)
    _chk(dhp_max, "f32", 1, "dhp_max", optional=True)
    _chk(hp_max, "f32", 1, "hp_max", optional=True); _chk(ddec_max, "f32", 1, "ddec_max", optional=True)
    hp_max = _absmax_of(hp, N, F2_DIM, F2_DIM, hp_max)
    ddec_max = _absmax_of(d_dec, N, PC_CELLS * (1 + A), PC_CELLS * (1 + A), ddec_max)
    _call("unreal_pc_deconv_bwd", N, A, ptr(hp), ptr(hp_max), ptr(d_dec), ptr(ddec_max), ptr(Wv), ptr(Wa), ptr(d_hp),
          ptr(dhp_max), ptr(dWv), ptr(dbv), ptr(dWa), ptr(dba))


def pc_deconv_train(N, A, hp, Wv, bv, Wa, ba, action, target, mask, lam, grad_scale, loss, d_hp, dWv, dbv, dWa, dba,
                    dhp_max=None, hp_max=None, d_dec=None):
    """pc_deconv_fwd (training mode) + pc_deconv_bwd in one launch; d_dec stays on chip (optional output for inspection)."""
    _chk(hp, "f32", N * F2_DIM); _chk(Wv, "f32", 512); _chk(bv, "f32", 1); _chk(Wa, "f32", 512 * A); _chk(ba, "f32", A)
    _chk(action, "i32", N); _chk(target, "f32", N * PC_CELLS); _chk(mask, "i32", N); _chk(loss, "f32", 1)
    _chk(d_hp, "f32", N * F2_DIM); _chk(dWv, "f32", 512); _chk(dbv, "f32", 1); _chk(dWa, "f32", 512 * A); _chk(dba, "f32", A)
    _chk(dhp_max, "f32", 1, "dhp_max", optional=True); _chk(hp_max, "f32", 1, "hp_max", optional=True)
    _chk(d_dec, "f32", N * PC_CELLS * (1 + A), optional=True)
    hp_max = _absmax_of(hp, N, F2_DIM, F2_DIM, hp_max)
    _call("unreal_pc_deconv_train", N, A, ptr(hp), ptr(hp_max), ptr(Wv), ptr(bv), ptr(Wa), ptr(ba), ptr(action), ptr(target),
          ptr(mask), float(lam), float(grad_scale), ptr(loss), ptr(d_hp), ptr(dhp_max), ptr(dWv), ptr(dbv), ptr(dWa), ptr(dba),
          ptr(d_dec))


def axpy(alpha, x, y):
    """y += alpha * x (f32 device vectors)."""
    _chk(x, "f32"); _chk(y, "f32", x.numel())
    _call("unreal_axpy_f32", x.numel(), float(alpha), ptr(x), ptr(y))


def copy_(dst, src):
    """dst[:] = src for device tensors of one 4-byte dtype (f32 / i32), as a kernel of this library."""
    if dst.dtype != src.dtype or dst.dtype not in (torch.float32, torch.int32):
        raise ValueError("copy_: %s <- %s" % (dst.dtype, src.dtype))
    if not (dst.is_cuda and src.is_cuda and dst.is_contiguous() and src.is_contiguous()):
        raise ValueError("copy_ needs contiguous device tensors (no CPU fallback)")
    if dst.numel() != src.numel():
        raise ValueError("copy_: %d <- %d elements" % (dst.numel(), src.numel()))
    _call("unreal_copy_words", dst.numel(), ptr(src), ptr(dst))
    return dst


# ---- optimiser -----------------------------------------------------------------------------------
def grad_norm(grad, scratch, norm_out):
    _chk(grad, "f32"); _chk(scratch, "f32", 256); _chk(norm_out, "f32", 1)
    _call("unreal_grad_norm", ptr(grad), grad.numel(), ptr(scratch), ptr(norm_out))


def rmsprop_step(var, ms, mom, grad, lr, decay, momentum, eps, clip_norm, norm):
    n = var.numel()
    _chk(var, "f32"); _chk(ms, "f32", n); _chk(mom, "f32", n); _chk(grad, "f32", n)
    _chk(norm, "f32", 1, optional=True)
    _call("unreal_rmsprop_step", ptr(var), ptr(ms), ptr(mom), ptr(grad), n, float(lr), float(decay),
          float(momentum), float(eps), float(clip_norm), ptr(norm))
