"""Host-fed batched environment: CPU simulators -> pinned uint8 staging -> HBM replay ring.

Keeps the contract of /root/reference/environment/lab_environment.py:78-119 for every actor (frame = obs/255,
state unchanged and pixel change 0 on a terminal step, reset by the trainer) while the learner stays the
batched device path: frames are stored as uint8 and the encoder applies frame_scale = 1/255 on load.
`simulator` is any object with reset(mask) -> frames and step(actions, active) -> (frames, rewards, terminals)
(see synthetic_sim.SyntheticBatchSimulator); DeepMind Lab itself is not in the image.

With objective_size > 0 it is the MINOS wrapper contract of /root/reference/environment/indoor_environment.py:63-139
instead: the simulator also returns a measurement vector per actor (reset -> (frames, objectives),
step -> (frames, rewards, terminals, objectives)), stored beside the frame in the ring and concatenated into the LSTM
input; rewards are divided by termination_time (:111) and not clipped."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import ops

# The simulators' frames reach the pinned staging by a host memcpy of 21 KB per actor and step -- the largest host
# cost of the path.  It is cut into row blocks copied by a few threads (torch's copy releases the GIL).
# UNREAL_COPY_THREADS overrides the count (default: up to 8; a GPU's share of the host is 16 cores on the test boxes).
_COPY_THREADS = int(os.environ.get("UNREAL_COPY_THREADS", 0)) or max(1, min(8, (os.cpu_count() or 1) // 2))
_COPY_POOL = ThreadPoolExecutor(_COPY_THREADS) if _COPY_THREADS > 1 else None
_COPY_IMPL = os.environ.get("UNREAL_COPY_IMPL", "numpy")


def _stage_frames(dst, frames):
    """dst (pinned uint8 [n,84,84,3]) <- frames (numpy uint8 [n,84,84,3]), in parallel row blocks.
    Plain memcpy per block (numpy releases the GIL): torch's copy_ starts its own intra-op team inside every pool
    thread, and 8 pool threads x that team ran at a third of the rate of 2 (UNREAL_COPY_IMPL=torch restores it)."""
    n = dst.shape[0]
    if _COPY_IMPL == "torch":
        src = torch.from_numpy(np.ascontiguousarray(frames))
        if _COPY_POOL is None or n < 4 * _COPY_THREADS:
            dst.copy_(src)
            return
        step = (n + _COPY_THREADS - 1) // _COPY_THREADS
        futs = [_COPY_POOL.submit(dst[a:a + step].copy_, src[a:a + step]) for a in range(0, n, step)]
        for f in futs:
            f.result()
        return
    d = dst.numpy()
    src = np.asarray(frames)
    if _COPY_POOL is None or n < 4 * _COPY_THREADS:
        np.copyto(d, src)
        return
    step = (n + _COPY_THREADS - 1) // _COPY_THREADS
    futs = [_COPY_POOL.submit(np.copyto, d[a:a + step], src[a:a + step]) for a in range(0, n, step)]
    for f in futs:
        f.result()


# Staging and PCIe as a pipeline (round 4 experiment, UNREAL_STAGE_CHUNKS > 1): the frames of a (half-)batch go to the pinned
# buffer in pieces and the H2D copy of a piece is issued as soon as the piece is staged, so that the copy engine moves piece i
# while the host threads stage piece i + 1.  Measured (tools/bench_hostfed.py, profiles/r04_hostfed.md): SLOWER -- 4096 actors
# 1.37 M env-steps/s with one piece, 0.89 M with four, 0.63 M with eight (per-piece thread-pool joins and copy submissions, and
# the DMA engine reading the pinned buffer while eight threads write it share the host's memory bandwidth).  Default 1 = one
# staging copy, one H2D copy per (half-)batch and step.
STAGE_CHUNKS = int(os.environ.get("UNREAL_STAGE_CHUNKS", 1))


def _stage_and_copy(h_frames, frames, staged):
    """h_frames (pinned [n,84,84,3]) <- frames, staged (device bytes) <- h_frames, piece by piece on the current stream."""
    n = h_frames.shape[0]
    chunks = max(1, min(STAGE_CHUNKS, n // 64))
    step = (n + chunks - 1) // chunks
    for a in range(0, n, step):
        b = min(n, a + step)
        _stage_frames(h_frames[a:b], frames[a:b])
        staged[a * ops.FRAME_BYTES:b * ops.FRAME_BYTES].copy_(h_frames[a:b].view(-1), non_blocking=True)


class HostFedEnvironment(object):
    def __init__(self, simulator, batch, history_size, device="cuda:0", action_size=6, clip_reward=True,
                 frame_max=255.0, objective_size=0, reward_divisor=1.0):
        self.B, self.sim = batch, simulator
        self.action_size = action_size
        self.clip_reward = clip_reward
        self.objective_size = int(objective_size)
        self.reward_divisor = float(reward_divisor)
        self.pc_denom = 48.0 * frame_max
        self.frame_scale = 1.0 / frame_max          # lab_environment.py:99-102: state = obs / 255
        self.device = torch.device(device)
        self.ring = ops.Ring(batch, history_size, self.device, objective_size=self.objective_size)
        if self.objective_size:
            self._h_obj = torch.empty((batch, self.objective_size), dtype=torch.float32).pin_memory()
            self._obj = torch.empty(batch * self.objective_size, dtype=torch.float32, device=self.device)
        self._h_frames = torch.empty((batch, 84, 84, 3), dtype=torch.uint8).pin_memory()
        self._h_rewards = torch.empty(batch, dtype=torch.float32).pin_memory()
        self._h_terminals = torch.empty(batch, dtype=torch.int32).pin_memory()
        self._staged = torch.empty(batch * ops.FRAME_BYTES, dtype=torch.uint8, device=self.device)
        self._rewards = torch.empty(batch, dtype=torch.float32, device=self.device)
        self._terminals = torch.empty(batch, dtype=torch.int32, device=self.device)
        self._h2d_done = None
        self.reset()

    def get_action_size(self):
        return self.action_size

    def _wait_staging(self):
        """The pinned staging buffers may be rewritten only after the H2D copies that read them have finished."""
        if self._h2d_done is not None:
            self._h2d_done.synchronize()

    def _mark_staging(self):
        if self._h2d_done is None:
            self._h2d_done = torch.cuda.Event()
        self._h2d_done.record()

    def _stage(self, frames):
        _stage_and_copy(self._h_frames, frames, self._staged)

    def _stage_objective(self, objectives, active):
        self._h_obj.copy_(torch.from_numpy(np.ascontiguousarray(objectives, dtype=np.float32)))
        self._obj.copy_(self._h_obj.view(-1), non_blocking=True)
        ops.objective_put(self.ring, self._obj, active)       # into the slot the frame just went to

    def reset(self, mask=None):
        m = None if mask is None else mask.cpu().numpy()
        out = self.sim.reset(m)
        frames, objectives = out if self.objective_size else (out, None)
        self._wait_staging()
        self._stage(frames)
        ops.hostfed_reset(self.ring, self._staged, mask)
        if self.objective_size:
            self._stage_objective(objectives, mask)
        self._mark_staging()

    def process(self, actions, active=None, out_reward=None, out_terminal=None, reset_on_terminal=True,
                track_score=False):
        a = actions.cpu().numpy()                       # the simulators live on the host: one D2H per step
        act = None if active is None else active.cpu().numpy()
        out = self.sim.step(a, act)
        frames, rewards, terminals = out[:3]
        if self.reward_divisor != 1.0:                  # indoor_environment.py:111
            rewards = (rewards.astype(np.float64) / self.reward_divisor).astype(np.float32)
        self._wait_staging()
        self._stage(frames)
        self._h_rewards.copy_(torch.from_numpy(rewards))
        self._h_terminals.copy_(torch.from_numpy(terminals))
        self._rewards.copy_(self._h_rewards, non_blocking=True)
        self._terminals.copy_(self._h_terminals, non_blocking=True)
        ops.hostfed_step(self.ring, self._staged, actions, self._rewards, self._terminals, active, out_reward,
                         out_terminal, reset_on_terminal, track_score, self.clip_reward, self.pc_denom)
        if self.objective_size:
            self._stage_objective(out[3], active)
        self._mark_staging()

    # ---- half-batch interface: host phase of one part overlaps the device phase of the other ---------
    def enable_parts(self, n_parts=2):
        """Split the actors into `n_parts` contiguous parts, each with its own HIP stream, pinned staging and ring view,
        so that the Trainer can alternate: while the host simulates / stages part k, the device ingests and forwards the
        other part.  Returns the part boundaries [(b0, b1), ...]."""
        if self.B % n_parts:
            raise ValueError("%d actors do not split into %d parts" % (self.B, n_parts))
        Bp = self.B // n_parts
        self.parts = []
        for k in range(n_parts):
            b0, b1 = k * Bp, (k + 1) * Bp
            part = dict(b0=b0, b1=b1, ring=ops.ring_view(self.ring, b0, b1), stream=torch.cuda.Stream(device=self.device),
                        h_frames=torch.empty((Bp, 84, 84, 3), dtype=torch.uint8).pin_memory(),
                        h_rewards=torch.empty(Bp, dtype=torch.float32).pin_memory(),
                        h_terminals=torch.empty(Bp, dtype=torch.int32).pin_memory(),
                        h_actions=torch.empty(Bp, dtype=torch.int32).pin_memory(),
                        h_active=torch.empty(Bp, dtype=torch.int32).pin_memory(),
                        staged=self._staged[b0 * ops.FRAME_BYTES:b1 * ops.FRAME_BYTES],
                        rewards=self._rewards[b0:b1], terminals=self._terminals[b0:b1], h2d_done=None,
                        act_ready=torch.cuda.Event())
            if self.objective_size:
                part["h_obj"] = torch.empty((Bp, self.objective_size), dtype=torch.float32).pin_memory()
                part["obj"] = self._obj[b0 * self.objective_size:b1 * self.objective_size]
            self.parts.append(part)
        self._act_full = np.zeros(self.B, np.int32)
        self._mask_full = np.zeros(self.B, np.int32)
        return [(p["b0"], p["b1"]) for p in self.parts]

    def part_request_actions(self, k, actions, active):
        """On part k's stream: start the D2H copy of its drawn actions (and active flags) into pinned memory."""
        p = self.parts[k]
        p["h_actions"].copy_(actions, non_blocking=True)
        if active is not None:
            p["h_active"].copy_(active, non_blocking=True)
        p["act_ready"].record()

    def part_host_step(self, k, has_active):
        """Host phase of part k: wait for its actions, step ITS simulators only, fill its pinned staging."""
        p = self.parts[k]
        p["act_ready"].synchronize()
        if p["h2d_done"] is not None:
            p["h2d_done"].synchronize()                # the previous H2D copies out of this staging have finished
        b0, b1 = p["b0"], p["b1"]
        self._act_full[b0:b1] = p["h_actions"].numpy()
        self._mask_full[:] = 0
        self._mask_full[b0:b1] = p["h_active"].numpy() if has_active else 1
        out = self.sim.step(self._act_full, self._mask_full)
        frames, rewards, terminals = out[:3]
        rewards = rewards[b0:b1]
        if self.reward_divisor != 1.0:                  # indoor_environment.py:111
            rewards = (rewards.astype(np.float64) / self.reward_divisor).astype(np.float32)
        with torch.cuda.stream(p["stream"]):           # H2D of every staged piece starts at once, on the part's own stream
            _stage_and_copy(p["h_frames"], frames[b0:b1], p["staged"])
        p["h_rewards"].copy_(torch.from_numpy(np.ascontiguousarray(rewards, dtype=np.float32)))
        p["h_terminals"].copy_(torch.from_numpy(np.ascontiguousarray(terminals[b0:b1], dtype=np.int32)))
        if self.objective_size:
            p["h_obj"].copy_(torch.from_numpy(np.ascontiguousarray(out[3][b0:b1], dtype=np.float32)))

    def part_ingest(self, k, actions, active, out_reward, out_terminal, reset_on_terminal=True, track_score=False):
        """On part k's stream: H2D of the staged part + the ring commit kernel for its actors."""
        p = self.parts[k]                              # (the frames' H2D copies were issued by part_host_step, piece by piece)
        p["rewards"].copy_(p["h_rewards"], non_blocking=True)
        p["terminals"].copy_(p["h_terminals"], non_blocking=True)
        ops.hostfed_step(p["ring"], p["staged"], actions, p["rewards"], p["terminals"], active, out_reward, out_terminal,
                         reset_on_terminal, track_score, self.clip_reward, self.pc_denom)
        if self.objective_size:
            p["obj"].copy_(p["h_obj"].view(-1), non_blocking=True)
            ops.objective_put(p["ring"], p["obj"], active)
        if p["h2d_done"] is None:
            p["h2d_done"] = torch.cuda.Event()
        p["h2d_done"].record()

    def stop(self):
        pass
