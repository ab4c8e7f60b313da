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
import numpy as np
import torch

from .. import ops


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
        self._h_frames.copy_(torch.from_numpy(np.ascontiguousarray(frames)))
        self._staged.copy_(self._h_frames.view(-1), non_blocking=True)

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

    def stop(self):
        pass
