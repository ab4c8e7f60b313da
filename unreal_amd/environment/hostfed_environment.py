"""Host-fed batched environment: CPU simulators -> pinned uint8 staging -> HBM replay ring.

Keeps the contract of /root/reference/environment/lab_environment.py:78-119 for every actor (frame = obs/255,
state unchanged and pixel change 0 on a terminal step, reset by the trainer) while the learner stays the
batched device path: frames are stored as uint8 and the encoder applies frame_scale = 1/255 on load.
`simulator` is any object with reset(mask) -> frames and step(actions, active) -> (frames, rewards, terminals)
(see synthetic_sim.SyntheticBatchSimulator); DeepMind Lab itself is not in the image."""
import numpy as np
import torch

from .. import ops


class HostFedEnvironment(object):
    def __init__(self, simulator, batch, history_size, device="cuda:0", action_size=6, clip_reward=True,
                 frame_max=255.0):
        self.B, self.sim = batch, simulator
        self.action_size = action_size
        self.clip_reward = clip_reward
        self.pc_denom = 48.0 * frame_max
        self.device = torch.device(device)
        self.ring = ops.Ring(batch, history_size, self.device)
        self._h_frames = torch.empty((batch, 84, 84, 3), dtype=torch.uint8).pin_memory()
        self._h_rewards = torch.empty(batch, dtype=torch.float32).pin_memory()
        self._h_terminals = torch.empty(batch, dtype=torch.int32).pin_memory()
        self._staged = torch.empty(batch * ops.FRAME_BYTES, dtype=torch.uint8, device=self.device)
        self._rewards = torch.empty(batch, dtype=torch.float32, device=self.device)
        self._terminals = torch.empty(batch, dtype=torch.int32, device=self.device)
        self.reset()

    def get_action_size(self):
        return self.action_size

    def _stage(self, frames):
        self._h_frames.copy_(torch.from_numpy(np.ascontiguousarray(frames)))
        self._staged.copy_(self._h_frames.view(-1), non_blocking=True)

    def reset(self, mask=None):
        m = None if mask is None else mask.cpu().numpy()
        self._stage(self.sim.reset(m))
        ops.hostfed_reset(self.ring, self._staged, mask)

    def process(self, actions, active=None, out_reward=None, out_terminal=None, reset_on_terminal=True,
                track_score=False):
        a = actions.cpu().numpy()                       # the simulators live on the host: one D2H per step
        act = None if active is None else active.cpu().numpy()
        frames, rewards, terminals = self.sim.step(a, act)
        self._stage(frames)
        self._h_rewards.copy_(torch.from_numpy(rewards))
        self._h_terminals.copy_(torch.from_numpy(terminals))
        self._rewards.copy_(self._h_rewards, non_blocking=True)
        self._terminals.copy_(self._h_terminals, non_blocking=True)
        ops.hostfed_step(self.ring, self._staged, actions, self._rewards, self._terminals, active, out_reward,
                         out_terminal, reset_on_terminal, track_score, self.clip_reward, self.pc_denom)

    def stop(self):
        pass
