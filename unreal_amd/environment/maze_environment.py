"""Maze environment on the device (/root/reference/environment/maze_environment.py:10-128).

`BatchedMazeEnvironment` steps B mazes with one kernel launch and writes frames straight into the
replay ring; `MazeEnvironment` is the reference's batch-1 object surface over the same kernels
(`process(action) -> (image, reward, terminal, pixel_change)`, `reset`, `last_state` dict with 'image',
`last_action`, `last_reward`), with the adapter semantics of SURVEY H1 (`flag` ignored)."""
import numpy as np
import torch

from .. import ops
from . import environment


class BatchedMazeEnvironment(object):
    ACTION_SIZE = 4
    frame_scale = 1.0          # ring bytes are the pixel values themselves (0 / 1)

    def __init__(self, batch, history_size, device="cuda:0"):
        self.B = batch
        self.ring = ops.Ring(batch, history_size, torch.device(device))
        self.reset()

    def view(self, b0, b1):
        """The environments [b0, b1) as a batched environment of their own (shares the ring memory).  `base_actor` = b0:
        frame indices the view's rollout_step prepares are indices into THIS environment's ring."""
        v = object.__new__(BatchedMazeEnvironment)
        v.B, v.ring = b1 - b0, ops.ring_view(self.ring, b0, b1)
        v.base_actor = b0
        return v

    @staticmethod
    def get_action_size():
        return 4

    def reset(self, mask=None):
        ops.maze_reset(self.ring, mask)

    def process(self, actions, active=None, out_reward=None, out_terminal=None, reset_on_terminal=True,
                track_score=False):
        ops.maze_step(self.ring, actions, active, out_reward, out_terminal, reset_on_terminal, track_score)

    def rollout_step(self, actions, out_reward, out_terminal, active, active_log_t, n_steps, terminal_end,
                     index_parent=False, **nxt):
        """process() + the rollout loop's bookkeeping (+ the next step's frame indices / LSTM-input columns) fused.
        `index_parent` (views only): the prepared frame indices address the ring this view was cut from."""
        ops.maze_rollout_step(self.ring, actions, out_reward, out_terminal, active, active_log_t, n_steps, terminal_end,
                              base_actor=getattr(self, "base_actor", 0) if index_parent else 0, **nxt)

    def policy_rollout_step(self, net, feat, ld, u, pi_out, v_out, actions, out_reward, out_terminal, active, active_log_t,
                            n_steps, terminal_end, index_parent=False, **nxt):
        """The policy head + action draw of `net` on the feature rows `feat` and rollout_step() in one launch."""
        p = net.p
        ops.maze_policy_rollout_step(self.ring, feat, ld, p["W_base_fc_p"], p["b_base_fc_p"], p["W_base_fc_v"],
                                     p["b_base_fc_v"], u, pi_out, v_out, actions, out_reward, out_terminal, active,
                                     active_log_t, n_steps, terminal_end,
                                     base_actor=getattr(self, "base_actor", 0) if index_parent else 0, **nxt)

    def stop(self):
        pass


class MazeEnvironment(environment.Environment):
    @staticmethod
    def get_action_size():
        return 4

    def __init__(self, device="cuda:0"):
        environment.Environment.__init__(self)
        self._env = BatchedMazeEnvironment(1, 2, device)
        self._a = torch.zeros(1, dtype=torch.int32, device=device)
        self._r = torch.zeros(1, dtype=torch.float32, device=device)
        self._t = torch.zeros(1, dtype=torch.int32, device=device)
        self.reset()

    def _image(self):
        ring = self._env.ring
        slot = int(ring.count.cpu()[0]) % ring.H1
        fr = ring.frames[slot * ops.FRAME_BYTES:(slot + 1) * ops.FRAME_BYTES]
        return fr.cpu().numpy().reshape(84, 84, 3).astype(np.float64)

    def reset(self):
        self._env.reset()
        self.last_state = {'image': self._image()}
        self.last_action = 0
        self.last_reward = 0

    def process(self, action, flag=0):
        ring = self._env.ring
        self._a[0] = int(action)
        slot = int(ring.count.cpu()[0]) % ring.H1
        self._env.process(self._a, None, self._r, self._t, reset_on_terminal=False)
        image = self._image()
        reward = int(self._r.cpu()[0])
        terminal = bool(self._t.cpu()[0])
        pc = ring.r_pc[slot * ops.PC_CELLS:(slot + 1) * ops.PC_CELLS].cpu().numpy().reshape(20, 20)
        self.last_state = {'image': image}
        self.last_action = int(action)
        self.last_reward = reward
        self._last_full_state = {"success": terminal}
        return image, reward, terminal, pc
