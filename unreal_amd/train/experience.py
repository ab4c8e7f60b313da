"""Experience replay as an HBM ring buffer.

Mirrors /root/reference/train/experience.py: ExperienceFrame 10-46, Experience 48-153.  The deque
of Python objects becomes per-actor ring arrays on the device (layout: include/unreal_hip.h, written
by the environment kernel itself); sampling runs as kernels for all actors at once:
  sample_sequence(L)     -> ops.replay_sample_seq  (start draw in [0, H-L-1), terminal rules :100-118)
  sample_rp_sequence()   -> ops.replay_sample_rp   (zero/neg vs positive buckets :121-153)
The 'pos'/'neg' index deques are not stored: they are exactly the frames of the window
[top+3, count) with reward > 0 / <= 0 (invariant of add_frame :72-93), selected by rank on device."""
import numpy as np
import torch

from .. import ops


class ExperienceFrame(object):
    """Host-side record with the reference's helpers (used by batch-1 callers and tests)."""

    def __init__(self, state, reward, action, terminal, pixel_change, last_action, last_reward):
        self.state, self.reward, self.action, self.terminal = state, reward, action, terminal
        self.pixel_change, self.last_action, self.last_reward = pixel_change, last_action, last_reward

    def get_last_action_reward(self, action_size):
        return ExperienceFrame.concat_action_and_reward(self.last_action, action_size, self.last_reward, self.state)

    def get_action_reward(self, action_size):
        return ExperienceFrame.concat_action_and_reward(self.action, action_size, self.reward, self.state)

    @staticmethod
    def concat_action_and_reward(action, action_size, reward, state):
        v = np.zeros([action_size + 1])
        v[action] = 1.0
        v[-1] = float(reward)
        objective = state.get('objective') if isinstance(state, dict) else None
        return np.concatenate((v, objective)) if objective is not None else v


class Experience(object):
    def __init__(self, history_size, random_state=None, ring=None):
        if ring is None:
            raise ValueError("the device Experience is a view of an environment ring (ops.Ring)")
        self._history_size = history_size
        self.ring = ring
        self.random_state = random_state

    def is_full(self):
        return int(self.ring.count.min().item()) >= self._history_size

    def get_debug_string(self):
        c = self.ring.count
        return "{} actors, {}..{} frames".format(self.ring.B, int(c.min()), int(c.max()))

    def sample_sequence(self, sequence_size, start_draw, seq_idx, seq_len):
        ops.replay_sample_seq(self.ring, sequence_size, start_draw, seq_idx, seq_len)
        return seq_idx, seq_len

    def sample_rp_sequence(self, coin, u, rp_idx, rp_class):
        ops.replay_sample_rp(self.ring, coin, u, rp_idx, rp_class)
        return rp_idx, rp_class
