"""Synthetic stand-in for an external CPU simulator (DeepMind Lab / MINOS are not in the image).

SURVEY 8d, config 4: uint8 [84,84,3] frames (uniform 0..255), rewards Bernoulli(p)*{+1} with an occasional
large reward to exercise clipping, a terminal every `episode_len` steps (Lab: 60 s at 60 fps / 4-frame repeat =
900 steps).  Every actor owns its own RandomState so a batch and B single actors produce identical streams.
This is a DATA GENERATOR with the frame/reward/terminal contract of lab_environment.py:16-49, not a model of Lab."""
import numpy as np


class SyntheticActorSim(object):
    def __init__(self, seed, episode_len=900, reward_p=0.01, big_reward_p=0.002):
        self.rs = np.random.RandomState(seed)
        self.episode_len, self.reward_p, self.big_reward_p = episode_len, reward_p, big_reward_p
        self.t = 0

    def _obs(self):
        return self.rs.randint(0, 256, size=(84, 84, 3)).astype(np.uint8)

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, action):
        """-> (obs uint8 or None when terminal, reward, terminal)   (lab_environment.py:26-41)"""
        self.t += 1
        u = self.rs.random_sample()
        reward = 10.0 if u < self.big_reward_p else (1.0 if u < self.reward_p else 0.0)
        terminal = self.t >= self.episode_len
        obs = None if terminal else self._obs()
        return obs, reward, terminal


class SyntheticBatchSimulator(object):
    """B independent actor simulators behind the batched host-fed interface."""

    def __init__(self, batch, seed=4, **kw):
        self.actors = [SyntheticActorSim(seed * 100003 + b, **kw) for b in range(batch)]
        self.B = batch

    def reset(self, mask=None):
        out = np.zeros((self.B, 84, 84, 3), np.uint8)
        for b, a in enumerate(self.actors):
            if mask is None or mask[b]:
                out[b] = a.reset()
        return out

    def step(self, actions, active=None):
        """-> frames uint8 [B,84,84,3] (post-reset observation where terminal), rewards f32 [B], terminals i32 [B]"""
        frames = np.zeros((self.B, 84, 84, 3), np.uint8)
        rewards = np.zeros(self.B, np.float32)
        terminals = np.zeros(self.B, np.int32)
        for b, a in enumerate(self.actors):
            if active is not None and not active[b]:
                continue
            obs, r, t = a.step(int(actions[b]))
            rewards[b], terminals[b] = r, int(t)
            frames[b] = a.reset() if t else obs          # the trainer's env.reset() after a terminal
        return frames, rewards, terminals
