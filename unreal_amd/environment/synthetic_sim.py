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


class SyntheticIndoorSim(SyntheticActorSim):
    """Stand-in for a MINOS RoomSimulator actor (indoor_environment.py:63-139): also returns a measurement vector
    ('objective') with every observation, and raw rewards that the wrapper divides by termination_time.  Raw rewards are
    multiples of termination_time / 8 so the scaled reward is exact in fp32."""

    def __init__(self, seed, objective_size=5, termination_time=50.0, **kw):
        SyntheticActorSim.__init__(self, seed, **kw)
        self.objective_size, self.termination_time = objective_size, termination_time

    def _meas(self):
        return self.rs.uniform(-1.0, 1.0, size=self.objective_size).astype(np.float32)

    def reset(self):
        return SyntheticActorSim.reset(self), self._meas()

    def step(self, action):
        """-> (obs uint8 or None when terminal, raw reward, terminal, measurements)"""
        self.t += 1
        u = self.rs.random_sample()
        unit = self.termination_time / 8.0
        reward = 4.0 * unit if u < self.big_reward_p else (unit if u < self.reward_p else (-unit if u < 3 * self.reward_p else 0.0))
        terminal = self.t >= self.episode_len
        obs = None if terminal else self._obs()
        return obs, reward, terminal, self._meas()


class SyntheticBatchIndoorSimulator(object):
    """B independent SyntheticIndoorSim actors behind the batched host-fed interface (with objectives)."""

    def __init__(self, batch, seed=5, objective_size=5, **kw):
        self.actors = [SyntheticIndoorSim(seed * 100003 + b, objective_size=objective_size, **kw) for b in range(batch)]
        self.B, self.objective_size = batch, objective_size
        self._obj = np.zeros((batch, objective_size), np.float32)

    def reset(self, mask=None):
        out = np.zeros((self.B, 84, 84, 3), np.uint8)
        for b, a in enumerate(self.actors):
            if mask is None or mask[b]:
                out[b], self._obj[b] = a.reset()
        return out, self._obj.copy()

    def step(self, actions, active=None):
        """-> frames, rewards (raw), terminals, objectives; where terminal: the post-reset observation and objective."""
        frames = np.zeros((self.B, 84, 84, 3), np.uint8)
        rewards = np.zeros(self.B, np.float32)
        terminals = np.zeros(self.B, np.int32)
        for b, a in enumerate(self.actors):
            if active is not None and not active[b]:
                continue
            obs, r, t, meas = a.step(int(actions[b]))
            rewards[b], terminals[b] = r, int(t)
            if t:
                frames[b], self._obj[b] = a.reset()       # the trainer's env.reset() after a terminal
            else:
                frames[b], self._obj[b] = obs, meas
        return frames, rewards, terminals, self._obj.copy()
