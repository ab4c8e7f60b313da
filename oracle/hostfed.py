"""Oracle: host-fed environment with the DeepMind-Lab wrapper contract.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/environment/lab_environment.py:78-119 around a per-actor simulator object with
reset() -> obs uint8 and step(action) -> (obs | None, reward, terminal)  (the child-process protocol of
lab_environment.py:16-49).  Frames are float32 obs/255 like `_preprocess_frame` (:99-102); on a terminal step the
state is the previous state, so the pixel change is 0.  The simulator itself (deepmind_lab) is absent from the
image: tests drive this with unreal_amd.environment.synthetic_sim.SyntheticActorSim -- parity unpinned by the
reference for this path (no fixtures exist for Lab)."""
import numpy as np

from .maze import calc_pixel_change


class OracleLabEnv(object):
    def __init__(self, sim, action_size=6):
        self.sim = sim
        self.action_size = action_size
        self.reset()

    @staticmethod
    def _preprocess_frame(image):
        return image.astype(np.float32) / 255.0

    def reset(self):
        obs = self.sim.reset()
        self.last_state = {'image': self._preprocess_frame(obs)}
        self.last_action = 0
        self.last_reward = 0

    def process(self, action, flag=0):
        obs, reward, terminal = self.sim.step(int(action))
        if not terminal:
            state = {'image': self._preprocess_frame(obs)}
        else:
            state = self.last_state
        pc = calc_pixel_change(state['image'], self.last_state['image'])
        self.last_state = state
        self.last_action = int(action)
        self.last_reward = reward
        return state, reward, terminal, pc

    def stop(self):
        pass
