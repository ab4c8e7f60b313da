"""Oracle: experience replay (pure Python / numpy).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/train/experience.py:10-153.  Written as an absolute-index ring
(not a deque) so that its state maps 1:1 onto the device ring.  Pinned by
tests/golden/replay_traces.npz.
"""
import numpy as np


def concat_action_and_reward(action, action_size, reward, objective=None):
    """experience.py:35-46."""
    v = np.zeros([action_size + 1])
    v[action] = 1.0
    v[-1] = float(reward)
    if objective is not None:
        return np.concatenate((v, objective))
    return v


class Frame(object):
    __slots__ = ("state", "reward", "action", "terminal", "pixel_change", "last_action", "last_reward")

    def __init__(self, state, reward, action, terminal, pixel_change, last_action, last_reward):
        self.state = state
        self.reward = reward
        self.action = action
        self.terminal = terminal
        self.pixel_change = pixel_change
        self.last_action = last_action
        self.last_reward = last_reward

    def get_last_action_reward(self, action_size):
        return concat_action_and_reward(self.last_action, action_size, self.last_reward, self.state.get('objective'))

    def get_action_reward(self, action_size):          # NB the objective of THIS frame's state (experience.py:27-32)
        return concat_action_and_reward(self.action, action_size, self.reward, self.state.get('objective'))


def clip_frame(frame):
    """experience_lab_ver.py:14,18: reward and last_reward are clipped when the frame is created."""
    frame.reward = float(np.clip(frame.reward, -1, 1))
    frame.last_reward = float(np.clip(frame.last_reward, -1, 1))
    return frame


class OracleExperience(object):
    def __init__(self, history_size, random_state=None, lab_ver=False):
        # lab_ver: upstream replay of /root/reference/train/experience_lab_ver.py (zero / non-zero reward buckets
        # :53-55,76-80,124-141; rewards are clipped to [-1, 1] when the frame is created :14,18 -- see clip_frame)
        self.lab_ver = lab_ver
        self.H = history_size
        self.frames = {}            # absolute index -> Frame (only the live window is kept)
        self.count = 0              # frames ever appended (absolute index of the next frame)
        self.random_state = random_state

    # -- state mirrors of the reference attributes ---------------------------------------------
    @property
    def top(self):                  # _top_frame_index
        return max(0, self.count - self.H)

    def __len__(self):
        return min(self.count, self.H)

    def is_full(self):              # :95-96
        return len(self) >= self.H

    def bucket(self, positive):
        """Contents of _pos_reward_indices / _neg_reward_indices.

        Invariant of experience.py:72-93: the union of both deques is exactly the absolute
        indices max(3, top+3) .. count-1; `reward > 0` selects the 'pos' deque, everything
        else (zero AND negative rewards) goes to 'neg' in this fork (:77-80).
        """
        lo = self.top + 3
        if self.lab_ver:     # 'positive' selects the non-zero bucket, the other one is the zero bucket
            return [i for i in range(lo, self.count) if (self.frames[i].reward != 0) == positive]
        return [i for i in range(lo, self.count) if (self.frames[i].reward > 0) == positive]

    # -- mutation ------------------------------------------------------------------------------
    def add_frame(self, frame):     # :63-93
        if frame.terminal and self.count > 0 and self.frames[self.count - 1].terminal:
            return False            # successive terminal frame is discarded
        self.frames[self.count] = frame
        self.count += 1
        self.frames.pop(self.count - self.H - 1, None)
        return True

    # -- sampling ------------------------------------------------------------------------------
    def sequence_from_start(self, start_pos, sequence_size):
        """:100-118 after the randint draw: returns the deque positions taken."""
        if self.frames[self.top + start_pos].terminal:
            start_pos += 1
        out = []
        for i in range(sequence_size):
            f = self.frames[self.top + start_pos + i]
            out.append(self.top + start_pos + i)
            if f.terminal:
                break
        return out

    def sample_sequence(self, sequence_size):
        start = self.random_state.randint(0, self.H - sequence_size - 1)
        return [self.frames[i] for i in self.sequence_from_start(start, sequence_size)]

    def rp_from_draws(self, coin, pick):
        """:121-153 with the two draws made explicit.  `pick` is either an int rank (the
        reference's randint(len(bucket))) or a callable n -> rank."""
        from_neg = (coin == 0)
        pos = self.bucket(True)
        neg = self.bucket(False)
        if len(pos) == 0:
            from_neg = True
        elif len(neg) == 0:
            from_neg = False
        b = neg if from_neg else pos
        rank = pick(len(b)) if callable(pick) else pick
        end = b[rank]
        return [end - 3 + i for i in range(4)]

    def sample_rp_sequence(self):
        coin = self.random_state.randint(2)
        idx = self.rp_from_draws(coin, lambda n: self.random_state.randint(n))
        return [self.frames[i] for i in idx]
