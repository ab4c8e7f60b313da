"""Oracle: global-norm clip + shared RMSProp.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/train/rmsprop_applier.py:38-43 (rms slot = 1, momentum slot = 0),
:83-93 (apply_rms_prop) and :121 (tf.clip_by_global_norm).  Pinned by the known-answer
arithmetic of train/rmsprop_applier_test.py:29-51 (tests/golden/rmsprop_known_answer.npz).
"""
import numpy as np


def clip_by_global_norm(grads, clip_norm, dtype=np.float32):
    """tf.clip_by_global_norm: g * clip_norm * min(1/norm, 1/clip_norm)."""
    sq = dtype(0)
    for g in grads:
        sq = sq + np.sum(np.square(g.astype(dtype)), dtype=dtype)
    norm = np.sqrt(sq)
    one = dtype(1.0)
    scale = dtype(clip_norm) * np.minimum(one / norm, one / dtype(clip_norm)) if norm > 0 else one
    return [g * scale for g in grads], norm


class OracleRMSProp(object):
    def __init__(self, decay=0.9, momentum=0.0, epsilon=1e-10, clip_norm=40.0, dtype=np.float32):
        self.decay, self.momentum, self.epsilon, self.clip_norm = decay, momentum, epsilon, clip_norm
        self.dtype = dtype
        self.ms = None
        self.mom = None

    def create_slots(self, params):
        """rmsprop_applier.py:38-43: the shared slots exist before any thread runs (graph-build time in the reference);
        creating them lazily inside step() raced when several threads made their first update at once."""
        if self.ms is None or self.mom is None:
            mom = [np.zeros_like(p, dtype=self.dtype) for p in params]
            ms = [np.ones_like(p, dtype=self.dtype) for p in params]
            self.mom, self.ms = mom, ms

    def step(self, params, grads, lr, clip=True):
        """In-place update of the list of numpy arrays `params`; returns the pre-clip norm."""
        dt = self.dtype
        self.create_slots(params)
        norm = dt(0)
        if clip:
            grads, norm = clip_by_global_norm(grads, self.clip_norm, dt)
        for p, g, ms, mom in zip(params, grads, self.ms, self.mom):
            g = g.astype(dt)
            ms += (g * g - ms) * dt(1.0 - self.decay)
            mom *= dt(self.momentum)
            mom += dt(lr) * g / np.sqrt(ms + dt(self.epsilon))
            p -= mom
        return norm
