"""Shared RMSProp with global-norm clipping over flat device buffers.

Mirrors /root/reference/train/rmsprop_applier.py: ctor 13-20, slots 38-43 (rms = 1, momentum = 0),
apply 83-93, clip 121.  The reference builds TF ops per variable; here one fused kernel walks the
flat parameter buffer (ops.rmsprop_step) after one deterministic norm reduction (ops.grad_norm)."""
import torch

from .. import ops


class RMSPropApplier(object):
    def __init__(self, learning_rate=None, decay=0.9, momentum=0.0, epsilon=1e-10, clip_norm=40.0,
                 device="cuda:0", name="RMSPropApplier"):
        self._name = name
        self._learning_rate = learning_rate
        self._decay, self._momentum, self._epsilon, self._clip_norm = decay, momentum, epsilon, clip_norm
        self._device = device
        self._slots = {}
        self.ms = self.mom = None
        self._scratch = self._norm = None

    def _create_slots(self, flat):
        if self.ms is None or self.ms.numel() != flat.numel():
            self.ms = torch.ones_like(flat)
            self.mom = torch.zeros_like(flat)
            self._scratch = torch.zeros(256, dtype=torch.float32, device=flat.device)
            self._norm = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self._slots = {"rms": self.ms, "momentum": self.mom}

    def get_slot(self, var, name):
        return self._slots.get(name)

    def step(self, flat_params, flat_grad, lr):
        """clip_by_global_norm + apply_rms_prop on the flat buffers; returns the (device) pre-clip norm."""
        self._create_slots(flat_params)
        ops.grad_norm(flat_grad, self._scratch, self._norm)
        ops.rmsprop_step(flat_params, self.ms, self.mom, flat_grad, lr, self._decay, self._momentum,
                         self._epsilon, self._clip_norm, self._norm)
        return self._norm

    def minimize_local(self, loss, global_var_list, local_var_list, thread_index=None):
        """Reference graph-build hook (rmsprop_applier.py:95-106).  Gradients here are produced by the
        hand-written backward kernels, so this only returns the (applier, norm handle) pair shape."""
        return self, self._norm
