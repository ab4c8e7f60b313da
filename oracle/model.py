"""Oracle: UnrealModel forward graph + losses on PyTorch-CPU.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/model/model.py (vanilla encoder, segnet_mode == 0):
  encoder 281-289, _base_fcn_layer 305-318, _base_lstm_layer 321-355, policy/value 358-377,
  _pc_deconv_layers 411-443, vr 446-470, rp 473-488, losses 490-598, initialisers 31-42/752-783.
TensorFlow is not available, so this is written from the source plus TF-1.x semantics:
  * tf.nn.conv2d NHWC/HWIO VALID; tf.nn.conv2d_transpose with filter [kh,kw,out_c,in_c]
  * BasicLSTMCell(256): kernel rows = [input ; h], gate order i, j, f, o, forget_bias = 1.0,
    c' = c*sigmoid(f+1) + sigmoid(i)*tanh(j), h' = tanh(c')*sigmoid(o); state tuple = (c, h)
  * tf.nn.l2_loss(x) = sum(x**2)/2
PARITY UNPINNED by the reference for everything in this file (no TF, no golden tensors in the
reference); cross-checked by tests/test_oracle_model.py (independent numpy forward, fp64 finite
differences, variable-count spec of model/model_test.py:14-58).

Parameters are kept in TensorFlow layouts so a flat vector is interchangeable with the
reference's `get_vars()` order.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


def param_spec(action_size, objective_size=0, use_lstm=True, use_pixel_change=True,
               use_value_replay=True, use_reward_prediction=True):
    """(name, shape, fan_in) in TF variable-creation order (model.py:106-136)."""
    A = action_size
    lstm_in = 256 + A + 1 + objective_size
    spec = [
        ("W_base_conv1", (8, 8, 3, 16), 3 * 8 * 8), ("b_base_conv1", (16,), 3 * 8 * 8),
        ("W_base_conv2", (4, 4, 16, 32), 16 * 4 * 4), ("b_base_conv2", (32,), 16 * 4 * 4),
        ("W_base_fc1", (2592, 256), 2592), ("b_base_fc1", (256,), 2592),
    ]
    if use_lstm:
        spec += [("lstm_kernel", (lstm_in + 256, 1024), None), ("lstm_bias", (1024,), 0)]
    spec += [
        ("W_base_fc_p", (256, A), 256), ("b_base_fc_p", (A,), 256),
        ("W_base_fc_v", (256, 1), 256), ("b_base_fc_v", (1,), 256),
    ]
    if use_pixel_change:
        spec += [
            ("W_pc_fc1", (256, 2592), 256), ("b_pc_fc1", (2592,), 256),
            # deconv fan_in uses weight_shape[3] = 32 (model.py:771-773)
            ("W_pc_deconv_v", (4, 4, 1, 32), 32 * 4 * 4), ("b_pc_deconv_v", (1,), 32 * 4 * 4),
            ("W_pc_deconv_a", (4, 4, A, 32), 32 * 4 * 4), ("b_pc_deconv_a", (A,), 32 * 4 * 4),
        ]
    if use_reward_prediction:
        spec += [("W_rp_fc1", (7776, 3), 7776), ("b_rp_fc1", (3,), 7776)]
    return spec


def init_params(action_size, objective_size=0, use_lstm=True, use_pixel_change=True,
                use_value_replay=True, use_reward_prediction=True, seed=0, dtype=torch.float32):
    """U(+-1/sqrt(fan_in)) for W and b (model.py:31-42); LSTM kernel glorot_uniform, bias 0."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape, fan_in in param_spec(action_size, objective_size, use_lstm, use_pixel_change,
                                          use_value_replay, use_reward_prediction):
        if fan_in is None:        # glorot_uniform
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            v = rs.uniform(-lim, lim, size=shape)
        elif fan_in == 0:
            v = np.zeros(shape)
        else:
            d = 1.0 / math.sqrt(fan_in)
            v = rs.uniform(-d, d, size=shape)
        out[name] = torch.tensor(v, dtype=dtype)
    return out


def encoder(x, p):
    """x: [N,84,84,3] -> [N,9,9,32] (model.py:281-289)."""
    xin = x.permute(0, 3, 1, 2)
    h1 = F.relu(F.conv2d(xin, p["W_base_conv1"].permute(3, 2, 0, 1), p["b_base_conv1"], stride=4))
    h2 = F.relu(F.conv2d(h1, p["W_base_conv2"].permute(3, 2, 0, 1), p["b_base_conv2"], stride=2))
    return h1.permute(0, 2, 3, 1), h2.permute(0, 2, 3, 1)


def fc1(conv_out, p):
    flat = conv_out.reshape(conv_out.shape[0], 2592)                   # NHWC flatten (model.py:331)
    return F.relu(flat @ p["W_base_fc1"] + p["b_base_fc1"])


def lstm_unroll(x_seq, c0, h0, p):
    """dynamic_rnn over time with batch 1 (model.py:339-353). x_seq: [T, in]; c0,h0: [256]."""
    W, b = p["lstm_kernel"], p["lstm_bias"]
    c, h = c0, h0
    outs = []
    for t in range(x_seq.shape[0]):
        g = torch.cat([x_seq[t], h]) @ W + b
        i, j, f, o = g[0:256], g[256:512], g[512:768], g[768:1024]
        c = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        h = torch.tanh(c) * torch.sigmoid(o)
        outs.append(h)
    return torch.stack(outs), (c, h)


def trunk(x, lar, p, use_lstm, state):
    """Shared encoder + fc (+ LSTM).  Returns features [N,256] and the final LSTM state."""
    _, h2 = encoder(x, p)
    f = fc1(h2, p)
    if not use_lstm:
        return f, None                                 # FF mode drops `lar` (model.py:317-318)
    xin = torch.cat([f, lar], 1)
    if state is None:
        z = torch.zeros(256, dtype=x.dtype)
        state = (z, z)
    out, st = lstm_unroll(xin, state[0], state[1], p)
    return out, st


def policy_value(feat, p):
    pi = torch.softmax(feat @ p["W_base_fc_p"] + p["b_base_fc_p"], dim=1)
    v = (feat @ p["W_base_fc_v"] + p["b_base_fc_v"]).reshape(-1)
    return pi, v


def pc_head(feat, p):
    """model.py:411-443: FC -> [9,9,32] -> two VALID stride-2 4x4 deconvs -> dueling Q."""
    h = F.relu(feat @ p["W_pc_fc1"] + p["b_pc_fc1"]).reshape(-1, 9, 9, 32).permute(0, 3, 1, 2)
    v = F.relu(F.conv_transpose2d(h, p["W_pc_deconv_v"].permute(3, 2, 0, 1), p["b_pc_deconv_v"], stride=2))
    a = F.relu(F.conv_transpose2d(h, p["W_pc_deconv_a"].permute(3, 2, 0, 1), p["b_pc_deconv_a"], stride=2))
    q = v + a - a.mean(dim=1, keepdim=True)
    q = q.permute(0, 2, 3, 1)                          # [N,20,20,A]
    return q, q.max(dim=3)[0]


def rp_head(x3, p):
    """model.py:473-488: three frames -> conv -> flatten all three -> FC 3 -> softmax."""
    _, h2 = encoder(x3, p)
    flat = h2.reshape(1, 3 * 2592)
    return torch.softmax(flat @ p["W_rp_fc1"] + p["b_rp_fc1"], dim=1)


# ---- losses (model.py:490-598); every reduction is a SUM over the time axis ------------------------

def base_loss(pi, v, a_onehot, adv, R, entropy_beta):
    log_pi = torch.log(torch.clamp(pi, 1e-20, 1.0))
    entropy = -(pi * log_pi).sum(1)
    policy_loss = -(((log_pi * a_onehot).sum(1)) * adv + entropy * entropy_beta).sum()
    value_loss = 0.5 * 0.5 * ((R - v) ** 2).sum()
    return policy_loss, value_loss, entropy


def pc_loss(q, a_onehot, pc_R, lam):
    qa = (q * a_onehot.reshape(-1, 1, 1, a_onehot.shape[1])).sum(3)
    return lam * 0.5 * ((pc_R - qa) ** 2).sum()


def vr_loss(v, R):
    return 0.5 * ((R - v) ** 2).sum()


def rp_loss(c, target):
    return -(target * torch.log(torch.clamp(c, 1e-20, 1.0))).sum()


def unreal_loss(p, batch, cfg):
    """Total loss of ONE actor for one `process()` batch (model.py:579-598).

    batch keys: base_x [n,84,84,3], base_lar [n,A+1], base_a [n,A], base_adv [n], base_R [n],
    base_state (c,h) or None; pc_x, pc_lar, pc_a, pc_R [m,20,20]; vr_x, vr_lar, vr_R; rp_x [3,...], rp_c [1,3].
    """
    out = {}
    feat, _ = trunk(batch["base_x"], batch["base_lar"], p, cfg["use_lstm"], batch.get("base_state"))
    pi, v = policy_value(feat, p)
    pl, vl, ent = base_loss(pi, v, batch["base_a"], batch["base_adv"], batch["base_R"], cfg["entropy_beta"])
    out.update(policy_loss=pl, value_loss=vl, entropy=ent, base_loss=pl + vl, base_pi=pi, base_v=v)
    total = pl + vl
    if cfg.get("use_pixel_change", False):
        feat, _ = trunk(batch["pc_x"], batch["pc_lar"], p, cfg["use_lstm"], None)
        q, _ = pc_head(feat, p)
        out["pc_loss"] = pc_loss(q, batch["pc_a"], batch["pc_R"], cfg["pixel_change_lambda"])
        total = total + out["pc_loss"]
    if cfg.get("use_value_replay", False):
        feat, _ = trunk(batch["vr_x"], batch["vr_lar"], p, cfg["use_lstm"], None)
        _, vv = policy_value(feat, p)
        out["vr_loss"] = vr_loss(vv, batch["vr_R"])
        total = total + out["vr_loss"]
    if cfg.get("use_reward_prediction", False):
        c = rp_head(batch["rp_x"], p)
        out["rp_loss"] = rp_loss(c, batch["rp_c"])
        total = total + out["rp_loss"]
    out["total_loss"] = total
    return out
