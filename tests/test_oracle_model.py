"""Cross-checks of the NN oracle (parity unpinned by the reference: no TensorFlow here).
  * variable-count spec of model/model_test.py:14-58 and the SURVEY parameter census
  * an independent numpy forward (explicit loops/einsum, no torch conv) vs oracle/model.py
  * fp64 finite differences vs autograd on the full UNREAL loss
  * the oracle trainer loop runs and learns nothing silly (smoke)
"""
import numpy as np
import torch

from oracle import model as M
from oracle.trainer import OracleTrainer

CFG = dict(action_size=4, use_lstm=True, use_pixel_change=True, use_value_replay=True,
           use_reward_prediction=True, pixel_change_lambda=0.05, entropy_beta=0.001,
           local_t_max=20, n_step_TD=20, gamma=0.99, gamma_pc=0.9, experience_history_size=40,
           max_time_step=10 ** 6, rmsp_alpha=0.99, rmsp_epsilon=0.1, grad_norm_clip=40.0,
           initial_alpha_low=1e-4, initial_alpha_high=5e-3, initial_alpha_log_rate=0.5)


def test_variable_counts():
    # model/model_test.py:14-58 uses action_size=1, LSTM on
    n = lambda **kw: len(M.param_spec(1, 0, True, **kw))
    assert n(use_pixel_change=True, use_value_replay=True, use_reward_prediction=True) == 20
    assert n(use_pixel_change=True, use_value_replay=False, use_reward_prediction=False) == 18
    assert n(use_pixel_change=False, use_value_replay=True, use_reward_prediction=False) == 12
    assert n(use_pixel_change=False, use_value_replay=False, use_reward_prediction=True) == 14
    tot = sum(int(np.prod(s)) for _, s, _ in M.param_spec(4))
    assert tot == 1898877
    ff = sum(int(np.prod(s)) for _, s, _ in M.param_spec(4, 0, False, False, False, False))
    assert ff == 676405


def _np_conv(x, W, b, stride):
    kh, kw, ci, co = W.shape
    N, H, Wd, _ = x.shape
    oh, ow = (H - kh) // stride + 1, (Wd - kw) // stride + 1
    out = np.zeros((N, oh, ow, co))
    for y in range(oh):
        for xx in range(ow):
            patch = x[:, y * stride:y * stride + kh, xx * stride:xx * stride + kw, :]
            out[:, y, xx, :] = np.einsum("nhwc,hwco->no", patch, W)
    return np.maximum(out + b, 0)


def _np_deconv(h, W, b, stride=2):
    kh, kw, co, ci = W.shape
    N, H, Wd, _ = h.shape
    out = np.zeros((N, (H - 1) * stride + kh, (Wd - 1) * stride + kw, co))
    for y in range(H):
        for x in range(Wd):
            out[:, y * stride:y * stride + kh, x * stride:x * stride + kw, :] += \
                np.einsum("nc,hwoc->nhwo", h[:, y, x, :], W)
    return np.maximum(out + b, 0)


def _sig(x):
    return 1 / (1 + np.exp(-x))


def test_numpy_forward_matches_oracle():
    rs = np.random.RandomState(3)
    p = M.init_params(4, seed=5, dtype=torch.float64)
    pn = {k: v.numpy() for k, v in p.items()}
    pn["lstm_bias"] = rs.uniform(-.1, .1, 1024)
    p["lstm_bias"] = torch.tensor(pn["lstm_bias"])
    T = 3
    x = rs.randint(0, 2, size=(T, 84, 84, 3)).astype(np.float64)
    lar = rs.uniform(-1, 1, size=(T, 5))
    c0, h0 = rs.uniform(-1, 1, 256), rs.uniform(-1, 1, 256)
    # numpy
    h1 = _np_conv(x, pn["W_base_conv1"], pn["b_base_conv1"], 4)
    h2 = _np_conv(h1, pn["W_base_conv2"], pn["b_base_conv2"], 2)
    assert h1.shape == (T, 20, 20, 16) and h2.shape == (T, 9, 9, 32)
    f = np.maximum(h2.reshape(T, 2592) @ pn["W_base_fc1"] + pn["b_base_fc1"], 0)
    c, h = c0, h0
    outs = []
    for t in range(T):
        g = np.concatenate([f[t], lar[t], h]) @ pn["lstm_kernel"] + pn["lstm_bias"]
        i, j, fg, o = np.split(g, 4)
        c = c * _sig(fg + 1.0) + _sig(i) * np.tanh(j)
        h = np.tanh(c) * _sig(o)
        outs.append(h)
    feat = np.stack(outs)
    logit = feat @ pn["W_base_fc_p"] + pn["b_base_fc_p"]
    pi = np.exp(logit - logit.max(1, keepdims=True))
    pi /= pi.sum(1, keepdims=True)
    v = (feat @ pn["W_base_fc_v"] + pn["b_base_fc_v"]).reshape(-1)
    hp = np.maximum(feat @ pn["W_pc_fc1"] + pn["b_pc_fc1"], 0).reshape(T, 9, 9, 32)
    dv = _np_deconv(hp, pn["W_pc_deconv_v"], pn["b_pc_deconv_v"])
    da = _np_deconv(hp, pn["W_pc_deconv_a"], pn["b_pc_deconv_a"])
    q = dv + da - da.mean(3, keepdims=True)
    rp_logit = h2.reshape(1, 7776) @ pn["W_rp_fc1"] + pn["b_rp_fc1"]
    rp = np.exp(rp_logit - rp_logit.max())
    rp /= rp.sum()
    # oracle
    tx, tl = torch.tensor(x), torch.tensor(lar)
    feat_o, (c_o, h_o) = M.trunk(tx, tl, p, True, (torch.tensor(c0), torch.tensor(h0)))
    pi_o, v_o = M.policy_value(feat_o, p)
    q_o, qmax_o = M.pc_head(feat_o, p)
    rp_o = M.rp_head(tx, p)
    np.testing.assert_allclose(feat_o.numpy(), feat, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(c_o.numpy(), c, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(pi_o.numpy(), pi, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(v_o.numpy(), v, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(q_o.numpy(), q, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(qmax_o.numpy(), q.max(3), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(rp_o.numpy(), rp, rtol=1e-10, atol=1e-12)
    assert q.shape == (T, 20, 20, 4)


def _rand_batch(rs, n=3):
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    a1h = np.eye(4)[rs.randint(0, 4, n)]
    b = dict(base_x=t(rs.randint(0, 2, (n, 84, 84, 3)).astype(float)), base_lar=t(rs.uniform(-1, 1, (n, 5))),
             base_a=t(a1h), base_adv=t(rs.normal(size=n)), base_R=t(rs.normal(size=n)),
             base_state=(t(rs.uniform(-1, 1, 256)), t(rs.uniform(-1, 1, 256))),
             pc_x=t(rs.randint(0, 2, (n, 84, 84, 3)).astype(float)), pc_lar=t(rs.uniform(-1, 1, (n, 5))),
             pc_a=t(np.eye(4)[rs.randint(0, 4, n)]), pc_R=t(rs.uniform(0, 1, (n, 20, 20))),
             vr_x=t(rs.randint(0, 2, (n, 84, 84, 3)).astype(float)), vr_lar=t(rs.uniform(-1, 1, (n, 5))),
             vr_R=t(rs.normal(size=n)), rp_x=t(rs.randint(0, 2, (3, 84, 84, 3)).astype(float)),
             rp_c=t([[0.0, 1.0, 0.0]]))
    return b


def test_finite_differences_fp64():
    rs = np.random.RandomState(11)
    p = M.init_params(4, seed=2, dtype=torch.float64)
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    batch = _rand_batch(rs)
    out = M.unreal_loss(p, batch, CFG)
    grads = dict(zip(p.keys(), torch.autograd.grad(out["total_loss"], list(p.values()))))
    eps = 1e-6
    for name in p:
        flat = p[name].detach().reshape(-1)
        for idx in rs.choice(flat.numel(), size=min(3, flat.numel()), replace=False):
            def loss_at(d):
                q = {k: v.detach().clone() for k, v in p.items()}
                q[name].reshape(-1)[idx] += d
                return float(M.unreal_loss(q, batch, CFG)["total_loss"])
            fd = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
            an = float(grads[name].reshape(-1)[idx])
            assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (name, idx, fd, an)


def test_trainer_smoke_async_equals_batched_for_one_actor():
    tr1 = OracleTrainer(CFG, n_actors=1, seed=1)
    tr2 = OracleTrainer(CFG, n_actors=1, seed=1)
    tr1.fill()
    tr2.fill()
    for k in range(2):
        d1, s1, l1 = tr1.process_async(0, 0)
        d2, infos, l2, _, norm = tr2.process_batched(0)
        assert d1 == d2 and 1 <= d1 <= 20
        assert abs(l1["total_loss"] - l2[0]["total_loss"]) < 1e-6
        assert abs(l1["grad_norm"] - norm) < 1e-4 * max(1, norm)
    for a, b in zip(tr1.params.values(), tr2.params.values()):
        np.testing.assert_array_equal(a.numpy(), b.numpy())
    assert tr1.anneal_lr(0) == tr1.initial_lr and abs(tr1.initial_lr - 7.0711e-4) < 1e-7
    assert tr1.anneal_lr(2 * 10 ** 6) == 0.0
