"""BASELINE.json configs[1] at FULL size (4096 actors, history 2000, T = 20, full UNREAL) on one MI355X: properties that
do not depend on the size and need no CPU replay of 82 k env steps -- frames decode to legal maze states, stored pixel
change == the generic uint8 pixel-change kernel on the stored frames, metadata chains, n-step returns against a float64
restatement of trainer.py:313-324, sampled sequences / reward-prediction triples obey experience.py:100-153."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, H, T = 4096, 2000, 20


@pytest.fixture(scope="module")
def trained():
    import argparse
    from bench import build_trainer
    free, total = torch.cuda.mem_get_info(0)
    if free < 200 * 2 ** 30:
        pytest.skip("full-size replay ring needs ~195 GB of HBM, %.0f GB free" % (free / 2 ** 30))
    args = argparse.Namespace(actors=B, history=H)
    flags, net, tr = build_trainer(args, 0, 1, torch.device(DEV))
    while not tr._full:
        assert tr.process(None, 0) == (0, None)
    p0 = net.params.flat.clone()
    tr.cnt_at_fill = tr.ring.count.cpu().numpy().astype(np.int64)     # the environments are reset at this point (:203-205)
    steps = [tr.process(None, 0)[0] for _ in range(2)]
    torch.cuda.synchronize()
    return flags, net, tr, p0, steps


def test_steps_losses_and_update(trained):
    flags, net, tr, p0, steps = trained
    n = tr.n_steps.cpu().numpy()
    assert steps[-1] == int(n.sum()) and 0 < steps[-1] <= B * T and (n >= 1).all() and (n <= T).all()
    te = tr.terminal_end.cpu().numpy()
    assert ((n < T) <= (te == 1)).all()                     # an actor stops early only on a terminal
    l = tr.last_losses
    assert all(np.isfinite(l[k]) for k in ("total_loss", "policy_loss", "value_loss", "pc_loss", "vr_loss", "rp_loss"))
    g = float(tr.last_grad_norm.cpu()[0])
    assert np.isfinite(g) and g > 0
    assert not torch.equal(net.params.flat, p0) and bool(torch.isfinite(net.params.flat).all())


def test_ring_frames_are_legal_maze_states(trained):
    from oracle import maze as OM
    flags, net, tr, _, _ = trained
    ring, H1 = tr.ring, H + 1
    cnt = ring.count.cpu().numpy()
    assert (cnt >= H).all()
    rs = np.random.RandomState(0)
    bs = rs.randint(0, B, size=512)
    back = rs.randint(0, H, size=512)                        # how far behind the current observation
    slots = (cnt[bs] - back) % H1
    idx = torch.as_tensor(bs.astype(np.int64) * H1 + slots, device=DEV)
    fr = ring.frames.view(B * H1, 84, 84, 3)[idx].cpu().numpy()
    walls = OM.render(0, 2)[:, :, 0].astype(np.uint8)        # channel 0 is the constant wall image
    assert (fr[..., 0] == walls[None]).all() and (fr[..., 2] == 0).all()
    agent = fr[..., 1]
    assert set(np.unique(agent)) <= {0, 1} and (agent.reshape(512, -1).sum(1) == 144).all()
    blocks = agent.reshape(512, 7, 12, 7, 12).sum((2, 4))    # one 12x12 block of ones, on a free cell
    assert ((blocks == 144).sum((1, 2)) == 1).all()
    cy, cx = np.nonzero(blocks == 144)[1:]
    assert not any(OM.is_wall(int(x), int(y)) for x, y in zip(cx, cy))


def test_stored_pixel_change_and_metadata_chains(trained):
    from unreal_amd import ops
    flags, net, tr, _, _ = trained
    ring, H1 = tr.ring, H + 1
    cnt = ring.count.cpu().numpy().astype(np.int64)
    rs = np.random.RandomState(1)
    bs = rs.randint(0, B, size=4096).astype(np.int64)
    back = rs.randint(1, H - 1, size=4096)
    i_old = (cnt[bs] - back - 1) % H1                         # frame i (state before the action) and i + 1
    i_new = (cnt[bs] - back) % H1
    f_old, f_new = bs * H1 + i_old, bs * H1 + i_new
    term = ring.r_terminal.cpu().numpy()[f_old]
    rew = ring.r_reward.cpu().numpy()
    act = ring.r_action.cpu().numpy()
    assert set(np.unique(rew)) <= {-1.0, 0.0, 1.0}
    assert ((term == 1) == (rew[f_old] == 1.0)).all()         # +1 only at the goal, which is the only terminal
    # after a terminal, and once when the replay filled up (trainer.py:203-205), the next slot holds the reset state
    keep = (term == 0) & ((cnt[bs] - back) != tr.cnt_at_fill[bs])
    # stored analytic pixel change of frame i == generic |new - old| kernel on the stored uint8 frames
    out = torch.zeros(int(keep.sum()) * 400, device=DEV)
    ops.pixel_change_u8(ring.frames, torch.as_tensor(f_new[keep].astype(np.int32), device=DEV),
                        torch.as_tensor(f_old[keep].astype(np.int32), device=DEV), 48.0, out)
    want = ring.r_pc.view(B * H1, 400)[torch.as_tensor(f_old[keep], device=DEV)]
    diff = (out.view(-1, 400) - want).abs()
    assert float(diff.max()) <= 1e-7, (float(diff.max()), int((diff > 1e-7).any(1).sum()), int(keep.sum()))
    # last_action / last_reward of frame i+1 are action / reward of frame i (0 after a reset)
    la, lr = ring.r_last_action.cpu().numpy(), ring.r_last_reward.cpu().numpy()
    assert (la[f_new][keep] == act[f_old][keep]).all() and (lr[f_new][keep] == rew[f_old][keep]).all()
    assert (la[f_new][~keep] == 0).all() and (lr[f_new][~keep] == 0).all()


def test_nstep_returns_match_float64_restatement(trained):
    flags, net, tr, _, _ = trained
    r = tr.rewards.cpu().numpy().reshape(T, B).astype(np.float64)
    v = tr.v.cpu().numpy().reshape(T, B).astype(np.float64)
    n = tr.n_steps.cpu().numpy()
    te = tr.terminal_end.cpu().numpy()
    boot = tr.boot_v.cpu().numpy().astype(np.float64)
    R_dev = tr.R.cpu().numpy().reshape(T, B)
    adv_dev = tr.adv.cpu().numpy().reshape(T, B)
    R = np.where(te == 1, 0.0, boot)                          # trainer.py:298-300
    for t in reversed(range(T)):
        live = t < n
        R = np.where(live, r[t] + flags.gamma * R, R)         # :313-316 (clipped rewards are already in {-1,0,1})
        np.testing.assert_allclose(R_dev[t][live], R[live], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(adv_dev[t][live], (R - v[t])[live], rtol=2e-6, atol=4e-6)


def test_sampled_sequences_and_rp_triples(trained):
    flags, net, tr, _, _ = trained
    ring, H1, L = tr.ring, H + 1, T + 1
    cnt = ring.count.cpu().numpy().astype(np.int64)
    term = ring.r_terminal.cpu().numpy().reshape(B, H1)
    rew = ring.r_reward.cpu().numpy().reshape(B, H1)
    seq = tr.seq_idx.cpu().numpy().reshape(L, B).astype(np.int64)          # the value-replay sample of the last call
    ln = tr.seq_len.cpu().numpy()
    assert ((ln >= 1) & (ln <= L)).all()
    for b in range(0, B, 7):
        idx = seq[:ln[b], b]
        assert (idx // H1 == b).all()
        slots = idx % H1
        assert ((slots[1:] - slots[:-1]) % H1 == 1).all()                  # consecutive frames of this actor's ring
        top = cnt[b] - H                                                   # oldest live absolute index
        live = {(top + k) % H1 for k in range(H)}
        assert set(slots.tolist()) <= live                                 # never the slot of the current observation
        t_flags = term[b, slots]
        assert not t_flags[:-1].any()                                      # stops at the first terminal, inclusive
        assert ln[b] == L or t_flags[-1] == 1
    rp = tr.rp_ws.frame_idx[:3 * B].cpu().numpy().reshape(B, 3).astype(np.int64)
    cls = tr.rp_class.cpu().numpy()
    assert (rp // H1 == np.arange(B)[:, None]).all()
    s = rp % H1
    assert ((s[:, 1] - s[:, 0]) % H1 == 1).all() and ((s[:, 2] - s[:, 1]) % H1 == 1).all()
    r4 = rew[np.arange(B), (s[:, 2] + 1) % H1]                             # reward of the 4th frame (trainer.py:427-434)
    assert (cls == np.where(r4 == 0, 0, np.where(r4 > 0, 1, 2))).all()
