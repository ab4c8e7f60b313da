"""Kernel-level parity tests (GPU): every HIP kernel, called through the C ABI (ctypes), against the
CPU oracle / a plain PyTorch fp32-or-fp64 reference on the same seeded inputs.

Bars: integer / byte / index work bit-exact; floating point within the tolerance written at each
assert (fp32-grade kernels vs an fp64 reference of the same op: SURVEY 8d's forward bar, abs 1e-5 + rel 1e-5, unless
noted; sums over K >> 1 terms of O(1) magnitude carry an absolute term that grows with K).  Measured margins:
profiles/r04_parity_margins.md."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

try:
    import margins
except ImportError:            # imported as tests.<module> (__graft_entry__.smoke): tests/ itself is not on sys.path
    from tests import margins

pytestmark = pytest.mark.gpu

from oracle import maze as OM
from oracle import model as M
from oracle.experience import OracleExperience, Frame
from oracle.rmsprop import OracleRMSProp


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from unreal_amd import ops as _ops
    return _ops


DEV = "cuda:0"


def dev(a, dt=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.to(DEV).contiguous()


def close(got, ref, atol=1e-5, rtol=1e-5, what=""):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref) - (atol + rtol * np.abs(ref))
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    margins.record_close(what, got, ref, atol, rtol)
    assert err.max() <= 0, "%s: max |d|=%g at %s (ref %g)" % (
        what, np.abs(got - ref).max(), np.unravel_index(np.argmax(err), err.shape), ref.flat[np.argmax(err)])


# ---------------------------------------------------------------------------------------------------
# environment + ring
# ---------------------------------------------------------------------------------------------------
def _run_env(ops, B, H, steps, actions, reset_on_terminal=True):
    """Drive the device env and B oracle envs + oracle replays with the same actions."""
    ring = ops.Ring(B, H, DEV)
    ops.maze_reset(ring)
    envs = [OM.OracleMaze() for _ in range(B)]
    exps = [OracleExperience(H) for _ in range(B)]
    out_r = torch.zeros(B, device=DEV)
    out_t = torch.zeros(B, dtype=torch.int32, device=DEV)
    for s in range(steps):
        a = actions[s]
        ops.maze_step(ring, dev(a, torch.int32), None, out_r, out_t, reset_on_terminal, True)
        rr, tt = out_r.cpu().numpy(), out_t.cpu().numpy()
        for b in range(B):
            e = envs[b]
            prev, la, lr = e.last_state, e.last_action, e.last_reward
            _, r, t, pc = e.process(int(a[b]))
            exps[b].add_frame(Frame(prev, r, int(a[b]), t, pc, la, lr))
            assert rr[b] == r and bool(tt[b]) == bool(t)
            if t and reset_on_terminal:
                e.reset()
    return ring, envs, exps


def _check_ring(ring, envs, exps):
    B, H1 = ring.B, ring.H1
    frames = ring.frames.cpu().numpy().reshape(B, H1, 84, 84, 3)
    rr = ring.r_reward.cpu().numpy().reshape(B, H1)
    ra = ring.r_action.cpu().numpy().reshape(B, H1)
    rt = ring.r_terminal.cpu().numpy().reshape(B, H1)
    rla = ring.r_last_action.cpu().numpy().reshape(B, H1)
    rlr = ring.r_last_reward.cpu().numpy().reshape(B, H1)
    rpc = ring.r_pc.cpu().numpy().reshape(B, H1, 20, 20)
    cnt = ring.count.cpu().numpy()
    pos = ring.pos.cpu().numpy().reshape(B, 2)
    for b in range(B):
        e, x = envs[b], exps[b]
        assert cnt[b] == x.count
        assert (pos[b, 0], pos[b, 1]) == (e.x, e.y)
        # current observation is rendered in slot count % H1
        np.testing.assert_array_equal(frames[b, cnt[b] % H1], e.last_state['image'].astype(np.uint8))
        for i in range(x.top, x.count):
            f = x.frames[i]
            s = i % H1
            np.testing.assert_array_equal(frames[b, s], f.state['image'].astype(np.uint8))
            assert rr[b, s] == f.reward and ra[b, s] == f.action and bool(rt[b, s]) == bool(f.terminal)
            assert rla[b, s] == f.last_action and rlr[b, s] == f.last_reward
            np.testing.assert_array_equal(rpc[b, s], f.pixel_change.astype(np.float32))   # bit-exact fp32
    assert int(ring.last_action.cpu()[0]) == envs[0].last_action
    assert float(ring.last_reward.cpu()[0]) == envs[0].last_reward


def test_maze_step_random_walk_bit_exact(ops):
    rs = np.random.RandomState(5)
    B, H, steps = 6, 16, 120
    actions = rs.randint(0, 4, size=(steps, B))
    ring, envs, exps = _run_env(ops, B, H, steps, actions)
    _check_ring(ring, envs, exps)


def test_maze_golden_trace_and_episode(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "maze_trace.npz"))
    n = 1500
    B = 2
    acts = np.stack([g["actions"][:n], g["actions"][:n][::-1]], 1)
    ring = ops.Ring(B, 32, DEV)
    ops.maze_reset(ring)
    out_r = torch.zeros(B, device=DEV)
    out_t = torch.zeros(B, dtype=torch.int32, device=DEV)
    for i in range(n):
        ops.maze_step(ring, dev(acts[i], torch.int32), None, out_r, out_t, True, False)
        if i % 50 == 0 or g["terminal"][i]:
            assert out_r.cpu()[0] == g["reward"][i] and bool(out_t.cpu()[0]) == bool(g["terminal"][i])
            p = ring.pos.cpu().numpy()
            exp_pos = tuple(g["pos"][i]) if not g["terminal"][i] else OM.START
            assert (p[0], p[1]) == exp_pos
    # scripted shortest path: 20 moves, return +1 (BASELINE.md derived fact)
    from collections import deque
    prev = {OM.START: None}
    dq = deque([OM.START])
    while dq:
        c = dq.popleft()
        for a in range(4):
            nx, ny, _ = OM.move(c[0], c[1], a)
            if (nx, ny) not in prev:
                prev[(nx, ny)] = (c, a)
                dq.append((nx, ny))
    path = []
    c = OM.GOAL
    while prev[c] is not None:
        c, a = prev[c]
        path.append(a)
    path.reverse()
    assert len(path) == 20
    ring = ops.Ring(1, 32, DEV)
    ops.maze_reset(ring)
    tot = 0.0
    for a in path:
        ops.maze_step(ring, dev([a], torch.int32), None, out_r[:1], out_t[:1], True, True)
        tot += float(out_r.cpu()[0])
    assert tot == 1.0 and int(out_t.cpu()[0]) == 1
    assert float(ring.score_out.cpu()[0]) == 1.0 and int(ring.score_valid.cpu()[0]) == 1


def test_maze_table_all_cells(ops, golden_dir):
    """Every (free cell, action): next cell, reward, terminal, frame, pixel-change vs the reference table."""
    g = np.load(os.path.join(golden_dir, "maze_table.npz"))
    cells = g["cells"]
    B = len(cells) * 4
    ring = ops.Ring(B, 8, DEV)
    ops.maze_reset(ring)
    pos = np.repeat(cells, 4, axis=0).astype(np.int32)
    ring.pos.copy_(dev(pos.reshape(-1)))
    acts = np.tile(np.arange(4), len(cells)).astype(np.int32)
    out_r = torch.zeros(B, device=DEV)
    out_t = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.maze_step(ring, dev(acts), None, out_r, out_t, False, False)
    np.testing.assert_array_equal(ring.pos.cpu().numpy().reshape(-1, 4, 2), g["next"])
    np.testing.assert_array_equal(out_r.cpu().numpy().reshape(-1, 4), g["reward"].astype(np.float32))
    np.testing.assert_array_equal(out_t.cpu().numpy().reshape(-1, 4), g["terminal"])
    pc = ring.r_pc.cpu().numpy().reshape(B, 9, 20, 20)[:, 0].reshape(-1, 4, 20, 20)
    np.testing.assert_array_equal(pc, g["pixel_change"].astype(np.float32))
    fr = ring.frames.cpu().numpy().reshape(B, 9, 84, 84, 3)[:, 1]
    cell_of = {tuple(c): i for i, c in enumerate(cells)}
    for k in range(B):
        np.testing.assert_array_equal(fr[k], g["frames"][cell_of[tuple(g["next"][k // 4, k % 4])]])
    # generic uint8 pixel-change kernel agrees with the analytic maze form
    before = np.stack([g["frames"][k // 4] for k in range(B)])
    pool = torch.cat([dev(before.reshape(-1)), dev(fr.reshape(-1))])
    idx_old = torch.arange(B, dtype=torch.int32, device=DEV)
    idx_new = idx_old + B
    out = torch.zeros(B * 400, device=DEV)
    ops.pixel_change_u8(pool, idx_new, idx_old, 48.0, out)
    np.testing.assert_array_equal(out.cpu().numpy().reshape(-1, 4, 20, 20), g["pixel_change"].astype(np.float32))


def test_replay_sampling_matches_oracle(ops):
    rs = np.random.RandomState(11)
    B, H, L = 8, 40, 11
    steps = 150
    actions = rs.randint(0, 4, size=(steps, B))
    ring, envs, exps = _run_env(ops, B, H, steps, actions)
    # inject terminals / rewards patterns that the random walk will not produce often
    rt = ring.r_terminal.cpu().numpy().reshape(B, H + 1)
    rr = ring.r_reward.cpu().numpy().reshape(B, H + 1)
    for b in range(B):
        x = exps[b]
        for i in range(x.top, x.count):
            if rs.rand() < 0.12:
                x.frames[i].terminal = True
                rt[b, i % (H + 1)] = 1
            if rs.rand() < 0.2:
                x.frames[i].reward = 1
                rr[b, i % (H + 1)] = 1.0
        # no successive terminals (the reference assumes so)
        for i in range(x.top + 1, x.count):
            if x.frames[i].terminal and x.frames[i - 1].terminal:
                x.frames[i].terminal = False
                rt[b, i % (H + 1)] = 0
    ring.r_terminal.copy_(dev(rt.reshape(-1)))
    ring.r_reward.copy_(dev(rr.reshape(-1)))
    for trial in range(20):
        start = rs.randint(0, H - L - 1, size=B).astype(np.int32)
        seq_idx = torch.zeros(L * B, dtype=torch.int32, device=DEV)
        seq_len = torch.zeros(B, dtype=torch.int32, device=DEV)
        ops.replay_sample_seq(ring, L, dev(start), seq_idx, seq_len)
        si = seq_idx.cpu().numpy().reshape(L, B)
        sl = seq_len.cpu().numpy()
        coin = rs.randint(0, 2, size=B).astype(np.int32)
        u = rs.random_sample(B)
        rp_idx = torch.zeros(3 * B, dtype=torch.int32, device=DEV)
        rp_cls = torch.zeros(B, dtype=torch.int32, device=DEV)
        ops.replay_sample_rp(ring, dev(coin), dev(u), rp_idx, rp_cls)
        ri = rp_idx.cpu().numpy().reshape(B, 3)
        rc = rp_cls.cpu().numpy()
        for b in range(B):
            x = exps[b]
            want = x.sequence_from_start(int(start[b]), L)
            assert sl[b] == len(want)
            assert [b * (H + 1) + i % (H + 1) for i in want] == list(si[:len(want), b])
            rp = x.rp_from_draws(int(coin[b]), lambda n: min(n - 1, int(u[b] * n)))
            assert [b * (H + 1) + i % (H + 1) for i in rp[:3]] == list(ri[b])
            r = x.frames[rp[3]].reward
            assert rc[b] == (0 if r == 0 else (1 if r > 0 else 2))


def test_lab_replay_matches_reference_fixture(ops, golden_dir):
    """Device ring + unreal_replay_sample_seq / unreal_replay_sample_rp(mode=1) vs the trace tests/golden/make_fixtures.py
    recorded from the reference's upstream replay (train/experience_lab_ver.py:14,18,63-93,100-151): the scripted
    reward / terminal stream is fed through unreal_hostfed_step (one actor, H = 64, reward clipping on), then at every
    sample point the reference's own draws (start position; coin and pick, logged around np.random.randint) select
    frames on the device.  Bit-exact: clipped stored rewards, window, sampled slots, reward class."""
    g = np.load(os.path.join(golden_dir, "replay_lab_ver.npz"))
    H = int(g["H"][0]); H1 = H + 1
    raw, terms = g["raw_reward"], g["terminals"]
    ring = ops.Ring(1, H, DEV)
    rs = np.random.RandomState(5)
    staged = dev(rs.randint(0, 256, size=ops.FRAME_BYTES).astype(np.uint8))
    ops.hostfed_reset(ring, staged)
    ring.last_action.fill_(5)                               # the fixture's frame 0 has last_action (0 + 5) % 6
    sid = list(g["stored_id"])
    abs_of = {i: j for j, i in enumerate(sid)}              # caller's frame id -> absolute ring index
    k = 0
    for i in range(len(raw)):
        ops.hostfed_step(ring, staged, dev([i % 6], torch.int32), dev([raw[i]], torch.float32),
                         dev([int(terms[i])], torch.int32), reset_on_terminal=False, clip_reward=True)
        cnt = int(ring.count.cpu()[0])
        assert cnt == sid.index(i) + 1 if i in abs_of else cnt == len([x for x in sid if x < i])
        assert max(0, cnt - H) == g["top"][i] and min(cnt, H) == g["length"][i]
        if k < len(g["sample_at"]) and i == g["sample_at"][k]:
            L = 21
            seq_idx = torch.zeros(L, dtype=torch.int32, device=DEV)
            seq_len = torch.zeros(1, dtype=torch.int32, device=DEV)
            ops.replay_sample_seq(ring, L, dev([g["seq_start"][k]], torch.int32), seq_idx, seq_len)
            want = [x for x in g["seq_ids"][k] if x >= 0]
            assert int(seq_len.cpu()[0]) == len(want)
            assert list(seq_idx.cpu().numpy()[:len(want)]) == [abs_of[x] % H1 for x in want]
            n, pick = int(g["rp_n"][k]), int(g["rp_pick"][k])
            rp_idx = torch.zeros(3, dtype=torch.int32, device=DEV)
            rp_cls = torch.zeros(1, dtype=torch.int32, device=DEV)
            ops.replay_sample_rp(ring, dev([g["rp_coin"][k]], torch.int32), dev([(pick + 0.5) / n], torch.float64), rp_idx,
                                 rp_cls, mode=1)
            ids = list(g["rp_ids"][k])
            assert list(rp_idx.cpu().numpy()) == [abs_of[x] % H1 for x in ids[:3]]
            r4 = float(np.clip(raw[ids[3]], -1, 1))
            assert int(rp_cls.cpu()[0]) == (0 if r4 == 0 else (1 if r4 > 0 else 2))
            k += 1
    assert k == len(g["sample_at"])
    rr = ring.r_reward.cpu().numpy(); rlr = ring.r_last_reward.cpu().numpy(); rla = ring.r_last_action.cpu().numpy()
    cnt = int(ring.count.cpu()[0])
    assert cnt == len(sid)
    for j in range(cnt - H, cnt):                           # the live window: rewards as the reference stored them
        assert rr[j % H1] == g["stored_reward"][j] and rlr[j % H1] == g["stored_last_reward"][j]
        assert rla[j % H1] == (sid[j] + 5) % 6


def test_pixel_change_u8_matches_reference_fixture(ops, golden_dir):
    """unreal_hostfed_step's pixel change and unreal_pixel_change_u8 vs Environment._calc_pixel_change
    (environment.py:88-99) evaluated by the reference on 96 uint8/255 float32 frame pairs.  The device sums
    |a-b| over the 4x4x3 bytes exactly and divides once (fp32): <= 3e-7 from the reference's float32 chain of means."""
    from tests.test_oracle_golden import lab_u8_pairs
    g = np.load(os.path.join(golden_dir, "pixel_change_u8.npz"))
    n = int(g["n"][0])
    u8 = lab_u8_pairs(int(g["seed"][0]), n)
    pool = dev(u8.reshape(-1))
    out = torch.zeros(n * 400, device=DEV)
    idx_old = dev(np.arange(n) * 2, torch.int32); idx_new = dev(np.arange(n) * 2 + 1, torch.int32)
    ops.pixel_change_u8(pool, idx_new, idx_old, 48.0 * 255.0, out)
    close(out.view(n, 20, 20), g["pixel_change"], atol=3e-7, rtol=0, what="pixel_change_u8")
    # the same through the host-fed step of n actors: old frame = current observation, new frame = staged
    ring = ops.Ring(n, 4, DEV)
    ops.hostfed_reset(ring, dev(u8[:, 0].reshape(-1)))
    z = torch.zeros(n, dtype=torch.int32, device=DEV)
    ops.hostfed_step(ring, dev(u8[:, 1].reshape(-1)), z, torch.zeros(n, device=DEV), z)
    pc = ring.r_pc.view(n, 5, 400)[:, 0].reshape(n, 20, 20)
    close(pc, g["pixel_change"], atol=3e-7, rtol=0, what="hostfed_step pixel change")
    assert float(pc[1].abs().max()) == 0.0


def test_return_scans(ops):
    rs = np.random.RandomState(3)
    B, T = 37, 20
    rewards = rs.choice([-1.0, 0.0, 1.0], size=(T, B)).astype(np.float32)
    values = rs.normal(size=(T, B)).astype(np.float32)
    n_steps = rs.randint(1, T + 1, size=B).astype(np.int32)
    n_steps[:5] = T
    term_end = (n_steps < T).astype(np.int32)
    boot = rs.normal(size=B).astype(np.float32)
    R_out = torch.zeros(T * B, device=DEV)
    adv_out = torch.zeros(T * B, device=DEV)
    ops.base_returns(B, T, dev(rewards.reshape(-1)), dev(values.reshape(-1)), dev(n_steps), dev(boot),
                     dev(term_end), 0.99, R_out, adv_out)
    Rr = np.zeros((T, B), np.float32)
    Ar = np.zeros((T, B), np.float32)
    for b in range(B):
        R = 0.0 if term_end[b] else float(boot[b])
        for t in reversed(range(n_steps[b])):
            R = float(rewards[t, b]) + 0.99 * R
            Rr[t, b] = R
            Ar[t, b] = R - float(values[t, b])
    np.testing.assert_array_equal(R_out.cpu().numpy().reshape(T, B), Rr)
    np.testing.assert_array_equal(adv_out.cpu().numpy().reshape(T, B), Ar)


# ---------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ta,tb,M,N,K", [
    (0, 0, 64, 256, 2592), (0, 0, 200, 1024, 261), (0, 0, 4096, 256, 512), (0, 0, 7, 5, 3),
    (0, 1, 130, 2592, 256), (0, 1, 64, 256, 1024), (0, 1, 33, 261, 1024),
    (1, 0, 2592, 256, 300), (1, 0, 261, 1024, 77), (1, 0, 256, 1024, 1000), (1, 1, 100, 70, 50),
])
def test_gemm_variants(ops, ta, tb, M, N, K):
    rs = np.random.RandomState(M + N + K)
    lda = ((M if ta else K) + 7) // 4 * 4
    ldb = ((K if tb else N) + 7) // 4 * 4
    ldc = N + 3
    A = rs.uniform(-1, 1, size=((K if ta else M), lda))
    B = rs.uniform(-1, 1, size=((N if tb else K), ldb))
    bias = rs.uniform(-1, 1, size=N)
    Aop = (A[:, :M].T if ta else A[:, :K])
    Bop = (B[:, :K].T if tb else B[:, :N])
    ref = Aop @ Bop
    tol = dict(atol=1e-7 * K, rtol=1e-5)      # (rounds 1-3: 3e-7 * K; measured worst 0.12 of that)
    C = torch.full((M, ldc), 7.0, device=DEV)
    ops.gemm(ta, tb, M, N, K, dev(A, torch.float32), lda, dev(B, torch.float32), ldb, C, ldc, bias=dev(bias, torch.float32))
    close(C[:, :N], ref + bias, what="bias", **tol)
    assert float(C[:, N:].min()) == 7.0                      # padding columns untouched
    # relu + accumulate
    C0 = rs.uniform(-1, 1, size=(M, ldc))
    C = dev(C0, torch.float32)
    ops.gemm(ta, tb, M, N, K, dev(A, torch.float32), lda, dev(B, torch.float32), ldb, C, ldc,
             flags=ops.GEMM_ACCUM | ops.GEMM_RELU)
    close(C[:, :N], np.maximum(ref + C0[:, :N], 0), what="accum+relu", **tol)
    # relu mask
    msk = rs.uniform(-1, 1, size=(M, N))
    C = torch.zeros(M, ldc, device=DEV)
    ops.gemm(ta, tb, M, N, K, dev(A, torch.float32), lda, dev(B, torch.float32), ldb, C, ldc,
             mask=dev(msk, torch.float32), ldm=N, flags=ops.GEMM_RELU_MASK)
    close(C[:, :N], ref * (msk > 0), what="mask", **tol)
    # split-K atomics
    C = torch.zeros(M, ldc, device=DEV)
    ops.gemm(ta, tb, M, N, K, dev(A, torch.float32), lda, dev(B, torch.float32), ldb, C, ldc,
             bias=dev(bias, torch.float32), flags=ops.GEMM_ATOMIC, splitk=4)
    close(C[:, :N], ref + bias, what="splitk", **tol)


def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    n = 96
    Bm = np.arange(n * n, dtype=np.float32).reshape(n, n) % 251
    C = torch.zeros(n, n, device=DEV)
    ops.gemm(0, 0, n, n, n, dev(np.eye(n, dtype=np.float32)), n, dev(Bm), n, C, n)
    np.testing.assert_array_equal(C.cpu().numpy(), Bm)


# ---------------------------------------------------------------------------------------------------
# encoder
# ---------------------------------------------------------------------------------------------------
def _enc_params(seed):
    p = M.init_params(4, seed=seed, dtype=torch.float64)
    rs = np.random.RandomState(seed)
    p["b_base_conv1"] = torch.tensor(rs.uniform(-.05, .05, 16))
    return p


@pytest.mark.parametrize("N,mode", [(1, "maze"), (5, "u8"), (67, "u8"), (600, "maze"), (1300, "u8")])   # > 512 workgroups: grid stride
def test_encoder_fwd(ops, N, mode):
    rs = np.random.RandomState(N)
    p = _enc_params(1)
    pool_n = N + 3
    if mode == "maze":
        pool = rs.randint(0, 2, size=(pool_n, 84, 84, 3)).astype(np.uint8)
        scale = 1.0
    else:
        pool = rs.randint(0, 256, size=(pool_n, 84, 84, 3)).astype(np.uint8)
        scale = 1.0 / 255.0
    idx = rs.randint(0, pool_n, size=N).astype(np.int32)
    x = torch.tensor(pool[idx].astype(np.float64)) * scale
    h1, h2 = M.encoder(x, p)
    f2 = torch.zeros(N * 2592, device=DEV)
    c1 = torch.zeros(N * 6400, device=DEV)
    P = {k: dev(v, torch.float32) for k, v in p.items()}
    ops.encoder_fwd(dev(pool.reshape(-1)), dev(idx), scale, P["W_base_conv1"], P["b_base_conv1"],
                    P["W_base_conv2"], P["b_base_conv2"], f2, c1)
    close(c1.reshape(N, 20, 20, 16), h1, what="conv1")
    close(f2.reshape(N, 9, 9, 32), h2, what="conv2")
    f2b = torch.zeros(N * 2592, device=DEV)
    bits = torch.full((N * 162,), -1, dtype=torch.int16, device=DEV)
    ops.encoder_fwd(dev(pool.reshape(-1)), dev(idx), scale, P["W_base_conv1"], P["b_base_conv1"],
                    P["W_base_conv2"], P["b_base_conv2"], f2b, None, relu_bits=bits)
    assert torch.equal(f2, f2b)
    # the 1-bit ReLU pattern: bit j % 16 of word j / 16 of a frame <=> f2[j] > 0 (what the fc dgrad masks with)
    w = bits.cpu().numpy().view(np.uint16).reshape(N, 162).astype(np.uint32)
    got = ((w[:, :, None] >> np.arange(16, dtype=np.uint32)[None, None, :]) & 1).reshape(N, 2592).astype(bool)
    np.testing.assert_array_equal(got, f2.cpu().numpy().reshape(N, 2592) > 0)
    # and the dgrad epilogue masks identically with the bits and with the fp32 activations
    rs2 = np.random.RandomState(N + 7)
    Wd = dev(rs2.uniform(-1, 1, size=2592 * 256), torch.float32)
    sh = ops.SplitWeights(Wd, 2592, 256, 256, False)
    d_fc = dev(rs2.uniform(-1, 1, size=N * 256), torch.float32)
    a = torch.zeros(N * 2592, device=DEV); b = torch.zeros(N * 2592, device=DEV)
    ops.gemm_split_nt(N, 2592, 256, d_fc, 256, sh, a, 2592, mask=f2, ldm=2592, flags=ops.GEMM_RELU_MASK)
    ops.gemm_split_nt(N, 2592, 256, d_fc, 256, sh, b, 2592, mask=bits, ldm=162, flags=ops.GEMM_RELU_BITS)
    assert torch.equal(a, b) and float(a.abs().max()) > 0


@pytest.mark.parametrize("N", [1, 6, 131, 1100])     # 1100 > 512 workgroups: the grid-stride / prefetch path
def test_encoder_bwd(ops, N):
    rs = np.random.RandomState(N + 100)
    p = _enc_params(2)
    pool = rs.randint(0, 256, size=(N, 84, 84, 3)).astype(np.uint8)
    scale = 1.0 / 255.0
    idx = rs.permutation(N).astype(np.int32)
    W1 = p["W_base_conv1"].clone().requires_grad_(True)
    b1 = p["b_base_conv1"].clone().requires_grad_(True)
    W2 = p["W_base_conv2"].clone().requires_grad_(True)
    b2 = p["b_base_conv2"].clone().requires_grad_(True)
    q = dict(W_base_conv1=W1, b_base_conv1=b1, W_base_conv2=W2, b_base_conv2=b2)
    x = torch.tensor(pool[idx].astype(np.float64)) * scale
    h1, h2 = M.encoder(x, q)
    g_out = torch.tensor(rs.normal(size=h2.shape))
    (h2 * g_out).sum().backward()
    d2 = (g_out * (h2 > 0)).detach()          # gradient wrt conv2 pre-activation
    P = {k: dev(v, torch.float32) for k, v in p.items()}
    dW1 = torch.zeros(3072, device=DEV); db1 = torch.zeros(16, device=DEV)
    dW2 = torch.zeros(8192, device=DEV); db2 = torch.zeros(32, device=DEV)
    ops.encoder_bwd(dev(pool.reshape(-1)), dev(idx), scale, P["W_base_conv2"],
                    dev(h1.detach().reshape(-1), torch.float32), dev(d2.reshape(-1), torch.float32),
                    dW1, db1, dW2, db2)
    tol = dict(atol=3e-5 * max(1, N ** 0.5), rtol=1e-4)
    close(dW2.reshape(4, 4, 16, 32), W2.grad, what="dW2", **tol)
    close(db2, b2.grad, what="db2", **tol)
    close(db1, b1.grad, what="db1", **tol)
    close(dW1.reshape(8, 8, 3, 16), W1.grad, what="dW1", **tol)


# ---------------------------------------------------------------------------------------------------
# LSTM gates, heads, losses
# ---------------------------------------------------------------------------------------------------
def test_lstm_gates_fwd_bwd(ops):
    rs = np.random.RandomState(9)
    rows = 70
    pre = torch.tensor(rs.normal(size=(rows, 1024)), requires_grad=True)
    bias = torch.tensor(rs.normal(size=1024) * 0.1, requires_grad=True)
    c0 = torch.tensor(rs.normal(size=(rows, 256)), requires_grad=True)
    g = pre + bias
    i, j, f, o = torch.sigmoid(g[:, :256]), torch.tanh(g[:, 256:512]), torch.sigmoid(g[:, 512:768] + 1.0), \
        torch.sigmoid(g[:, 768:])
    c = c0 * f + i * j
    h = torch.tanh(c) * o
    dh = torch.tensor(rs.normal(size=(rows, 256)))
    dhr = torch.tensor(rs.normal(size=(rows, 256)))
    dc = torch.tensor(rs.normal(size=(rows, 256)))
    ((h * (dh + dhr)).sum() + (c * dc).sum()).backward()
    ga = torch.zeros(rows * 1024, device=DEV); co = torch.zeros(rows * 256, device=DEV)
    ho = torch.zeros(rows * 264, device=DEV)
    ops.lstm_gates_fwd(rows, dev(pre.detach().reshape(-1), torch.float32), dev(bias.detach(), torch.float32),
                       dev(c0.detach().reshape(-1), torch.float32), ga, co, ho, 264)
    close(co.reshape(rows, 256), c, what="c")
    close(ho.reshape(rows, 264)[:, :256], h, what="h")
    close(ga.reshape(rows, 1024), torch.cat([i, j, f, o], 1), what="gates")
    dpre = torch.zeros(rows * 1024, device=DEV)
    dcio = dev(dc.reshape(-1), torch.float32)
    ops.lstm_gates_bwd(rows, dev(dh.reshape(-1), torch.float32), dev(dhr.reshape(-1), torch.float32), dcio, ga,
                       dev(c0.detach().reshape(-1), torch.float32), co, dpre)
    close(dpre.reshape(rows, 1024), pre.grad, what="dpre")
    close(dcio.reshape(rows, 256), c0.grad, what="dc_prev")


@pytest.mark.parametrize("rows,K,NOUT", [(37, 256, 4), (37, 256, 1), (300, 7776, 3), (5, 256, 6)])
def test_linear_small(ops, rows, K, NOUT):
    rs = np.random.RandomState(rows + K)
    ldx = K + 8
    X = torch.tensor(rs.normal(size=(rows, ldx)), requires_grad=True)
    W = torch.tensor(rs.normal(size=(K, NOUT)) / np.sqrt(K), requires_grad=True)
    b = torch.tensor(rs.normal(size=NOUT), requires_grad=True)
    out = X[:, :K] @ W + b
    dO = torch.tensor(rs.normal(size=(rows, NOUT)))
    (out * dO).sum().backward()
    o = torch.zeros(rows * NOUT, device=DEV)
    ops.linear_small_fwd(rows, K, NOUT, dev(X.detach().reshape(-1), torch.float32), ldx,
                         dev(W.detach().reshape(-1), torch.float32), dev(b.detach(), torch.float32), o, NOUT)
    close(o.reshape(rows, NOUT), out, what="fwd")
    dX0 = rs.normal(size=(rows, K))
    dX = dev(dX0.reshape(-1), torch.float32)
    dW = torch.zeros(K * NOUT, device=DEV); db = torch.zeros(NOUT, device=DEV)
    ops.linear_small_bwd(rows, K, NOUT, dev(X.detach().reshape(-1), torch.float32), ldx,
                         dev(dO.reshape(-1), torch.float32), NOUT, dev(W.detach().reshape(-1), torch.float32), dX, K,
                         True, dW, db)
    close(dX.reshape(rows, K), X.grad[:, :K] + dX0, what="dX")
    close(dW.reshape(K, NOUT), W.grad, what="dW", atol=2e-4, rtol=1e-4)
    close(db, b.grad, what="db", atol=2e-4, rtol=1e-4)


def test_softmax_sample_matches_numpy_choice(ops):
    rs = np.random.RandomState(4)
    rows, A = 4096, 4
    logits = (rs.normal(size=(rows, A)) * 3).astype(np.float32)
    u = rs.random_sample(rows)
    u[:8] = [0.0, 1.0 - 2 ** -53, 0.25, 0.5, 0.75, 0.1, 0.9, 0.999999]
    lp = dev(logits.reshape(-1))
    act = torch.zeros(rows, dtype=torch.int32, device=DEV)
    ops.softmax_sample(rows, A, lp, A, dev(u), act)
    pi = lp.cpu().numpy().reshape(rows, A)
    ref = torch.softmax(torch.tensor(logits.astype(np.float64)), 1).numpy()
    close(pi, ref, atol=1e-6, rtol=1e-5, what="pi")
    # inverse-CDF draw exactly as numpy RandomState.choice does it (trainer.py:147-148), on the device's own pi
    cdf = np.cumsum(pi.astype(np.float64), 1)
    cdf /= cdf[:, -1:]
    want = np.array([np.searchsorted(cdf[r], u[r], side='right') for r in range(rows)])
    np.testing.assert_array_equal(act.cpu().numpy(), np.minimum(want, A - 1))


def test_base_vr_rp_loss_grads(ops):
    rs = np.random.RandomState(8)
    rows, A = 333, 4
    beta, gs = 0.001, 1.0 / 7
    logits = torch.tensor(rs.normal(size=(rows, A)) * 2, requires_grad=True)
    v = torch.tensor(rs.normal(size=rows), requires_grad=True)
    act = rs.randint(0, A, rows)
    adv = torch.tensor(rs.normal(size=rows)); R = torch.tensor(rs.normal(size=rows))
    active = (rs.rand(rows) < 0.8).astype(np.int32)
    m = torch.tensor(active.astype(np.float64))
    pi = torch.softmax(logits, 1)
    a1h = torch.tensor(np.eye(A)[act])
    log_pi = torch.log(torch.clamp(pi, 1e-20, 1.0))
    ent = -(pi * log_pi).sum(1)
    pl = -(((log_pi * a1h).sum(1) * adv + ent * beta) * m).sum()
    vl = 0.25 * (((R - v) ** 2) * m).sum()
    ((pl + vl) * gs).backward()
    dl = torch.zeros(rows * A, device=DEV); dv = torch.zeros(rows, device=DEV); losses = torch.zeros(3, device=DEV)
    ops.base_loss_grad(rows, A, dev(pi.detach().reshape(-1), torch.float32), A, dev(v.detach(), torch.float32),
                       dev(act, torch.int32), dev(adv, torch.float32), dev(R, torch.float32), dev(active), beta, gs,
                       dl, dv, losses)
    close(dl.reshape(rows, A), logits.grad, what="dlogits", atol=1e-6, rtol=1e-4)
    close(dv, v.grad, what="dv", atol=1e-6, rtol=1e-4)
    close(losses, [float(pl) * gs, float(vl) * gs, float((ent * m).sum()) * gs], atol=1e-4, rtol=1e-4, what="losses")
    # value replay
    v2 = torch.tensor(rs.normal(size=rows), requires_grad=True)
    l2 = 0.5 * (((R - v2) ** 2) * m).sum()
    (l2 * gs).backward()
    dv2 = torch.zeros(rows, device=DEV); ls = torch.zeros(1, device=DEV)
    ops.vr_loss_grad(rows, dev(v2.detach(), torch.float32), dev(R, torch.float32), dev(active), gs, dv2, ls)
    close(dv2, v2.grad, what="vr dv", atol=1e-6, rtol=1e-4)
    close(ls, [float(l2) * gs], atol=1e-4, rtol=1e-4, what="vr loss")
    # reward prediction
    z = torch.tensor(rs.normal(size=(rows, 3)) * 2, requires_grad=True)
    cls = rs.randint(0, 3, rows)
    pr = torch.softmax(z, 1)
    l3 = -(torch.tensor(np.eye(3)[cls]) * torch.log(torch.clamp(pr, 1e-20, 1.0))).sum()
    (l3 * gs).backward()
    prob = torch.zeros(rows * 3, device=DEV); dz = torch.zeros(rows * 3, device=DEV); ls = torch.zeros(1, device=DEV)
    ops.rp_loss_grad(rows, dev(z.detach().reshape(-1), torch.float32), dev(cls, torch.int32), gs, prob, dz, ls)
    close(prob.reshape(rows, 3), pr, atol=1e-6, rtol=1e-5, what="rp prob")
    close(dz.reshape(rows, 3), z.grad, atol=1e-6, rtol=1e-4, what="rp dlogits")
    close(ls, [float(l3) * gs], atol=1e-4, rtol=1e-4, what="rp loss")


def test_colsum_relu_mask_lar(ops):
    rs = np.random.RandomState(2)
    rows, cols, ld = 777, 300, 304
    X = rs.normal(size=(rows, ld)).astype(np.float32)
    out = torch.zeros(cols, device=DEV)
    ops.colsum(rows, cols, dev(X.reshape(-1)), ld, out)
    close(out, X[:, :cols].astype(np.float64).sum(0), atol=1e-3, rtol=1e-5, what="colsum")
    S = rs.normal(size=(rows, cols)).astype(np.float32)
    D = dev(X.reshape(-1).copy())
    ops.relu_mask(rows, cols, D, ld, dev(S.reshape(-1)), cols)
    want = X.copy()
    want[:, :cols] *= (S > 0)
    np.testing.assert_array_equal(D.cpu().numpy().reshape(rows, ld), want)
    la = rs.randint(0, 4, 50).astype(np.int32); lr = rs.choice([-1.0, 0.0, 1.0], 50).astype(np.float32)
    idx = rs.randint(0, 50, 20).astype(np.int32)
    xc = torch.full((20 * 264,), 9.0, device=DEV)
    ops.lar_fill(20, 4, dev(la), dev(lr), dev(idx), xc, 264)
    got = xc.cpu().numpy().reshape(20, 264)
    assert (got[:, :256] == 9.0).all() and (got[:, 261:] == 0).all()
    for r in range(20):
        from oracle.experience import concat_action_and_reward
        np.testing.assert_array_equal(got[r, 256:261], concat_action_and_reward(la[idx[r]], 4, lr[idx[r]]))


# ---------------------------------------------------------------------------------------------------
# pixel-control head
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,A", [(1, 4), (9, 4), (130, 4), (1300, 4), (7, 6), (5, 3), (6, 7)])
def test_pc_deconv_fwd_bwd(ops, N, A):
    rs = np.random.RandomState(N * 10 + A)
    lam, gs = 0.05, 0.25
    hp = torch.tensor(np.maximum(rs.normal(size=(N, 9, 9, 32)), 0), requires_grad=True)
    Wv = torch.tensor(rs.uniform(-.1, .1, (4, 4, 1, 32)), requires_grad=True)
    bv = torch.tensor(rs.uniform(-.1, .1, 1), requires_grad=True)
    Wa = torch.tensor(rs.uniform(-.1, .1, (4, 4, A, 32)), requires_grad=True)
    ba = torch.tensor(rs.uniform(-.1, .1, A), requires_grad=True)
    x = hp.permute(0, 3, 1, 2)
    v = F.relu(F.conv_transpose2d(x, Wv.permute(3, 2, 0, 1), bv, stride=2))
    a = F.relu(F.conv_transpose2d(x, Wa.permute(3, 2, 0, 1), ba, stride=2))
    q = (v + a - a.mean(1, keepdim=True)).permute(0, 2, 3, 1)
    qmax_ref = q.max(3)[0]
    act = rs.randint(0, A, N)
    tgt = torch.tensor(rs.uniform(0, 1, (N, 20, 20)))
    mask = (rs.rand(N) < 0.8).astype(np.int32)
    mask[0] = 1
    m = torch.tensor(mask.astype(np.float64)).reshape(N, 1, 1)
    qa = (q * torch.tensor(np.eye(A)[act]).reshape(N, 1, 1, A)).sum(3)
    loss = lam * 0.5 * (((tgt - qa) ** 2) * m).sum()
    (loss * gs).backward()
    f32 = torch.float32
    d = dict(hp=dev(hp.detach().reshape(-1), f32), Wv=dev(Wv.detach().reshape(-1), f32), bv=dev(bv.detach(), f32),
             Wa=dev(Wa.detach().reshape(-1), f32), ba=dev(ba.detach(), f32))
    qmax = torch.zeros(N * 400, device=DEV)
    ops.pc_deconv_fwd(N, A, d["hp"], d["Wv"], d["bv"], d["Wa"], d["ba"], qmax=qmax)
    close(qmax.reshape(N, 20, 20), qmax_ref, what="qmax")
    d_dec = torch.zeros(N * 400 * (1 + A), device=DEV)
    ls = torch.zeros(1, device=DEV)
    ops.pc_deconv_fwd(N, A, d["hp"], d["Wv"], d["bv"], d["Wa"], d["ba"], action=dev(act, torch.int32),
                      target=dev(tgt.reshape(-1), f32), mask=dev(mask), lam=lam, grad_scale=gs, d_dec=d_dec, loss=ls)
    close(ls, [float(loss) * gs], atol=1e-4, rtol=1e-4, what="pc loss")
    d_hp = torch.zeros(N * 2592, device=DEV)
    dWv = torch.zeros(512, device=DEV); dbv = torch.zeros(1, device=DEV)
    dWa = torch.zeros(512 * A, device=DEV); dba = torch.zeros(A, device=DEV)
    ops.pc_deconv_bwd(N, A, d["hp"], d_dec, d["Wv"], d["Wa"], d_hp, dWv, dbv, dWa, dba)
    tol = dict(atol=1e-5, rtol=1e-5)
    close(d_hp.reshape(N, 9, 9, 32), hp.grad * (hp.detach() > 0), what="d_hp", **tol)
    close(dWv.reshape(4, 4, 1, 32), Wv.grad, what="dWv", **tol)
    close(dWa.reshape(4, 4, A, 32), Wa.grad, what="dWa", **tol)
    close(dbv, bv.grad, what="dbv", **tol)
    close(dba, ba.grad, what="dba", **tol)
    # the same pass in one launch (d_dec kept on chip, per-frame scale): the same fp64 reference at the same bars; its
    # optional d_dec output is the forward kernel's bit for bit
    d_dec2 = torch.zeros_like(d_dec); ls2 = torch.zeros(1, device=DEV); d_hp2 = torch.full_like(d_hp, 7.0)
    dWv2 = torch.zeros(512, device=DEV); dbv2 = torch.zeros(1, device=DEV)
    dWa2 = torch.zeros(512 * A, device=DEV); dba2 = torch.zeros(A, device=DEV); mx = torch.zeros(1, device=DEV)
    ops.pc_deconv_train(N, A, d["hp"], d["Wv"], d["bv"], d["Wa"], d["ba"], dev(act, torch.int32), dev(tgt.reshape(-1), f32),
                        dev(mask), lam, gs, ls2, d_hp2, dWv2, dbv2, dWa2, dba2, dhp_max=mx, d_dec=d_dec2)
    assert torch.equal(d_dec2, d_dec)
    close(ls2, [float(loss) * gs], atol=1e-4, rtol=1e-4, what="pc loss (one launch)")
    close(d_hp2.reshape(N, 9, 9, 32), hp.grad * (hp.detach() > 0), what="d_hp (one launch)", **tol)
    close(dWv2.reshape(4, 4, 1, 32), Wv.grad, what="dWv (one launch)", **tol)
    close(dWa2.reshape(4, 4, A, 32), Wa.grad, what="dWa (one launch)", **tol)
    close(dbv2, bv.grad, what="dbv (one launch)", **tol)
    close(dba2, ba.grad, what="dba (one launch)", **tol)
    assert float(mx) == float(d_hp2.abs().max())
    # without the inspection output: identical results
    ls3 = torch.zeros(1, device=DEV); d_hp3 = torch.zeros_like(d_hp)
    g3 = [torch.zeros(512, device=DEV), torch.zeros(1, device=DEV), torch.zeros(512 * A, device=DEV), torch.zeros(A, device=DEV)]
    ops.pc_deconv_train(N, A, d["hp"], d["Wv"], d["bv"], d["Wa"], d["ba"], dev(act, torch.int32), dev(tgt.reshape(-1), f32),
                        dev(mask), lam, gs, ls3, d_hp3, *g3)
    assert torch.equal(d_hp3, d_hp2)


def test_pc_vr_return_scans(ops):
    rs = np.random.RandomState(21)
    B, H, L = 5, 40, 21
    ring, envs, exps = _run_env(ops, B, H, 90, rs.randint(0, 4, size=(90, B)))
    rt = ring.r_terminal.cpu().numpy().reshape(B, H + 1)
    for b in range(B):
        x = exps[b]
        for i in range(x.top, x.count, 7 + b):
            x.frames[i].terminal = True
            rt[b, i % (H + 1)] = 1
    ring.r_terminal.copy_(dev(rt.reshape(-1)))
    start = rs.randint(0, H - L - 1, size=B).astype(np.int32)
    seq_idx = torch.zeros(L * B, dtype=torch.int32, device=DEV)
    seq_len = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.replay_sample_seq(ring, L, dev(start), seq_idx, seq_len)
    qmax = rs.uniform(0, 1, size=(B, 400)).astype(np.float32)
    bv = rs.normal(size=B).astype(np.float32)
    pcR = torch.zeros((L - 1) * B * 400, device=DEV)
    vrR = torch.zeros((L - 1) * B, device=DEV)
    ops.pc_returns(ring, L, seq_idx, seq_len, dev(qmax.reshape(-1)), 0.9, pcR)
    ops.vr_returns(ring, L, seq_idx, seq_len, dev(bv), 0.99, vrR)
    msk = torch.zeros((L - 1) * B, dtype=torch.int32, device=DEV)
    ops.seq_mask(B, L - 1, seq_len, msk)
    pcR = pcR.cpu().numpy().reshape(L - 1, B, 20, 20); vrR = vrR.cpu().numpy().reshape(L - 1, B)
    msk = msk.cpu().numpy().reshape(L - 1, B)
    for b in range(B):
        x = exps[b]
        fr = [x.frames[i] for i in x.sequence_from_start(int(start[b]), L)]
        rev = fr[::-1]
        pc_R = np.zeros((20, 20), np.float32) if rev[1].terminal else qmax[b].reshape(20, 20)
        vr_R = 0.0 if rev[1].terminal else float(bv[b])
        pcs, vrs = [], []
        for f in rev[1:]:
            pc_R = f.pixel_change + 0.9 * pc_R
            vr_R = f.reward + 0.99 * vr_R
            pcs.append(pc_R); vrs.append(vr_R)
        pcs.reverse(); vrs.reverse()
        n = len(fr)
        assert list(msk[:, b]) == [1] * (n - 1) + [0] * (L - n)
        # the reference mixes float32 (Q-max, first product) and float64 here; the kernel scans in fp64
        close(pcR[:n - 1, b], np.asarray(pcs), atol=1e-7, rtol=1e-6, what="pc_R")
        np.testing.assert_array_equal(vrR[:n - 1, b], np.asarray(vrs, dtype=np.float32))
        assert (pcR[n - 1:, b] == 0).all() and (vrR[n - 1:, b] == 0).all()


# ---------------------------------------------------------------------------------------------------
# optimiser, RNG
# ---------------------------------------------------------------------------------------------------
def test_rmsprop_known_answer_and_oracle(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "rmsprop_known_answer.npz"))
    var = torch.zeros(4, device=DEV); var[:2] = dev(g["var0"], torch.float32)
    ms = torch.ones(4, device=DEV); mom = torch.zeros(4, device=DEV)
    for k in range(2):
        gr = torch.zeros(4, device=DEV); gr[:2] = dev(g["grads"][k], torch.float32)
        ops.rmsprop_step(var, ms, mom, gr, 2.0, 0.9, 0.0, 1.0, 0.0, None)
        close(var[:2], g["steps"][k, :2], atol=0, rtol=1e-6, what="known answer step %d" % k)
    rs = np.random.RandomState(0)
    n = 1898877
    p0 = rs.normal(size=n).astype(np.float32)
    var = dev(p0.copy()); ms = torch.ones(n, device=DEV); mom = torch.zeros(n, device=DEV)
    orc = OracleRMSProp(decay=0.99, momentum=0.0, epsilon=0.1, clip_norm=40.0)
    pl = [p0.copy()]
    scratch = torch.zeros(256, device=DEV); norm = torch.zeros(1, device=DEV)
    for k, scale in enumerate((0.5, 0.001)):       # first step clips (norm ~ 689), second does not (~1.4)
        gr = (rs.normal(size=n) * scale).astype(np.float32)
        gd = dev(gr)
        ops.grad_norm(gd, scratch, norm)
        ops.rmsprop_step(var, ms, mom, gd, 7e-4, 0.99, 0.0, 0.1, 40.0, norm)
        nr = orc.step(pl, [gr], 7e-4)
        close(norm, [np.sqrt((gr.astype(np.float64) ** 2).sum())], atol=0, rtol=2e-6, what="norm")
        close(var, pl[0], atol=1e-7, rtol=1e-6, what="var step %d" % k)
        close(ms, orc.ms[0], atol=1e-7, rtol=1e-6, what="ms")


def test_philox(ops):
    n = 100000
    a = torch.zeros(n, dtype=torch.float64, device=DEV); b = torch.zeros(n, dtype=torch.float64, device=DEV)
    ops.philox_uniform(0xA3C, 1, a); ops.philox_uniform(0xA3C, 1, b)
    assert torch.equal(a, b)
    ops.philox_uniform(0xA3C, 2, b)
    assert not torch.equal(a, b)
    x = a.cpu().numpy()
    assert 0.0 <= x.min() and x.max() < 1.0 and abs(x.mean() - 0.5) < 0.005 and abs(x.var() - 1 / 12) < 0.002
    r = torch.zeros(n, dtype=torch.int32, device=DEV)
    ops.philox_randint(7, 3, 1978, r)
    y = r.cpu().numpy()
    assert y.min() >= 0 and y.max() <= 1977 and abs(y.mean() - 988.5) < 8
    # Philox4x32-10 known-answer (Random123 kat_vectors: counter = key = 0 -> 6627e8d5 e169c58d ...)
    ops.philox_uniform(0, 0, a[:1])
    want = ((0x6627e8d5 >> 5) * 67108864.0 + (0xe169c58d >> 6)) / 9007199254740992.0
    assert float(a.cpu()[0]) == want
    # sharded view: rank r of W reads columns [r*B, (r+1)*B) of every [.., W*B] row of the global draw
    T, B, W = 5, 37, 3
    g = torch.zeros(T * B * W, dtype=torch.float64, device=DEV)
    gi = torch.zeros(T * B * W, dtype=torch.int32, device=DEV)
    ops.philox_uniform(11, 9, g, B * W)
    ops.philox_randint(11, 10, 1000, gi, B * W)
    for rk in range(W):
        s = torch.zeros(T * B, dtype=torch.float64, device=DEV)
        si = torch.zeros(T * B, dtype=torch.int32, device=DEV)
        ops.philox_uniform(11, 9, s, B, B * W, rk * B)
        ops.philox_randint(11, 10, 1000, si, B, B * W, rk * B)
        assert torch.equal(s.view(T, B), g.view(T, B * W)[:, rk * B:(rk + 1) * B])
        assert torch.equal(si.view(T, B), gi.view(T, B * W)[:, rk * B:(rk + 1) * B])
    with pytest.raises(ValueError):
        ops.philox_uniform(11, 9, g, B, B, 1)            # columns outside the row


def test_objective_put_and_fill():
    """Objective vectors ride with the frames (indoor_environment.py:70-73,113; experience.py:42-44): put writes the
    staged vector to each active actor's current slot, fill gathers it (optionally from a neighbouring slot of the same
    actor's ring, wrapping) into the LSTM-input columns."""
    from unreal_amd import ops
    B, H, OBJ, LD, COL0 = 5, 6, 3, 272, 261
    H1 = H + 1
    ring = ops.Ring(B, H, DEV, objective_size=OBJ)
    rs = np.random.RandomState(3)
    want = np.zeros((B, H1, OBJ), np.float32)
    counts = np.array([0, 3, 6, 7, 20], np.int32)
    ring.count.copy_(torch.from_numpy(counts))
    staged = rs.uniform(-1, 1, size=(B, OBJ)).astype(np.float32)
    active = np.array([1, 0, 1, 1, 1], np.int32)
    ops.objective_put(ring, torch.from_numpy(staged.reshape(-1)).to(DEV), torch.from_numpy(active).to(DEV))
    for b in range(B):
        if active[b]:
            want[b, counts[b] % H1] = staged[b]
    np.testing.assert_array_equal(ring.r_objective.cpu().numpy().reshape(B, H1, OBJ), want)
    full = rs.uniform(-1, 1, size=(B, H1, OBJ)).astype(np.float32)
    ring.r_objective.copy_(torch.from_numpy(full.reshape(-1)))
    slots = np.array([0, 6, 3, 0, 5], np.int32)
    idx = (np.arange(B) * H1 + slots).astype(np.int32)
    for off in (0, -1, 1, -8):
        x = torch.full((B * LD,), 9.0, device=DEV)
        ops.objective_fill(ring, B, torch.from_numpy(idx).to(DEV), x, LD, COL0, slot_offset=off)
        got = x.cpu().numpy().reshape(B, LD)
        for b in range(B):
            np.testing.assert_array_equal(got[b, COL0:COL0 + OBJ], full[b, (slots[b] + off) % H1])
        got[:, COL0:COL0 + OBJ] = 9.0
        assert (got == 9.0).all()
    with pytest.raises(ValueError):
        ops.objective_put(ops.Ring(2, 3, DEV), torch.zeros(2, device=DEV))


@pytest.mark.parametrize("A", [3, 4, 6])
def test_policy_step_is_the_three_kernel_path(A):
    """unreal_policy_step == linear_small_fwd (policy) + linear_small_fwd (value) + softmax_sample, bit for bit,
    sampled and greedy, with a padded feature stride."""
    from unreal_amd import ops
    rows, ld = 1000, 264
    rs = np.random.RandomState(A)
    X = torch.as_tensor(rs.normal(size=rows * ld), dtype=torch.float32).to(DEV)
    Wp = torch.as_tensor(rs.normal(size=256 * A) * 0.3, dtype=torch.float32).to(DEV)
    bp = torch.as_tensor(rs.normal(size=A), dtype=torch.float32).to(DEV)
    Wv = torch.as_tensor(rs.normal(size=256) * 0.1, dtype=torch.float32).to(DEV)
    bv = torch.as_tensor(rs.normal(size=1), dtype=torch.float32).to(DEV)
    u = torch.as_tensor(rs.random_sample(rows)).to(DEV)
    for uu in (u, None):
        pi0, v0, a0 = torch.zeros(rows * A, device=DEV), torch.zeros(rows, device=DEV), torch.zeros(rows, dtype=torch.int32, device=DEV)
        ops.linear_small_fwd(rows, 256, A, X, ld, Wp, bp, pi0, A)
        ops.linear_small_fwd(rows, 256, 1, X, ld, Wv, bv, v0, 1)
        ops.softmax_sample(rows, A, pi0, A, uu, a0)
        pi1, v1, a1 = torch.zeros(rows * A, device=DEV), torch.zeros(rows, device=DEV), torch.zeros(rows, dtype=torch.int32, device=DEV)
        ops.policy_step(rows, A, X, ld, Wp, bp, Wv, bv, uu, pi1, v1, a1)
        assert torch.equal(pi0, pi1) and torch.equal(v0, v1) and torch.equal(a0, a1)
    assert len(torch.unique(a1)) > 1


@pytest.mark.parametrize("B", [4096, 1500, 512, 3])       # 8 actors per workgroup / ragged / 2 per workgroup / tiny
def test_fused_policy_maze_rollout_step_is_the_two_launch_path(ops, B):
    """unreal_maze_policy_rollout_step == unreal_policy_step + unreal_maze_rollout_step, bit for bit: pi, V, actions,
    rewards / terminals, the loop bookkeeping, the next step's frame indices and last_action_reward columns, and the
    ring itself (frames, metadata, pixel change) over several chained steps with actors finishing on the way."""
    H, A, xld = 6, 4, 264
    rs = np.random.RandomState(B)
    Wp = dev(rs.uniform(-.3, .3, 256 * A), torch.float32); bp = dev(rs.uniform(-.1, .1, A), torch.float32)
    Wv = dev(rs.uniform(-.3, .3, 256), torch.float32); bv = dev(rs.uniform(-.1, .1, 1), torch.float32)
    rings = [ops.Ring(B, H, DEV), ops.Ring(B, H, DEV)]
    st = []
    for ring in rings:
        ring.frames.zero_(); ring.r_pc.zero_()         # (torch.empty: slots no step writes would hold stale allocator bytes)
        ops.maze_reset(ring)
        pos = ring.pos.cpu()
        pos[0::2] = 5; pos[1::2] = 0                   # one RIGHT from the goal: episodes end inside the test
        pos[0:2 * (B // 2):2] = 0; pos[1:2 * (B // 2):2] = 2
        ring.pos.copy_(pos)
        z = lambda n, dt=torch.int32: torch.zeros(n, dtype=dt, device=DEV)
        st.append(dict(active=torch.ones(B, dtype=torch.int32, device=DEV), log=z(B), n=z(B), te=z(B), r=z(B, torch.float32),
                       t=z(B), a=z(B), pi=z(B * A, torch.float32), v=z(B, torch.float32), idx=z(B),
                       lar=torch.zeros(B * xld, device=DEV)))
    for step in range(5):
        X = dev(rs.uniform(-1, 1, (B, 256)), torch.float32).view(-1)
        u = dev(rs.uniform(0, 1, B), torch.float64)
        s0, s1 = st
        ops.policy_step(B, A, X, 256, Wp, bp, Wv, bv, u, s0["pi"], s0["v"], s0["a"])
        ops.maze_rollout_step(rings[0], s0["a"], s0["r"], s0["t"], s0["active"], s0["log"], s0["n"], s0["te"],
                              next_idx=s0["idx"], next_lar=s0["lar"], lar_ld=xld, lar_col0=256, A=A)
        ops.maze_policy_rollout_step(rings[1], X, 256, Wp, bp, Wv, bv, u, s1["pi"], s1["v"], s1["a"], s1["r"], s1["t"],
                                     s1["active"], s1["log"], s1["n"], s1["te"], next_idx=s1["idx"], next_lar=s1["lar"],
                                     lar_ld=xld, lar_col0=256, A=A)
        for k in s0:
            assert torch.equal(s0[k], s1[k]), (step, k)
        for name in ("frames", "r_reward", "r_action", "r_terminal", "r_last_action", "r_last_reward", "pos", "count",
                     "last_action", "last_reward", "episode_reward", "score_out", "score_valid"):
            assert torch.equal(getattr(rings[0], name), getattr(rings[1], name)), (step, name)
        n_live = (rings[0].count.cpu().numpy() > 0).sum()
        cnt = rings[0].count.cpu().numpy().astype(np.int64)
        H1 = H + 1
        # pixel change of the slots written so far (the rest of r_pc is uninitialised memory)
        for b in (0, B // 2, B - 1):
            for c in range(min(int(cnt[b]), H)):
                a_ = rings[0].r_pc[(b * H1 + c) * 400:(b * H1 + c + 1) * 400]
                b_ = rings[1].r_pc[(b * H1 + c) * 400:(b * H1 + c + 1) * 400]
                assert torch.equal(a_, b_)
    assert int(st[0]["te"].sum()) > 0 and int(st[0]["active"].sum()) < B       # some actors did finish


def test_encoder_fwd_is_bit_reproducible(ops):
    """A frame's conv outputs do not depend on the launch it is part of: the same frame gives the same bits in two launches, at
    a different position of the frame list and in a launch of a different size (the per-kernel scales come from maxima and
    from L1 norms reduced in a fixed order -- ADVICE r3; the partial tiles of the two K halves are added in a fixed order)."""
    rs = np.random.RandomState(3)
    p = _enc_params(4)
    N = 700
    pool = rs.randint(0, 256, size=(N, 84, 84, 3)).astype(np.uint8)
    P = {k: dev(v, torch.float32) for k, v in p.items()}
    fr = dev(pool.reshape(-1))

    def run(idx):
        n = len(idx)
        f2 = torch.zeros(n * 2592, device=DEV); c1 = torch.zeros(n * 6400, device=DEV)
        ops.encoder_fwd(fr, dev(np.asarray(idx, dtype=np.int32)), 1.0 / 255.0, P["W_base_conv1"], P["b_base_conv1"],
                        P["W_base_conv2"], P["b_base_conv2"], f2, c1)
        return f2.view(n, 2592), c1.view(n, 6400)

    a2, a1 = run(np.arange(N))
    b2, b1 = run(np.arange(N))
    assert torch.equal(a2, b2) and torch.equal(a1, b1)
    perm = rs.permutation(N)
    c2, c1_ = run(perm)
    assert torch.equal(c2, a2[torch.as_tensor(perm, device=DEV)]) and torch.equal(c1_, a1[torch.as_tensor(perm, device=DEV)])
    d2, d1 = run(np.arange(5))                                  # five workgroups instead of 512
    assert torch.equal(d2, a2[:5]) and torch.equal(d1, a1[:5])


@pytest.mark.parametrize("N,save_c1", [(700, True), (3, True), (40, False)])
def test_encoder_fwd_with_prepared_weights_is_the_same_kernel(ops, N, save_c1):
    """unreal_encoder_prepare (the weights' share of the forward prologue -- scales and MFMA operand fragments -- once per
    weight update) + unreal_encoder_fwd(prepared=...) against the launch that derives them per workgroup: every output
    bit for bit (f2, the saved conv1 activation, the ReLU bit words, both absmax slots); a block made for other weights
    changes the result (it is really read)."""
    rs = np.random.RandomState(N)
    P = {k: dev(v, torch.float32) for k, v in _enc_params(7).items()}
    fr = dev(rs.randint(0, 256, size=(N, 84, 84, 3)).astype(np.uint8).reshape(-1))
    idx = dev(rs.permutation(N).astype(np.int32))
    scale = 1.0 / 255.0

    def run(prepared):
        f2 = torch.zeros(N * 2592, device=DEV); c1 = torch.zeros(N * 6400, device=DEV) if save_c1 else None
        bits = torch.zeros(N * ops.RELU_WORDS, dtype=torch.int16, device=DEV) if save_c1 else None
        s2, s1 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
        ops.encoder_fwd(fr, idx, scale, P["W_base_conv1"], P["b_base_conv1"], P["W_base_conv2"], P["b_base_conv2"], f2, c1,
                        relu_bits=bits, f2_max=s2, c1_max=s1 if save_c1 else None, prepared=prepared)
        return f2, c1, bits, s2, s1

    prep = ops.encoder_prepare(P["W_base_conv1"], P["b_base_conv1"], P["W_base_conv2"], scale)
    assert prep.numel() == ops.ENC_PREPARED_BYTES
    a, b = run(None), run(prep)
    for x, y in zip(a, b):
        assert (x is None and y is None) or torch.equal(x, y)
    assert float(a[3]) == float(a[0].max()) > 0
    other = ops.encoder_prepare(P["W_base_conv1"] * 1.5, P["b_base_conv1"], P["W_base_conv2"], scale)
    assert not torch.equal(run(other)[0], a[0])
    with pytest.raises((RuntimeError, ValueError)):
        ops.encoder_prepare(P["W_base_conv1"], P["b_base_conv1"], P["W_base_conv2"], scale, prepared=prep[:1000])
