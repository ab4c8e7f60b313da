"""Pin the CPU oracle against golden vectors produced by the reference's own numpy modules
(tests/golden/make_fixtures.py) and the reference's known-answer tests."""
import os

import numpy as np

from oracle import maze as OM
from oracle.experience import OracleExperience, Frame, concat_action_and_reward
from oracle.rmsprop import OracleRMSProp, clip_by_global_norm


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_maze_constants(golden_dir):
    g = _load(golden_dir, "maze_table.npz")
    assert tuple(g["start"]) == OM.START == (0, 2)
    assert tuple(g["goal"]) == OM.GOAL == (6, 0)
    assert len(g["cells"]) == 34
    assert sum(OM.is_wall(x, y) for x in range(7) for y in range(7)) == 15
    np.testing.assert_array_equal(g["maze_image"], OM.maze_image().astype(np.uint8))


def test_maze_transition_table_and_frames(golden_dir):
    g = _load(golden_dir, "maze_table.npz")
    for ci, (x, y) in enumerate(g["cells"]):
        np.testing.assert_array_equal(OM.render(x, y).astype(np.uint8), g["frames"][ci])
        for a in range(4):
            env = OM.OracleMaze()
            env.x, env.y = int(x), int(y)
            env.last_state = {'image': OM.render(env.x, env.y)}
            st, r, t, pc = env.process(a)
            assert (env.x, env.y) == tuple(g["next"][ci, a])
            assert r == g["reward"][ci, a]
            assert bool(t) == bool(g["terminal"][ci, a])
            np.testing.assert_array_equal(pc, g["pixel_change"][ci, a])     # bit-exact float64
    vals = np.unique(np.round(g["pixel_change"] * 12).astype(int))
    assert set(vals) <= {0, 1, 2, 4}        # {0, 1/12, 1/6, 1/3}


def test_maze_random_trace(golden_dir):
    g = _load(golden_dir, "maze_trace.npz")
    env = OM.OracleMaze()
    for i, a in enumerate(g["actions"]):
        _, r, t, pc = env.process(int(a))
        assert (env.x, env.y) == tuple(g["pos"][i])
        assert r == g["reward"][i] and bool(t) == bool(g["terminal"][i])
        assert pc.sum() == g["pc_sum"][i]
        if t:
            env.reset()


def test_shortest_path_is_20():
    from collections import deque
    dist = {OM.START: 0}
    dq = deque([OM.START])
    while dq:
        c = dq.popleft()
        for a in range(4):
            nx, ny, _ = OM.move(c[0], c[1], a)
            if (nx, ny) not in dist:
                dist[(nx, ny)] = dist[c] + 1
                dq.append((nx, ny))
    assert dist[OM.GOAL] == 20


def _frame(i, reward, terminal):
    return Frame({"id": i}, reward, i % 4, terminal, None, (i + 3) % 4, 0)


def test_replay_unit_test_scenario(golden_dir):
    """The reference's own unit test (train/experience_test.py:16-36) + probed bucket contents."""
    g = _load(golden_dir, "replay_traces.npz")
    rs = np.random.RandomState(0xA3C)
    exp = OracleExperience(10, rs)
    for i in range(10):
        exp.add_frame(_frame(i, 1 if i == 5 else 0, False))
    assert exp.is_full() and exp.top == 0 == g["t10_top"][0]
    assert exp.bucket(True) == list(g["t10_pos"]) == [5]
    assert exp.bucket(False) == list(g["t10_neg"]) == [3, 4, 6, 7, 8, 9]
    exp.add_frame(_frame(10, 0, False))
    assert exp.top == 1 == g["t11_top"][0]
    assert exp.bucket(False) == list(g["t11_neg"]) == [4, 6, 7, 8, 9, 10]
    assert exp.bucket(True) == list(g["t11_pos"])
    for k in range(100):
        fr = exp.sample_rp_sequence()
        assert len(fr) == 4
        assert [f.state["id"] for f in fr] == list(g["t11_rp_ids"][k])


def test_replay_scripted_stream(golden_dir):
    g = _load(golden_dir, "replay_traces.npz")
    H = int(g["s_H"][0])
    rs = np.random.RandomState(0xA3C)
    exp = OracleExperience(H, rs)
    k = 0
    for i, (r, t) in enumerate(zip(g["s_rewards"], g["s_terminals"])):
        exp.add_frame(_frame(i, int(r), bool(t)))
        assert exp.top == g["s_top"][i]
        assert len(exp) == g["s_len"][i]
        assert len(exp.bucket(True)) == g["s_npos"][i]
        assert len(exp.bucket(False)) == g["s_nneg"][i]
        if k < len(g["s_sample_at"]) and i == g["s_sample_at"][k]:
            fr = exp.sample_sequence(21)
            ids = [f.state["id"] for f in fr] + [-1] * (21 - len(fr))
            assert ids == list(g["s_seq_ids"][k])
            fr = exp.sample_rp_sequence()
            assert [f.state["id"] for f in fr] == list(g["s_rp_ids"][k])
            k += 1
    assert k == len(g["s_sample_at"]) > 50
    # frame ids map through the discarded double-terminal frame: ids keep the caller's numbering
    assert [exp.frames[i].state["id"] for i in exp.bucket(True)] != []
    assert len(exp.bucket(True)) == len(g["s_final_pos"])
    assert exp.bucket(True) == list(g["s_final_pos"]) and exp.bucket(False) == list(g["s_final_neg"])


def test_concat_action_and_reward(golden_dir):
    g = _load(golden_dir, "replay_traces.npz")
    np.testing.assert_array_equal(concat_action_and_reward(2, 4, -1), g["car_2_4_m1"])
    np.testing.assert_array_equal(concat_action_and_reward(2, 4, -1), [0, 0, 1, 0, -1])
    np.testing.assert_array_equal(concat_action_and_reward(0, 4, 0), g["car_0_4_0"])
    np.testing.assert_array_equal(concat_action_and_reward(1, 3, 1, np.array([0.5, 0.25])), g["car_obj"])


def test_rmsprop_known_answer(golden_dir):
    """train/rmsprop_applier_test.py:29-51 (lr=2, decay=.9, eps=1, rms0=1, no clipping in that test)."""
    g = _load(golden_dir, "rmsprop_known_answer.npz")
    opt = OracleRMSProp(decay=0.9, momentum=0.0, epsilon=1.0, dtype=np.float64)
    var = [g["var0"].copy()]
    for k in range(2):
        opt.step(var, [g["grads"][k]], 2.0, clip=False)
        np.testing.assert_allclose(var[0], g["steps"][k, :2], rtol=1e-12)
        np.testing.assert_allclose(opt.ms[0], g["steps"][k, 2:], rtol=1e-12)
    opt32 = OracleRMSProp(decay=0.9, momentum=0.0, epsilon=1.0, dtype=np.float32)
    v32 = [g["var0"].astype(np.float32)]
    for k in range(2):
        opt32.step(v32, [g["grads"][k].astype(np.float32)], 2.0, clip=False)
    np.testing.assert_allclose(v32[0], g["steps"][1, :2], rtol=1e-6)


def test_clip_by_global_norm():
    g = [np.full(100, 3.0, np.float32), np.full(44, 4.0, np.float32)]
    clipped, norm = clip_by_global_norm(g, 40.0)
    np.testing.assert_allclose(norm, np.sqrt(900 + 704), rtol=1e-6)
    np.testing.assert_allclose(np.sqrt(sum((c ** 2).sum() for c in clipped)), 40.0, rtol=1e-6)
    small, n2 = clip_by_global_norm([np.ones(4, np.float32)], 40.0)
    np.testing.assert_array_equal(small[0], np.ones(4, np.float32))   # scale is exactly 1


# ---------------------------------------------------------------------------------------------------
# upstream (Lab) replay and the generic pixel change: fixtures made by importing the reference's
# train/experience_lab_ver.py and environment/environment.py (tests/golden/make_fixtures.py)
# ---------------------------------------------------------------------------------------------------
def lab_u8_pairs(seed, n):
    """The uint8 frame pairs of pixel_change_u8.npz, regenerated exactly as make_fixtures.pixel_change_fixture does."""
    rs = np.random.RandomState(seed)
    u8 = rs.randint(0, 256, size=(n, 2, 84, 84, 3)).astype(np.uint8)
    u8[1, 1] = u8[1, 0]
    u8[2, 0] = 0
    u8[2, 1] = 255
    u8[3, 1] = u8[3, 0]
    u8[3, 1, 40:44, 40:44, :] ^= 0x80
    for k in range(4, 20):
        u8[k, 1] = u8[k, 0]
        y, x = rs.randint(0, 70, size=2)
        u8[k, 1, y:y + 12, x:x + 12, :] = rs.randint(0, 256, size=(12, 12, 3))
    return u8


def test_lab_replay_scripted_stream(golden_dir):
    """oracle/experience.py (lab_ver=True + clip_frame) vs train/experience_lab_ver.py:14,18,53-55,76-80,100-151 on
    a scripted stream: clipped stored rewards, window top / length, zero / non-zero bucket sizes after EVERY add, and
    the sampled frame ids of sample_sequence(21) / sample_rp_sequence() under the same global-RandomState draws."""
    from oracle.experience import clip_frame
    g = _load(golden_dir, "replay_lab_ver.npz")
    H = int(g["H"][0])
    raw, terms = g["raw_reward"], g["terminals"]
    np.random.seed(0xA3C)                                   # the reference draws from the global RandomState
    exp = OracleExperience(H, np.random, lab_ver=True)
    k = 0
    stored = []
    for i in range(len(raw)):
        f = clip_frame(Frame({"id": i}, raw[i], i % 6, bool(terms[i]), None, (i + 5) % 6, raw[i - 1] if i else 0.0))
        if exp.add_frame(f):
            stored.append((i, f.reward, f.last_reward))
        assert exp.top == g["top"][i] and len(exp) == g["length"][i]
        assert len(exp.bucket(False)) == g["n_zero"][i]            # lab_ver: bucket(False) = zero rewards
        assert len(exp.bucket(True)) == g["n_nonzero"][i]
        if k < len(g["sample_at"]) and i == g["sample_at"][k]:
            fr = exp.sample_sequence(21)
            assert [x.state["id"] for x in fr] + [-1] * (21 - len(fr)) == list(g["seq_ids"][k])
            fr = exp.sample_rp_sequence()
            assert [x.state["id"] for x in fr] == list(g["rp_ids"][k])
            k += 1
    assert k == len(g["sample_at"]) > 50
    assert [s[0] for s in stored] == list(g["stored_id"]) and 101 not in g["stored_id"]    # successive terminals dropped
    assert len(raw) - 10 < len(stored) < len(raw)
    np.testing.assert_array_equal([s[1] for s in stored], g["stored_reward"])
    np.testing.assert_array_equal([s[2] for s in stored], g["stored_last_reward"])
    assert set(np.unique(g["stored_reward"])) == {-1.0, -0.5, 0.0, 0.5, 1.0}                # +-3, 2.5 were clipped
    assert exp.bucket(False) == list(g["final_zero"]) and exp.bucket(True) == list(g["final_nonzero"])
    # the logged draws reproduce the picks through the explicit-draw entry points the device test uses
    np.testing.assert_array_equal(
        clip_frame(Frame({}, 2.5, 1, False, None, 3, -7.0)).get_last_action_reward(6), g["car_clip"])


def test_pixel_change_on_lab_frames(golden_dir):
    """oracle.maze.calc_pixel_change vs Environment._calc_pixel_change (environment.py:88-99) on 96 uint8/255 float32
    frame pairs, bit for bit (same numpy expression on the same float32 inputs); and the exact integer form the
    device uses (sum of |a-b| over 4x4x3 bytes / (48*255)) within 2e-7 of the reference's float32 result."""
    g = _load(golden_dir, "pixel_change_u8.npz")
    n = int(g["n"][0])
    u8 = lab_u8_pairs(int(g["seed"][0]), n)
    assert int(u8.astype(np.uint64).sum()) == int(g["checksum"][0])
    for k in range(n):
        a = u8[k, 0].astype(np.float32) / 255.0
        b = u8[k, 1].astype(np.float32) / 255.0
        pc = OM.calc_pixel_change(b, a)
        assert pc.dtype == np.float32
        np.testing.assert_array_equal(pc, g["pixel_change"][k])
    d = np.abs(u8[:, 1, 2:-2, 2:-2, :].astype(np.int64) - u8[:, 0, 2:-2, 2:-2, :].astype(np.int64))
    sad = d.reshape(n, 20, 4, 20, 4, 3).sum(axis=(2, 4, 5))
    exact = (sad / (48.0 * 255.0)).astype(np.float32)
    assert np.abs(exact.astype(np.float64) - g["pixel_change"]).max() <= 2e-7
    assert (g["pixel_change"][1] == 0).all() and np.abs(g["pixel_change"][2] - 1.0).max() <= 2e-7
