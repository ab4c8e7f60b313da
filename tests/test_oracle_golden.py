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
