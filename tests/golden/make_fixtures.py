#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's numpy-only modules.

Runs only in the build container (needs /root/reference on disk); the outputs
(`*.npz`, small) are committed and are the only thing that travels.  Nothing from
the reference source is copied: these are inputs/outputs of its functions.

Reference functions exercised (file:line under /root/reference):
  environment/maze_environment.py:30-128   MazeEnvironment (_setup, reset, _move, process)
  environment/environment.py:88-102        Environment._calc_pixel_change / _subsample
  train/experience.py:10-153               ExperienceFrame, Experience
  train/experience_lab_ver.py:10-152       upstream replay (reward clip, zero / non-zero buckets, global np.random)
  environment/environment.py:88-102        _calc_pixel_change on uint8/255 float32 frames (the Lab wrapper's frames)

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py
"""
import os
import sys
import io
import contextlib

import numpy as np

REF = os.environ.get("UNREAL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from environment.maze_environment import MazeEnvironment  # noqa: E402
from train.experience import Experience, ExperienceFrame  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def maze_fixture():
    env = MazeEnvironment()
    free = [(x, y) for y in range(7) for x in range(7) if not env._is_wall(x, y)]
    cells = np.array(free, dtype=np.int32)                      # [34, 2] (x, y)
    frames = np.zeros((len(free), 84, 84, 3), dtype=np.uint8)
    nxt = np.zeros((len(free), 4, 2), dtype=np.int32)
    rew = np.zeros((len(free), 4), dtype=np.int32)
    term = np.zeros((len(free), 4), dtype=np.uint8)
    pc = np.zeros((len(free), 4, 20, 20), dtype=np.float64)
    for ci, (x, y) in enumerate(free):
        env.x, env.y = x, y
        img = env._get_current_image()
        assert set(np.unique(img)) <= {0.0, 1.0}
        frames[ci] = img.astype(np.uint8)
        for a in range(4):
            env.x, env.y = x, y
            env.last_state = {"image": env._get_current_image()}
            s, r, t, p = env.process(a)
            nxt[ci, a] = (env.x, env.y)
            rew[ci, a] = r
            term[ci, a] = t
            pc[ci, a] = p
    env.reset()
    np.savez_compressed(
        os.path.join(OUT, "maze_table.npz"),
        cells=cells, frames=frames, next=nxt, reward=rew, terminal=term, pixel_change=pc,
        start=np.array(env._start_pos, dtype=np.int32), goal=np.array(env._goal_pos, dtype=np.int32),
        maze_image=env._maze_image.astype(np.uint8),
    )

    # 5000-step random-action trace with the Trainer's reset-on-terminal rule
    rs = np.random.RandomState(1234)
    env = MazeEnvironment()
    acts = rs.randint(0, 4, size=5000).astype(np.int32)
    pos = np.zeros((5000, 2), np.int32)
    rews = np.zeros(5000, np.int32)
    terms = np.zeros(5000, np.uint8)
    pcsum = np.zeros(5000, np.float64)
    for i, a in enumerate(acts):
        s, r, t, p = env.process(int(a))
        pos[i] = (env.x, env.y)
        rews[i] = r
        terms[i] = t
        pcsum[i] = p.sum()
        if t:
            env.reset()
    # a scripted shortest path to the goal (20 moves) for an end-to-end episode
    np.savez_compressed(os.path.join(OUT, "maze_trace.npz"), actions=acts, pos=pos, reward=rews,
                        terminal=terms, pc_sum=pcsum)


def _frame(i, reward, terminal):
    # state carries an integer id so samples can be identified
    return ExperienceFrame({"id": i}, reward, i % 4, terminal, None, (i + 3) % 4, 0)


def replay_fixture():
    out = {}
    # (a) the reference's own unit test scenario (train/experience_test.py:16-36)
    rs = np.random.RandomState(0xA3C)
    exp = Experience(10, rs)
    for i in range(10):
        exp.add_frame(_frame(i, 1 if i == 5 else 0, False))
    out["t10_full"] = np.array([exp.is_full()])
    out["t10_top"] = np.array([exp._top_frame_index])
    out["t10_pos"] = np.array(exp._pos_reward_indices)
    out["t10_neg"] = np.array(exp._neg_reward_indices)
    exp.add_frame(_frame(10, 0, False))
    out["t11_top"] = np.array([exp._top_frame_index])
    out["t11_pos"] = np.array(exp._pos_reward_indices)
    out["t11_neg"] = np.array(exp._neg_reward_indices)
    rp = []
    for _ in range(100):
        fr = exp.sample_rp_sequence()
        assert len(fr) == 4
        rp.append([f.state["id"] for f in fr])
    out["t11_rp_ids"] = np.array(rp)

    # (b) scripted stream, H=64, rewards in {-1,0,+1}, terminals sprinkled (incl. a double terminal)
    H = 64
    n_add = 400
    srs = np.random.RandomState(7)
    rewards = srs.choice([-1, 0, 0, 0, 1], size=n_add)
    terms = (srs.random_sample(n_add) < 0.08)
    terms[100] = terms[101] = True         # successive terminals -> second must be discarded
    rs = np.random.RandomState(0xA3C)
    exp = Experience(H, rs)
    tops, npos, nneg, lens = [], [], [], []
    seq_ids = []      # sample_sequence(21) results, padded with -1
    rp_ids = []
    sample_at = []
    with contextlib.redirect_stdout(io.StringIO()):
        for i in range(n_add):
            exp.add_frame(_frame(i, int(rewards[i]), bool(terms[i])))
            tops.append(exp._top_frame_index)
            npos.append(len(exp._pos_reward_indices))
            nneg.append(len(exp._neg_reward_indices))
            lens.append(len(exp._frames))
            if exp.is_full() and i % 3 == 0:
                fr = exp.sample_sequence(21)
                ids = [f.state["id"] for f in fr] + [-1] * (21 - len(fr))
                seq_ids.append(ids)
                fr = exp.sample_rp_sequence()
                rp_ids.append([f.state["id"] for f in fr])
                sample_at.append(i)
    out.update(
        s_H=np.array([H]), s_rewards=rewards.astype(np.int32), s_terminals=terms.astype(np.uint8),
        s_top=np.array(tops), s_npos=np.array(npos), s_nneg=np.array(nneg), s_len=np.array(lens),
        s_seq_ids=np.array(seq_ids), s_rp_ids=np.array(rp_ids), s_sample_at=np.array(sample_at),
        s_final_pos=np.array(exp._pos_reward_indices), s_final_neg=np.array(exp._neg_reward_indices),
    )

    # (c) concat_action_and_reward known answers (train/experience.py:35-46)
    out["car_2_4_m1"] = ExperienceFrame.concat_action_and_reward(2, 4, -1, {})
    out["car_0_4_0"] = ExperienceFrame.concat_action_and_reward(0, 4, 0, {})
    out["car_obj"] = ExperienceFrame.concat_action_and_reward(1, 3, 1, {"objective": np.array([0.5, 0.25])})
    np.savez_compressed(os.path.join(OUT, "replay_traces.npz"), **out)


def lab_replay_fixture():
    """Upstream replay used for Lab (train/experience_lab_ver.py): rewards clipped to [-1, 1] when the frame is built
    (:14,18), zero / non-zero reward buckets (:76-80), draws from the GLOBAL numpy RandomState (:102,124,138,141).
    The draws are logged by wrapping np.random.randint around the calls (the reference itself is untouched)."""
    import train.experience_lab_ver as LV
    H, n_add = 64, 400
    srs = np.random.RandomState(17)
    raw = srs.choice([-3.0, -1.0, -0.5, 0.0, 0.0, 0.0, 0.0, 0.5, 1.0, 2.5], size=n_add)
    terms = (srs.random_sample(n_add) < 0.08)
    terms[100] = terms[101] = True            # successive terminals -> second must be discarded
    terms[200:206] = False
    np.random.seed(0xA3C)
    log = []
    real_randint = np.random.randint

    def logging_randint(*a, **k):
        v = real_randint(*a, **k)
        log.append((a[-1] if a else k.get("high"), int(v)))
        return v

    exp = LV.Experience(H, None)
    tops, nzero, nnon, lens = [], [], [], []
    st_reward, st_last_reward, st_id = [], [], []
    seq_ids, seq_start, rp_ids, rp_coin, rp_pick, rp_n, sample_at = [], [], [], [], [], [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for i in range(n_add):
            last_r = raw[i - 1] if i else 0.0
            f = LV.ExperienceFrame({"id": i}, raw[i], i % 6, bool(terms[i]), None, (i + 5) % 6, last_r)
            before = len(exp._frames) + exp._top_frame_index
            exp.add_frame(f)
            if len(exp._frames) + exp._top_frame_index != before:          # the frame was accepted
                st_reward.append(float(f.reward)); st_last_reward.append(float(f.last_reward)); st_id.append(i)
            tops.append(exp._top_frame_index)
            nzero.append(len(exp._zero_reward_indices))
            nnon.append(len(exp._non_zero_reward_indices))
            lens.append(len(exp._frames))
            if exp.is_full() and i % 3 == 0:
                np.random.randint = logging_randint
                try:
                    del log[:]
                    fr = exp.sample_sequence(21)
                    seq_start.append(log[0][1])
                    seq_ids.append([x.state["id"] for x in fr] + [-1] * (21 - len(fr)))
                    del log[:]
                    fr = exp.sample_rp_sequence()
                    rp_coin.append(log[0][1]); rp_n.append(log[1][0]); rp_pick.append(log[1][1])
                    rp_ids.append([x.state["id"] for x in fr])
                finally:
                    np.random.randint = real_randint
                sample_at.append(i)
    np.savez_compressed(
        os.path.join(OUT, "replay_lab_ver.npz"), H=np.array([H]), raw_reward=raw, terminals=terms.astype(np.uint8),
        top=np.array(tops), n_zero=np.array(nzero), n_nonzero=np.array(nnon), length=np.array(lens),
        stored_reward=np.array(st_reward), stored_last_reward=np.array(st_last_reward), stored_id=np.array(st_id),
        seq_ids=np.array(seq_ids), seq_start=np.array(seq_start), rp_ids=np.array(rp_ids), rp_coin=np.array(rp_coin),
        rp_pick=np.array(rp_pick), rp_n=np.array(rp_n), sample_at=np.array(sample_at),
        final_zero=np.array(exp._zero_reward_indices), final_nonzero=np.array(exp._non_zero_reward_indices),
        car_clip=LV.ExperienceFrame({}, 2.5, 1, False, None, 3, -7.0).get_last_action_reward(6))


def pixel_change_fixture():
    """Environment._calc_pixel_change (environment.py:93-99) on frames as the Lab wrapper makes them
    (lab_environment.py:99-102: uint8 / 255 as float32).  Inputs are regenerated from the seed by the tests; only
    the outputs (float32 [N,20,20]) and an input checksum are stored."""
    env = MazeEnvironment()
    rs = np.random.RandomState(4242)
    N = 96
    u8 = rs.randint(0, 256, size=(N, 2, 84, 84, 3)).astype(np.uint8)
    u8[1, 1] = u8[1, 0]                                   # identical frames -> exact zeros
    u8[2, 0] = 0; u8[2, 1] = 255                          # maximum change -> exact ones
    u8[3, 1] = u8[3, 0]; u8[3, 1, 40:44, 40:44, :] ^= 0x80  # one 4x4 block
    for k in range(4, 20):                                # sparse changes like a moving sprite
        u8[k, 1] = u8[k, 0]
        y, x = rs.randint(0, 70, size=2)
        u8[k, 1, y:y + 12, x:x + 12, :] = rs.randint(0, 256, size=(12, 12, 3))
    out = np.zeros((N, 20, 20), np.float32)
    for k in range(N):
        a = u8[k, 0].astype(np.float32) / 255.0
        b = u8[k, 1].astype(np.float32) / 255.0
        pc = env._calc_pixel_change(b, a)
        assert pc.dtype == np.float32 and pc.shape == (20, 20)
        out[k] = pc
    np.savez_compressed(os.path.join(OUT, "pixel_change_u8.npz"), seed=np.array([4242]), n=np.array([N]),
                        pixel_change=out, checksum=np.array([int(u8.astype(np.uint64).sum())]))


def known_answers():
    # RMSProp known-answer arithmetic as written in train/rmsprop_applier_test.py:29-51
    # (the test itself needs TensorFlow; these are the values its assertions compute).
    import math
    ms_x = ms_y = 1.0
    x, y = 1.0, 2.0
    steps = []
    for dx, dy in ((2.0, 4.0), (3.0, 6.0)):
        ms_x = ms_x + (dx * dx - ms_x) * (1.0 - 0.9)
        ms_y = ms_y + (dy * dy - ms_y) * (1.0 - 0.9)
        x = x - (2.0 * dx / math.sqrt(ms_x + 1.0))
        y = y - (2.0 * dy / math.sqrt(ms_y + 1.0))
        steps.append([x, y, ms_x, ms_y])
    np.savez_compressed(os.path.join(OUT, "rmsprop_known_answer.npz"), steps=np.array(steps),
                        var0=np.array([1.0, 2.0]), grads=np.array([[2.0, 4.0], [3.0, 6.0]]),
                        lr=np.array([2.0]), decay=np.array([0.9]), eps=np.array([1.0]))


if __name__ == "__main__":
    maze_fixture()
    replay_fixture()
    lab_replay_fixture()
    pixel_change_fixture()
    known_answers()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
