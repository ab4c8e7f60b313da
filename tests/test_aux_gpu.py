"""GPU tests of the widened rows of SURVEY 8(f): checkpoint/resume (f-2) and the evaluation harness (f-3)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import maze as OM
from oracle.experience import concat_action_and_reward
from oracle.trainer import OracleActor, ExplicitDraws
from tests.test_trainer_gpu import _cfg, _build, RecordingDraws

DEV = "cuda:0"


def test_checkpoint_round_trip(tmp_path):
    from unreal_amd import checkpoint as ck
    cfg = _cfg(True, True, 40, 20)
    net, applier, tr, draws = _build(cfg, 2, seed=21)
    while not tr._full:
        tr.process(None, 0)
    steps, _ = tr.process(None, 0)
    d = str(tmp_path / "lab_ckpt")
    path = ck.save(d, net, applier, global_t=steps, wall_t=3.25, best_score=-0.25)
    assert path.endswith("checkpoint-25-%d.pt" % steps)
    net2, applier2, tr2, _ = _build(cfg, 2, seed=99)
    assert not torch.equal(net2.params.flat, net.params.flat)
    g, w, s = ck.restore(d, net2, applier2)
    assert (g, w, s) == (steps, 3.25, -0.25)
    assert torch.equal(net2.params.flat, net.params.flat)                 # bit-exact
    assert torch.equal(applier2.ms, applier.ms) and torch.equal(applier2.mom, applier.mom)
    # the restored learner keeps training (LR schedule continues from the parsed global_t)
    while not tr2._full:
        tr2.process(None, g)
    d2, _ = tr2.process(None, g)
    assert d2 > 0 and np.isfinite(tr2.last_losses["total_loss"])
    assert tr2._anneal_learning_rate(g) < tr2.initial_learning_rate


@pytest.mark.parametrize("greedy", [False, True])
def test_evaluate_matches_oracle_rollout(greedy):
    """The evaluator's lock-step device rollout vs the oracle network + oracle maze driven by the same draws:
    identical actions, episode returns, lengths and success flags."""
    from unreal_amd.evaluate import Evaluate
    cfg = _cfg(True, True, 40, 20)
    net, applier, tr, _ = _build(cfg, 1, seed=31)
    B, n_ep, cap = 4, 7, 45
    draws = RecordingDraws(__import__("unreal_amd.train.trainer", fromlist=["PhiloxDraws"]).PhiloxDraws(5, 0))
    ev = Evaluate(net, batch_size=B, device=DEV, greedy=greedy, draws=draws)
    res = ev.process(n_ep, max_episode_steps=cap)

    params = {k: torch.tensor(v, dtype=torch.float64) for k, v in net.export_named().items()}
    actors = [OracleActor(cfg, ExplicitDraws(), dtype=torch.float64) for _ in range(B)]
    steps = [0] * B
    ep_r = [0.0] * B
    returns, lengths, succ, timeouts, done, k = [], [], 0, 0, 0, 0
    while done < n_ep:
        u = draws.log[k] if not greedy else None
        k += 1
        for b, a in enumerate(actors):
            lar = concat_action_and_reward(a.env.last_action, 4, a.env.last_reward)
            pi, _ = a.run_base_policy_and_value(params, a.env.last_state, lar)
            if greedy:
                act = int(np.argmax(pi))
            else:
                a.draws.action_u = [float(u[b])]
                act = a.draws.choose_action(pi)
            _, r, t, _ = a.env.process(act)
            ep_r[b] += r
            steps[b] += 1
            if t:
                returns.append(ep_r[b]); lengths.append(steps[b]); succ += 1; done += 1
                ep_r[b], steps[b] = 0.0, 0
                a.env.reset(); a.reset_state()
            elif steps[b] >= cap:
                returns.append(ep_r[b]); lengths.append(cap); timeouts += 1; done += 1
                ep_r[b], steps[b] = 0.0, 0
                a.env.reset(); a.reset_state()
    assert res["episodes"] == len(returns) and res["timeouts"] == timeouts
    assert abs(res["mean_return"] - sum(returns) / len(returns)) < 1e-9
    assert abs(res["mean_length"] - sum(lengths) / len(lengths)) < 1e-9
    assert abs(res["success_rate"] - succ / len(returns)) < 1e-12
