#!/usr/bin/env python3
"""Config 1 (maze, FF, no aux, 1 actor): how long do the device's per-update losses track the oracle's on the same draws?
Prints the relative loss difference at checkpoints and the first update whose actions differ (if any).  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_trainer_gpu as T
from oracle.trainer import OracleTrainer, ExplicitDraws
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dt = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
B, H, TT = 1, 30, 20
cfg = T._cfg(False, False, H, TT)
cfg["initial_learning_rate"] = 7.0711e-4
net, applier, tr, draws = T._build(cfg, B, seed=3)
params = {k: torch.tensor(v, dtype=dt) for k, v in net.export_named().items()}
ed = [ExplicitDraws() for _ in range(B)]
orc = OracleTrainer(cfg, n_actors=B, draws=ed, dtype=dt, params=params)
while not tr._full:
    tr.process(None, 0)
for step_u in draws.log:
    ed[0].action_u.append(float(step_u[0]))
orc.fill()
g_t, worst, t0 = 0, 0.0, time.time()
for it in range(N):
    draws.log.clear()
    steps, _ = tr.process(None, g_t)
    ld = tr.last_losses
    ed[0].action_u = [float(x) for x in draws.log[0].reshape(-1)]
    steps_o, infos, losses_o, mean_g, norm_o = orc.process_batched(g_t)
    acts = tr.actions.cpu().numpy().reshape(-1)[:infos[0]["n"]]
    if steps != steps_o or list(acts) != infos[0]["actions"]:
        print("update %d: trajectories differ (steps %d vs %d)" % (it, steps, steps_o)); break
    rel = max(abs(ld[k] - losses_o[0][k]) / (abs(losses_o[0][k]) + 1e-3) for k in ("policy_loss", "value_loss", "total_loss"))
    pd = max(float(np.abs(net.p[n].cpu().double().numpy() - ref.double().numpy().reshape(-1)).max()) for n, ref in orc.params.items())
    worst = max(worst, rel)
    if it in (0, 9, 99, 299, 499, 999, 1999) or it == N - 1:
        print("update %4d: loss rel diff %.2e (worst so far %.2e), max |param diff| %.2e, total_loss %.4f, %.0f s" % (
            it + 1, rel, worst, pd, ld["total_loss"], time.time() - t0), flush=True)
    g_t += steps
