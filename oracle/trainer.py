"""Oracle: per-actor `Trainer.process()` loop on the CPU restatement.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/train/trainer.py:
  _fill_experience 176-205, _process_base 218-336, _process_pc 339-380, _process_vr 383-412,
  _process_rp 415-436, process 438-636, _anneal_learning_rate 140-144, choose_action 147-148
with the maze adapter semantics of SURVEY H1 (flag ignored; state wrapped as {'image': ...};
success := terminal).  One `OracleActor` is one reference thread.  `OracleTrainer` owns the shared
("global") parameters and the shared RMSProp slots (main.py:217-274, rmsprop_applier.py:61-65).

Two ways of combining actors are provided:
  * process_async(i, global_t): exactly the reference -- actor i computes its own gradient from
    the current shared parameters, clips it at clip_norm and applies RMSProp immediately (hogwild
    order = call order).  Used for the CPU baseline and config-1 (B = 1) parity.
  * process_batched(global_t): the batched-learner semantics of the build: every
    actor rolls out with the SAME parameters, the per-actor gradients (sums over time) are
    averaged over actors, clipped once, one RMSProp step.  With one actor both coincide.
"""
import numpy as np
import torch

from . import model as M
from .maze import OracleMaze
from .experience import OracleExperience, Frame, concat_action_and_reward, clip_frame
from .rmsprop import OracleRMSProp


def log_uniform(lo, hi, rate):                       # main.py:61-65
    return float(np.exp(np.log(lo) * (1 - rate) + np.log(hi) * rate))


class RefDraws(object):
    """Reference draw order from one shared RandomState (main.py:213)."""

    def __init__(self, random_state):
        self.rs = random_state

    def choose_action(self, pi):
        return int(self.rs.choice(len(pi), p=pi))    # trainer.py:147-148

    def seq_start(self, high):
        return int(self.rs.randint(0, high))         # experience.py:103

    def rp(self):
        coin = int(self.rs.randint(2))               # experience.py:125
        return coin, (lambda n: int(self.rs.randint(n)))


class ExplicitDraws(object):
    """Draws supplied from outside (recorded device draws): uniforms for actions and the RP pick,
    integers for sequence starts and the RP coin."""

    def __init__(self):
        self.action_u = []
        self.seq_starts = []
        self.rp_coin = []
        self.rp_u = []

    def choose_action(self, pi):
        u = self.action_u.pop(0)
        cdf = np.cumsum(np.asarray(pi, dtype=np.float64))
        cdf /= cdf[-1]
        return int(np.searchsorted(cdf, u, side='right'))

    def seq_start(self, high):
        s = int(self.seq_starts.pop(0))
        assert 0 <= s < high
        return s

    def rp(self):
        coin = int(self.rp_coin.pop(0))
        u = self.rp_u.pop(0)
        return coin, (lambda n: min(n - 1, int(u * n)))


class OracleActor(object):
    def __init__(self, cfg, draws, dtype=torch.float32, env=None):
        self.cfg = cfg
        self.A = cfg["action_size"]
        self.draws = draws
        self.dtype = dtype
        self.env = env if env is not None else OracleMaze()
        self.lab_ver = bool(cfg.get("lab_ver", False))      # upstream replay (experience_lab_ver.py)
        self.exp = OracleExperience(cfg["experience_history_size"], lab_ver=self.lab_ver)
        self.local_t = 0
        self.episode_reward = 0
        self.reset_state()

    def reset_state(self):                           # model.py:625-628
        z = torch.zeros(256, dtype=self.dtype)
        self.lstm_state = (z, z.clone())

    def _img(self, state):
        return torch.tensor(state['image'], dtype=self.dtype).unsqueeze(0)

    def _lar(self, v):
        return torch.tensor(v, dtype=self.dtype).unsqueeze(0)

    @torch.no_grad()
    def run_base_policy_and_value(self, p, state, lar, advance=True):    # model.py:630-660 / 687-704
        feat, st = M.trunk(self._img(state), self._lar(lar), p, self.cfg["use_lstm"],
                           self.lstm_state if self.cfg["use_lstm"] else None)
        pi, v = M.policy_value(feat, p)
        if advance and self.cfg["use_lstm"]:
            self.lstm_state = st
        return pi[0].numpy().astype(np.float32), float(v[0])

    def fill_step(self, p):                          # trainer.py:176-205
        env = self.env
        prev_state, last_action, last_reward = env.last_state, env.last_action, env.last_reward
        lar = concat_action_and_reward(last_action, self.A, last_reward, prev_state.get('objective'))
        pi, _ = self.run_base_policy_and_value(p, prev_state, lar)
        action = self.draws.choose_action(pi)
        _, reward, terminal, pc = env.process(action)
        self.exp.add_frame(self._frame(prev_state, reward, action, terminal, pc, last_action, last_reward))
        if terminal:
            env.reset()
        if self.exp.is_full():
            env.reset()

    def _frame(self, *a):
        f = Frame(*a)
        return clip_frame(f) if self.lab_ver else f

    def process_base(self, p):                       # trainer.py:218-336
        cfg, env = self.cfg, self.env
        states, lars, actions, rewards, values = [], [], [], [], []
        terminal_end = False
        start_state = self.lstm_state
        score = None
        for _ in range(cfg["n_step_TD"]):
            last_action, last_reward = env.last_action, env.last_reward
            lar = concat_action_and_reward(last_action, self.A, last_reward, env.last_state.get('objective'))
            pi, v = self.run_base_policy_and_value(p, env.last_state, lar)
            action = self.draws.choose_action(pi)
            states.append(env.last_state)
            lars.append(lar)
            actions.append(action)
            values.append(v)
            prev_state = env.last_state
            new_state, reward, terminal, pc = env.process(action)
            frame = self._frame(prev_state, reward, action, terminal, pc, last_action, last_reward)
            self.exp.add_frame(frame)
            self.episode_reward += reward
            rewards.append(reward)
            self.local_t += 1
            if terminal:
                terminal_end = True
                score = self.episode_reward
                self.episode_reward = 0
                env.reset()
                self.reset_state()
                break
        R = 0.0
        if not terminal_end:
            _, R = self.run_base_policy_and_value(p, new_state, frame.get_action_reward(self.A), advance=False)
        n = len(actions)
        adv = np.zeros(n)
        Rs = np.zeros(n)
        for i in reversed(range(n)):
            R = rewards[i] + cfg["gamma"] * R
            Rs[i] = R
            adv[i] = R - values[i]
        a1h = np.zeros((n, self.A))
        a1h[np.arange(n), actions] = 1.0
        t = lambda x: torch.tensor(np.asarray(x), dtype=self.dtype)
        batch = dict(base_x=t(np.stack([s['image'] for s in states])), base_lar=t(np.stack(lars)),
                     base_a=t(a1h), base_adv=t(adv), base_R=t(Rs),
                     base_state=start_state if cfg["use_lstm"] else None)
        info = dict(n=n, actions=actions, rewards=rewards, values=values, score=score,
                    terminal_end=terminal_end)
        return batch, info

    @torch.no_grad()
    def _aux_trunk(self, p, frame):
        feat, _ = M.trunk(self._img(frame.state), self._lar(frame.get_last_action_reward(self.A)), p,
                          self.cfg["use_lstm"], None)
        return feat

    def _sample_seq(self):
        H, L = self.cfg["experience_history_size"], self.cfg["local_t_max"] + 1
        start = self.draws.seq_start(H - L - 1)
        idx = self.exp.sequence_from_start(start, L)
        return [self.exp.frames[i] for i in idx], start

    def process_pc(self, p):                         # trainer.py:339-380
        frames, start = self._sample_seq()
        rev = frames[::-1]
        pc_R = np.zeros([20, 20], dtype=np.float32)
        if not rev[1].terminal:
            _, qmax = M.pc_head(self._aux_trunk(p, rev[0]), p)
            pc_R = qmax[0].detach().numpy().astype(np.float32)
        xs, acts, Rs, lars = [], [], [], []
        for f in rev[1:]:
            pc_R = f.pixel_change + self.cfg["gamma_pc"] * pc_R
            a = np.zeros([self.A])
            a[f.action] = 1.0
            xs.append(f.state['image'])
            acts.append(a)
            Rs.append(pc_R)
            lars.append(f.get_last_action_reward(self.A))
        for l in (xs, acts, Rs, lars):
            l.reverse()
        t = lambda x: torch.tensor(np.asarray(x), dtype=self.dtype)
        return dict(pc_x=t(np.stack(xs)), pc_lar=t(np.stack(lars)), pc_a=t(np.stack(acts)),
                    pc_R=t(np.stack(Rs))), dict(start=start, n=len(frames))

    def process_vr(self, p):                         # trainer.py:383-412
        frames, start = self._sample_seq()
        rev = frames[::-1]
        vr_R = 0.0
        if not rev[1].terminal:
            _, v = M.policy_value(self._aux_trunk(p, rev[0]), p)
            vr_R = float(v[0])
        xs, Rs, lars = [], [], []
        for f in rev[1:]:
            vr_R = f.reward + self.cfg["gamma"] * vr_R
            xs.append(f.state['image'])
            Rs.append(vr_R)
            lars.append(f.get_last_action_reward(self.A))
        for l in (xs, Rs, lars):
            l.reverse()
        t = lambda x: torch.tensor(np.asarray(x), dtype=self.dtype)
        return dict(vr_x=t(np.stack(xs)), vr_lar=t(np.stack(lars)), vr_R=t(np.asarray(Rs))), \
            dict(start=start, n=len(frames))

    def process_rp(self):                            # trainer.py:415-436
        coin, pick = self.draws.rp()
        idx = self.exp.rp_from_draws(coin, pick)
        fr = [self.exp.frames[i] for i in idx]
        r = fr[3].reward
        c = [0.0, 0.0, 0.0]
        if -1e-10 < r < 1e-10:
            c[0] = 1.0
        elif r > 0:
            c[1] = 1.0
        else:
            c[2] = 1.0
        t = lambda x: torch.tensor(np.asarray(x), dtype=self.dtype)
        return dict(rp_x=t(np.stack([f.state['image'] for f in fr[:3]])), rp_c=t([c])), dict(idx=idx)

    def gather_batch(self, p):
        """Rollout + aux sampling with the shared parameters `p` (trainer.py:463-534)."""
        batch, info = self.process_base(p)
        if self.cfg.get("use_pixel_change"):
            b, i = self.process_pc(p)
            batch.update(b)
            info["pc"] = i
        if self.cfg.get("use_value_replay"):
            b, i = self.process_vr(p)
            batch.update(b)
            info["vr"] = i
        if self.cfg.get("use_reward_prediction"):
            b, i = self.process_rp()
            batch.update(b)
            info["rp"] = i
        return batch, info


LOSS_KEYS = ("total_loss", "base_loss", "policy_loss", "value_loss", "pc_loss", "vr_loss", "rp_loss")


class OracleTrainer(object):
    def __init__(self, cfg, n_actors=1, draws=None, seed=0, dtype=torch.float32, params=None, envs=None):
        self.cfg = dict(cfg)
        self.dtype = dtype
        kw = dict(use_lstm=cfg["use_lstm"], use_pixel_change=cfg.get("use_pixel_change", False),
                  use_value_replay=cfg.get("use_value_replay", False),
                  use_reward_prediction=cfg.get("use_reward_prediction", False))
        self.params = params if params is not None else M.init_params(cfg["action_size"], cfg.get("objective_size", 0), seed=seed,
                                                                      dtype=dtype, **kw)
        if draws is None:
            shared = RefDraws(np.random.RandomState(0xA3C))
            draws = [shared] * n_actors
        self.actors = [OracleActor(self.cfg, draws[i], dtype, env=(envs[i] if envs else None)) for i in range(n_actors)]
        npdt = np.float32 if dtype == torch.float32 else np.float64
        self.opt = OracleRMSProp(decay=cfg["rmsp_alpha"], momentum=0.0, epsilon=cfg["rmsp_epsilon"],
                                 clip_norm=cfg["grad_norm_clip"], dtype=npdt)
        self.opt.create_slots([v.numpy() for v in self.params.values()])
        self.initial_lr = cfg.get("initial_learning_rate",
                                  log_uniform(cfg["initial_alpha_low"], cfg["initial_alpha_high"],
                                              cfg["initial_alpha_log_rate"]))

    def anneal_lr(self, global_t):                   # trainer.py:140-144
        T = self.cfg["max_time_step"]
        return max(0.0, self.initial_lr * (T - global_t) / T)

    def is_full(self):
        return all(a.exp.is_full() for a in self.actors)

    def fill(self):
        while not self.is_full():
            for a in self.actors:
                if not a.exp.is_full():
                    a.fill_step(self.params)

    def actor_grad(self, actor, params=None):
        p = {k: v.detach().clone().requires_grad_(True) for k, v in (params or self.params).items()}
        batch, info = actor.gather_batch({k: v.detach() for k, v in p.items()})
        out = M.unreal_loss(p, batch, self.cfg)
        grads = torch.autograd.grad(out["total_loss"], list(p.values()), allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, p.values())]
        losses = {k: float(out[k].detach()) for k in LOSS_KEYS if k in out}
        losses["entropy"] = out["entropy"].detach().numpy()
        return grads, losses, info, batch

    def _apply(self, grads, lr):
        plist = [v.numpy() for v in self.params.values()]      # views: in-place update
        return float(self.opt.step(plist, [g.numpy() for g in grads], lr))

    def process_async(self, i, global_t):
        """Reference semantics for actor i (trainer.py:438-636).  Returns (diff_local_t, score)."""
        a = self.actors[i]
        if not a.exp.is_full():
            a.fill_step(self.params)
            return 0, None, None
        t0 = a.local_t
        lr = self.anneal_lr(global_t)
        grads, losses, info, _ = self.actor_grad(a)
        losses["grad_norm"] = self._apply(grads, lr)
        return a.local_t - t0, info["score"], losses

    def process_batched(self, global_t):
        """Batched-learner semantics (mean over actors, one clip, one RMSProp step)."""
        lr = self.anneal_lr(global_t)
        frozen = {k: v.detach().clone() for k, v in self.params.items()}
        tot = None
        infos, losses_all = [], []
        steps = 0
        for a in self.actors:
            t0 = a.local_t
            g, l, info, _ = self.actor_grad(a, frozen)
            steps += a.local_t - t0
            tot = g if tot is None else [x + y for x, y in zip(tot, g)]
            infos.append(info)
            losses_all.append(l)
        mean = [x / len(self.actors) for x in tot]
        norm = self._apply(mean, lr)
        return steps, infos, losses_all, mean, norm
