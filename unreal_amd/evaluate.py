"""Evaluation harness, after /root/reference/evaluate.py:92-169, 250-275: run the policy of a
(restored) network WITHOUT learning for a number of episodes and report return / success statistics.  The
reference evaluates one MINOS episode at a time with batch-1 session calls; here B maze actors roll in
lock-step on the device with the same kernels the trainer uses (policy sampled like `choose_action`, or
greedy).  The maze has no step limit (maze_environment.py:114-118), so `max_episode_steps` bounds an episode;
such episodes count as failures ("success := terminal")."""
import torch

from . import ops
from .environment.maze_environment import BatchedMazeEnvironment
from .model.model import PathWS
from .train.trainer import PhiloxDraws


class Evaluate(object):
    def __init__(self, network, batch_size=64, device="cuda:0", seed=0xE7A1, greedy=False, draws=None):
        self.net, self.B, self.greedy = network, int(batch_size), greedy
        self.device = torch.device(device)
        self.draws = draws if draws is not None else PhiloxDraws(seed)
        B, A = self.B, network._action_size
        self.env = BatchedMazeEnvironment(B, 2, self.device)
        network.bind_frame_scale(self.env.frame_scale)
        self.ws = PathWS(B, B, self.device, save_c1=False, lstm=network._use_lstm, xld=network.xld)
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=self.device)
        self.pi, self.v = z(B * A, torch.float32), z(B, torch.float32)
        self.u, self.actions = z(B, torch.float64), z(B, torch.int32)
        self.rewards, self.terminals = z(B, torch.float32), z(B, torch.int32)

    def process(self, n_episodes, max_episode_steps=2000, one_episode_per_actor=False):
        """-> dict(episodes, success_rate, mean_return, return_std, mean_length, timeouts).
        `one_episode_per_actor`: count only the FIRST episode of each of the B lock-step actors and stop when all B have
        finished or timed out (n_episodes is ignored).  Stopping at the first n finished episodes instead over-represents
        short episodes whenever actors restart while others are still in their first one."""
        B, A, net, ws, ring = self.B, self.net._action_size, self.net, self.ws, self.env.ring
        net.refresh_shadows()
        self.env.reset()
        ring.episode_reward.zero_()
        if net._use_lstm:
            ws.c0.zero_()
            ws.h0.zero_()
        steps = [0] * B
        done, returns, lengths, successes, timeouts = 0, [], [], 0, 0
        counted = [False] * B
        if one_episode_per_actor:
            n_episodes = B
        while done < n_episodes:
            net.begin_pass()
            ring.cur_idx(out=ws.frame_idx[:B])
            net.encode_rows(ring, ws, 0, B, lar_from_ring=False, save_c1=False, lstm_x=False)
            if net._use_lstm:
                net.lstm_step(ws, 0, B, fused_x=True)
            feat, ld = net.features(ws, 0)
            if not self.greedy:
                self.draws.uniform(self.u)
            net.policy_step(B, feat, ld, None if self.greedy else self.u, self.pi, self.v, self.actions)
            self.env.process(self.actions, None, self.rewards, self.terminals, reset_on_terminal=True,
                             track_score=True)
            if net._use_lstm:                      # carry the state; zero it where the episode ended
                ops.copy_(ws.c0, ws.c[:B * 256])
                ops.copy_(ws.h0, ws.h[:B * 256])
                ops.reset_state(B, self.terminals, ws.c0, ws.h0)
            term = self.terminals.cpu().numpy()
            score = ring.score_out.cpu().numpy()
            force = torch.zeros(B, dtype=torch.int32)
            ep_r = None
            for b in range(B):
                steps[b] += 1
                skip = one_episode_per_actor and counted[b]
                if term[b]:
                    if not skip:
                        returns.append(float(score[b])); lengths.append(steps[b]); successes += 1; done += 1
                    steps[b] = 0; counted[b] = True
                elif steps[b] >= max_episode_steps:
                    if ep_r is None:
                        ep_r = ring.episode_reward.cpu()
                    if not skip:
                        timeouts += 1; done += 1
                        returns.append(float(ep_r[b])); lengths.append(max_episode_steps)
                    steps[b] = 0; force[b] = 1; counted[b] = True
            if int(force.sum()):                   # abandon timed-out episodes
                m = force.to(self.device)
                self.env.reset(m)
                ring.episode_reward.mul_((1 - m).to(torch.float32))
                if net._use_lstm:
                    ops.reset_state(B, m, ws.c0, ws.h0)
        n = len(returns)
        mean = sum(returns) / n
        return dict(episodes=n, success_rate=successes / float(n), mean_return=mean,
                    return_std=(sum((r - mean) ** 2 for r in returns) / n) ** 0.5,
                    mean_length=sum(lengths) / float(n), timeouts=timeouts)
