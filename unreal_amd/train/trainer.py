"""Batched actor-learner: `Trainer.process()` for B maze actors per GPU on gfx950 kernels.

Mirrors /root/reference/train/trainer.py: ctor 31-128, prepare 132-135, stop 137-138,
_anneal_learning_rate 140-144, choose_action 147-148, set_start_time 172-173,
_fill_experience 176-205, _process_base 218-336, _process_pc 339-380, _process_vr 383-412,
_process_rp 415-436, process 438-636.

What changes, and only this: the reference runs `parallel_size` Python threads, each
rolling ONE environment with batch-1 session calls and applying its own clipped gradient to shared
RMSProp slots.  Here ONE call advances all B actors of this GPU by <= n_step_TD steps in lock-step
(an actor whose episode ends stops for the rest of the call, exactly like the reference's `break`),
samples B pixel-control / value-replay / reward-prediction batches from the device ring, and makes
one update with the MEAN over actors of the reference's per-actor gradient (sums over time), one
global-norm clip and one RMSProp step.  With B == 1 this is the reference algorithm; staleness is 0.
The rollout's activations are kept in HBM and reused by the backward pass (the reference recomputes
the same forward with the same synced weights, trainer.py:457,543-570).

`groups=G` restores update density on the batched path: the B actors of the GPU are dealt into G groups of B/G and
one process() call runs G complete actor-learner passes one after the other -- group g rolls out with the weights
group g-1 has just updated, its mean gradient is clipped and applied by itself (G RMSProp steps per call).  G = 1 is
the lock-step learner above; G = B is the reference's algorithm executed actor after actor (zero staleness, the
order in which hogwild threads would take the lock if there were one).

Per-actor random draws come from a counter RNG (Philox) on the device; `draws` can be replaced to
replay a recorded stream (parity tests).  TF-only arguments are accepted and ignored.
"""
import time

import torch

from .. import ops
from ..environment.environment import Environment
from ..environment.maze_environment import BatchedMazeEnvironment
from ..model.model import UnrealModel, PathWS, GradWS
from .experience import Experience

PERFORMANCE_LOG_INTERVAL = 2000


def log_uniform(lo, hi, rate):
    """Initial learning rate (main.py:61-65): exp((1-rate) log lo + rate log hi) = 7.0711e-4 by default."""
    import math
    return math.exp(math.log(lo) * (1 - rate) + math.log(hi) * rate)


class PhiloxDraws(object):
    """Device draws; every call consumes a fresh Philox stream id.  Every rank makes the same sequence of calls and
    reads its own actors' columns [rank*B, (rank+1)*B) of the call's global draw, so the trajectories of a job do not
    depend on how many GPUs its actors are sharded over (batch=None: a plain stream, single process)."""

    def __init__(self, seed, rank=0, batch=None, world_size=1):
        self.seed = int(seed)
        self.counter = 1
        self.batch = None if batch is None else int(batch)
        self.stride = None if batch is None else int(batch) * int(world_size)
        self.col0 = 0 if batch is None else int(rank) * int(batch)

    def _next(self):
        self.counter += 1
        return self.counter

    def uniform(self, out):
        ops.philox_uniform(self.seed, self._next(), out, self.batch, self.stride, self.col0)
        return out

    def randint(self, high, out):
        ops.philox_randint(self.seed, self._next(), high, out, self.batch, self.stride, self.col0)
        return out


class Trainer(object):
    def __init__(self, thread_index, global_network, initial_learning_rate, learning_rate_input, grad_applier,
                 env_type, env_name, use_lstm, use_pixel_change, use_value_replay, use_reward_prediction,
                 pixel_change_lambda, entropy_beta, local_t_max, n_step_TD, gamma, gamma_pc,
                 experience_history_size, max_global_time_step, device, segnet_param_dict=None,
                 image_shape=(84, 84), is_training=True, n_classes=0, random_state=None, termination_time=50.0,
                 segnet_lambda=1.0, dropout=0.0, batch_size=1, world_size=1, rank=0, seed=0xA3C, draws=None,
                 grad_sync=None, simulator=None, groups=1, overlap_host=None):
        if env_type != "maze" and simulator is None:
            raise NotImplementedError("env_type=%r needs a host simulator object (simulator=...); only 'maze' runs "
                                      "entirely on the device" % env_type)
        self.simulator = simulator
        # host-fed actors: alternate two half-batches between the host (simulators, staging) and the device.  None = on
        # from 2048 actors: measured with a cost-free simulator the halves win 1.22x at 4096 actors (654 k vs 537 k
        # env-steps/s) and lose 0.85x at 1024 (the device phase of a step is then too short to be worth twice the
        # launches); with a simulator that dominates the step the two schedules are within 5 %
        self._overlap_request = overlap_host
        # upstream replay semantics for host-fed (Lab-contract) actors: zero / non-zero reward buckets and reward
        # clipping (train/experience_lab_ver.py:14,18,76-80); this fork's buckets for the maze (train/experience.py)
        self.rp_mode = 1 if env_type == "lab" else 0      # indoor: this fork's train/experience.py, like the maze
        self.thread_index = thread_index
        self.learning_rate_input = learning_rate_input
        self.env_type, self.env_name = env_type, env_name
        self.use_lstm = use_lstm
        self.use_pixel_change = use_pixel_change
        self.use_value_replay = use_value_replay
        self.use_reward_prediction = use_reward_prediction
        self.pixel_change_lambda = pixel_change_lambda
        self.entropy_beta = entropy_beta
        self.local_t_max, self.n_step_TD = local_t_max, n_step_TD
        self.gamma, self.gamma_pc = gamma, gamma_pc
        self.experience_history_size = experience_history_size
        self.max_global_time_step = max_global_time_step
        self.action_size = Environment.get_action_size(env_type, env_name)
        self.objective_size = Environment.get_objective_size(env_type, env_name)
        self.device = torch.device(device if device not in (None, "/gpu:0", "/cpu:0") else "cuda:0")
        self.B = int(batch_size)                     # actors of this rank
        self.groups = int(groups)                    # sequential updates per process() call
        if self.groups < 1 or self.B % self.groups:
            raise ValueError("groups=%d must divide batch_size=%d" % (self.groups, self.B))
        if self.groups > 1 and env_type != "maze":
            raise NotImplementedError("groups > 1 needs an environment that steps a sub-range of its actors (maze)")
        self.Bg = self.B // self.groups              # actors per group = batch of every kernel launch
        self.world_size, self.rank = int(world_size), int(rank)
        self.grad_scale = 1.0 / float(self.Bg * self.world_size)
        self.grad_sync = grad_sync
        self.local_network = global_network          # one parameter copy per GPU (sync_from is a no-op)
        self.grad_applier = grad_applier
        self.apply_gradients = grad_applier.minimize_local(None, global_network.get_vars(),
                                                           global_network.get_vars(), thread_index)
        self.sync = self.local_network.sync_from(global_network)
        self.initial_learning_rate = initial_learning_rate
        self._own_draws = draws is None
        self.draws = draws if draws is not None else PhiloxDraws(seed, self.rank, self.B, self.world_size)
        self.local_t = 0
        self.episode_reward = 0
        self.prev_local_t = -1
        self.start_time = time.time()
        self.environment = None
        self.experience = None
        self.last_losses = {}
        self.last_grad_norm = None
        self.time_grad_sync = False                  # bench.py: time the gradient exchange of every update
        self._sync_events = []

    # ---------------------------------------------------------------------------------------------------
    def prepare(self, termination_time=50.0, termination_dist_value=-10.0):
        B, A, dev = self.B, self.action_size, self.device
        T, Ta = self.n_step_TD, self.local_t_max
        if self.env_type == "maze":
            self.environment = BatchedMazeEnvironment(B, self.experience_history_size, dev)
        else:
            from ..environment.hostfed_environment import HostFedEnvironment
            indoor = self.env_type == "indoor"
            if indoor and getattr(self.simulator, "objective_size", 0) != self.objective_size:
                raise ValueError("simulator.objective_size != Environment.get_objective_size(%r, %r) = %d"
                                 % (self.env_type, self.env_name, self.objective_size))
            self.environment = HostFedEnvironment(self.simulator, B, self.experience_history_size, dev,
                                                  action_size=A, clip_reward=not indoor,
                                                  objective_size=self.objective_size,
                                                  reward_divisor=termination_time if indoor else 1.0)
        self.overlap_host = False
        if self.env_type != "maze":
            want = self._overlap_request if self._overlap_request is not None else (B % 2 == 0 and B >= 2048)
            if want:
                self.environment.enable_parts(2)
                self.overlap_host = True
        self.full_environment = self.environment
        self.full_ring = self.ring = self.environment.ring
        self.local_network.bind_frame_scale(self.environment.frame_scale)
        # [last action | last reward] columns of the LSTM input: within 1 for the maze (rewards -1 / 0 / +1); host-fed
        # actors feed raw rewards and measurement vectors, whose maximum is reduced per pass (model.encode_rows)
        self.local_network.lar_bounded = self.env_type == "maze"
        self.experience = Experience(self.experience_history_size, ring=self.full_ring)
        B = self.Bg                                  # everything below is sized for ONE group
        lstm = self.use_lstm
        aux = self.use_pixel_change or self.use_value_replay
        xld = self.local_network.xld
        self.base_ws = PathWS(T * B, B, dev, save_c1=True, lstm=lstm, xld=xld)
        self.boot_ws = PathWS(B, B, dev, save_c1=False, lstm=lstm, xld=xld)
        self.rp_ws = PathWS(3 * B, B, dev, save_c1=True, lstm=False, xld=xld) if self.use_reward_prediction else None
        # pixel control and value replay as ONE batch of 2B replayed sequences (actor 2b = pc sample of b, 2b + 1 = vr
        # sample): see _train_aux_batched.  Decided HERE, once (the class default is read at prepare() time only).
        self.batch_aux = bool(self.batch_aux_default and self.use_pixel_change and self.use_value_replay)
        # per-branch workspaces (_train_pc / _train_vr: one of the two tasks on, or batch_aux off) are allocated on first
        # use: with the batched pass they would be ~5 GB of dead HBM per rank at 4096 actors
        self._aux_ws = None
        self._aux_ws_args = (Ta * B, B, dev, lstm, xld) if aux else None
        per_branch = aux and not self.batch_aux
        rows = max(T, Ta if per_branch else 0, 3 if self.use_reward_prediction else 0) * B
        self.gws = GradWS(rows, B, dev, lstm=lstm, pc=self.use_pixel_change and per_branch, A=A)
        if self.batch_aux:
            self.aux2_ws = PathWS(Ta * 2 * B, 2 * B, dev, save_c1=True, lstm=lstm, xld=xld)
            self.boot2_ws = PathWS(2 * B, 2 * B, dev, save_c1=False, lstm=lstm, xld=xld)
            self.gws2 = GradWS(Ta * 2 * B, 2 * B, dev, lstm=lstm, pc=True, A=A, pc_rows=Ta * B)
        f = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        i = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
        d = lambda n: torch.zeros(n, dtype=torch.float64, device=dev)
        self.full_lstm_c, self.full_lstm_h = f(self.B * 256), f(self.B * 256)
        self.pi, self.v = f(T * B * A), f(T * B)
        self.actions, self.rewards, self.terminals = i(T * B), f(T * B), i(T * B)
        self.active_log, self.active = i(T * B), i(B)
        self.n_steps, self.terminal_end = i(B), i(B)
        self.boot_v, self.R, self.adv = f(B), f(T * B), f(T * B)
        self.dlogits, self.dv = f(max(T, Ta) * B * A), f(max(T, Ta) * B)
        self.u_act = d(T * B)
        self.losses = f(8)            # policy, value, entropy, pc, vr, rp
        self.stats = d(3)
        if aux:
            L = Ta + 1
            self.seq_idx, self.seq_len, self.seq_start = i(L * B), i(B), i(B)
            self.seq_mask, self.seq_act = i(Ta * B), i(Ta * B)
            self.aux_v, self.aux_R, self.aux_boot_v = f(Ta * B), f(Ta * B), f(B)
            if self.use_pixel_change:
                self.boot_hp, self.boot_qmax = f(B * ops.F2_DIM), f(B * ops.PC_CELLS)
        if self.use_reward_prediction:
            self.rp_coin, self.rp_u, self.rp_class = i(B), d(B), i(B)
            self.rp_logits, self.rp_dlogits = f(B * 3), f(B * 3)
        if self.batch_aux:
            L = Ta + 1
            self.seq_idx_cat = i(2 * L * B)
            self.seq_idx2 = [self.seq_idx_cat[:L * B], self.seq_idx_cat[L * B:]]
            self.seq_len2, self.seq_start2 = [i(B), i(B)], [i(B), i(B)]
            self.seq_mask2 = [i(Ta * B), i(Ta * B)]
            self.last2 = i(2 * B)
            r = torch.arange(Ta * B, dtype=torch.int32)
            self.map_seq = torch.stack([r, r + L * B], dim=1).reshape(-1).contiguous().to(dev)      # [2r + s] = s*L*B + r
            b = torch.arange(B, dtype=torch.int32)
            self.map_boot = torch.stack([b, b + B], dim=1).reshape(-1).contiguous().to(dev)          # [2b + s] = s*B + b
            self.aux_dv = f(Ta * B)
        self.loss_sum = f(8)          # losses summed over the groups of one process() call
        self._fill_calls = 0
        self._full = False
        self._group_views = []
        for g in range(self.groups):
            b0, b1 = g * self.Bg, (g + 1) * self.Bg
            env = self.full_environment if self.groups == 1 else self.full_environment.view(b0, b1)
            self._group_views.append((env, self.full_lstm_c[b0 * 256:b1 * 256], self.full_lstm_h[b0 * 256:b1 * 256], b0))
        self._select_group(0)
        self._rollout_split_setup()

    @property
    def aux_ws(self):
        """Workspace of the one-branch-at-a-time replay schedule (lazy: see prepare())."""
        if self._aux_ws is None and self._aux_ws_args is not None:
            rows, B, dev, lstm, xld = self._aux_ws_args
            self._aux_ws = PathWS(rows, B, dev, save_c1=True, lstm=lstm, xld=xld)
            self.gws.ensure_rows(rows, B, dev)
        return self._aux_ws

    def _select_group(self, g):
        """Point the pipeline at the actors [g*Bg, (g+1)*Bg): environment / ring views, carried LSTM state, draws."""
        env, c, h, b0 = self._group_views[g]
        self.group = g
        self.environment, self.ring, self.lstm_c, self.lstm_h = env, env.ring, c, h
        if self._own_draws:
            self.draws.batch = self.Bg
            self.draws.col0 = self.rank * self.B + b0

    def stop(self):
        if self.environment is not None:
            self.environment.stop()

    def set_start_time(self, start_time):
        self.start_time = start_time

    def _anneal_learning_rate(self, global_time_step):
        lr = self.initial_learning_rate * (self.max_global_time_step - global_time_step) / self.max_global_time_step
        return max(lr, 0.0)

    def choose_action(self, pi_values, u):
        """Inverse-CDF draw (numpy RandomState.choice semantics) -- done on device by ops.softmax_sample."""
        raise NotImplementedError("actions are drawn on the device; see ops.softmax_sample")

    # ---------------------------------------------------------------------------------------------------
    def _policy_step(self, ws, t, u, actions_out, pi_out, v_out, prefilled=False, heads=True):
        """Forward the current observations of all actors as time row-block t of `ws` and draw actions.  `prefilled`:
        the previous environment step has already written this block's frame indices and last_action_reward columns.
        `heads` False: the caller's environment step computes pi / V / action itself (fused launch); -> (feat, ld)."""
        B, A, net = self.Bg, self.action_size, self.local_network
        if not prefilled:
            self.ring.cur_idx(out=ws.frame_idx[t * B:(t + 1) * B])
        net.encode_rows(self.ring, ws, t * B, B, lar_from_ring=False, save_c1=ws.c1 is not None,
                        lar_prefilled=prefilled and self.use_lstm, lstm_x=False)
        if self.use_lstm:
            net.lstm_step(ws, t, B, fused_x=True)
        feat, ld = net.features(ws, t * B)
        if heads:
            net.policy_step(B, feat, ld, u, pi_out, v_out, actions_out)
        return feat, ld

    FILL_SYNC_EVERY = 64
    GROUP_SYNC_EVERY = 16

    def _fill_experience(self, sess=None):
        """One policy step per call until every actor's replay is full (trainer.py:176-205)."""
        # weights are frozen while the replay fills: split them on the first call (or after an announced load)
        self.local_network.refresh_shadows(only_if_stale=self._fill_calls > 0)
        self.local_network.begin_pass()
        for g in range(self.groups):
            self._select_group(g)
            self._fill_group()
        self._select_group(0)
        self._fill_calls += 1
        # bound the queue of un-synchronised dispatches: a fill call is ~12 launches and returns at once, so a caller's
        # `while not full: process()` loop would put ~24 k dispatches in flight within a second -- rocprofv3's counter
        # thread does not survive that (profiles/r02_pmc_fault.log).  The fill is untimed; a sync every 64 calls is free.
        if self._fill_calls % self.FILL_SYNC_EVERY == 0:
            torch.cuda.current_stream().synchronize()
        if self._fill_calls >= self.experience_history_size:
            full = self.experience.is_full()
            if self.world_size > 1:            # every rank leaves the fill phase in the SAME call (the learn phase
                from .. import parallel        # issues one gradient all-reduce per call: ranks must stay paired)
                full = parallel.all_true(full, self.device)
            if full:
                self.full_environment.reset()  # trainer.py:203-205
                self._full = True

    def _fill_group(self):
        B, ws = self.Bg, self.base_ws
        if self.use_lstm:
            ops.copy_(ws.c0, self.lstm_c)
            ops.copy_(ws.h0, self.lstm_h)
        self.draws.uniform(self.u_act[:B])
        self._policy_step(ws, 0, self.u_act[:B], self.actions[:B], self.pi[:B * self.action_size], self.v[:B])
        self.environment.process(self.actions[:B], None, self.rewards[:B], self.terminals[:B],
                                 reset_on_terminal=True, track_score=False)
        if self.use_lstm:                      # state advances; NOT reset on terminal here (:201-202)
            ops.copy_(self.lstm_c, ws.c[:B * 256])
            ops.copy_(self.lstm_h, ws.h[:B * 256])

    def _rollout_steps_overlapped(self):
        """The T rollout steps for host-fed actors in two half-batches: while the host steps / stages one half (its
        simulators live on the CPU), the device ingests and forwards the other half on that half's own stream.  Same
        kernels, same rows, same draws as the lock-step loop -- only the schedule differs."""
        B, T, A, ws, net = self.Bg, self.n_step_TD, self.action_size, self.base_ws, self.local_network
        env = self.environment
        parts = env.parts
        main = torch.cuda.current_stream()
        start = torch.cuda.Event()
        start.record(main)

        def forward(k, t):          # on part k's stream: policy forward of its actors for step t + action D2H request
            p = parts[k]
            b0, n = p["b0"], p["b1"] - p["b0"]
            r0 = t * B + b0
            p["ring"].cur_idx(out=ws.frame_idx[r0:r0 + n], base_actor=b0)
            net.encode_rows(self.ring, ws, r0, n, lar_from_ring=False, save_c1=ws.c1 is not None, actor_ring=p["ring"],
                            lstm_x=False)
            if self.use_lstm:
                net.lstm_step(ws, t, B, b0, n, fused_x=True)
            feat, ld = net.features(ws, r0)
            net.policy_step(n, feat, ld, self.u_act[r0:r0 + n], self.pi[r0 * A:(r0 + n) * A], self.v[r0:r0 + n],
                            self.actions[r0:r0 + n])
            env.part_request_actions(k, self.actions[r0:r0 + n], self.active[b0:b0 + n])

        for k, p in enumerate(parts):
            with torch.cuda.stream(p["stream"]):
                p["stream"].wait_event(start)
                forward(k, 0)
        for t in range(T):
            for k, p in enumerate(parts):
                b0, n = p["b0"], p["b1"] - p["b0"]
                r0 = t * B + b0
                env.part_host_step(k, has_active=True)           # host: the other part's device work runs meanwhile
                with torch.cuda.stream(p["stream"]):
                    env.part_ingest(k, self.actions[r0:r0 + n], self.active[b0:b0 + n], self.rewards[r0:r0 + n],
                                    self.terminals[r0:r0 + n], reset_on_terminal=True, track_score=True)
                    ops.rollout_advance(n, self.terminals[r0:r0 + n], self.active[b0:b0 + n],
                                        self.active_log[r0:r0 + n], self.n_steps[b0:b0 + n], self.terminal_end[b0:b0 + n])
                    if t + 1 < T:
                        forward(k, t + 1)
        for p in parts:                                          # join: the learner continues on the caller's stream
            done = torch.cuda.Event()
            done.record(p["stream"])
            main.wait_event(done)

    # Maze actors: the T rollout steps as `rollout_parts` half-batches on their own HIP streams.  A
    # step is a chain of five dependent launches (conv encoder, fc, LSTM cell, policy + draw, environment) of which only
    # the first fills the chip; with two halves in flight one half's latency-bound launches run beside the other half's
    # encoder.  Same kernels, same rows, same draws as the lock-step loop.  0 / 1 = off.  Decided by same-process A/B
    # (tools/exp/rollout_split_ab.py, profiles/r04_ab_summary.md).
    rollout_parts_default = 0
    ROLLOUT_SPLIT_MIN_ACTORS = 2048
    # Maze rollout: policy head + softmax + action draw inside the environment step's launch (unreal_maze_policy_rollout_step,
    # bit-identical to unreal_policy_step + unreal_maze_rollout_step; one ~5-7 us launch less per step)
    fuse_policy_env = True

    def _rollout_split_setup(self):
        """Per group: the halves' environment views, streams and running absmax slot pairs (built once)."""
        n_parts = self.rollout_parts_default
        self._split = None
        if self.env_type != "maze" or n_parts < 2 or self.Bg % n_parts or self.Bg < self.ROLLOUT_SPLIT_MIN_ACTORS:
            return
        Bp = self.Bg // n_parts
        streams = [torch.cuda.Stream(device=self.device) for _ in range(n_parts)]
        self._split = []
        for env, _, _, _ in self._group_views:
            self._split.append([dict(b0=k * Bp, n=Bp, env=env.view(k * Bp, (k + 1) * Bp), stream=streams[k])
                                for k in range(n_parts)])

    def _rollout_steps_split(self):
        B, T, A, ws, net = self.Bg, self.n_step_TD, self.action_size, self.base_ws, self.local_network
        parts = self._split[self.group]
        main = torch.cuda.current_stream()
        s_f2, s_x, _ = net.ws_slots(ws)
        start = torch.cuda.Event()
        start.record(main)
        for p in parts:
            p["slots"] = (net.new_slot(), net.new_slot())
            p["stream"].wait_event(start)
        for t in range(T):
            for p in parts:
                b0, n = p["b0"], p["n"]
                r0 = t * B + b0
                with torch.cuda.stream(p["stream"]):
                    if t == 0:
                        p["env"].ring.cur_idx(out=ws.frame_idx[r0:r0 + n], base_actor=b0)
                    net.encode_rows(self.ring, ws, r0, n, lar_from_ring=False, save_c1=ws.c1 is not None,
                                    actor_ring=p["env"].ring, lar_prefilled=t > 0 and self.use_lstm, lstm_x=False,
                                    slots=p["slots"])
                    if self.use_lstm:
                        net.lstm_step(ws, t, B, b0, n, fused_x=True)
                    feat, ld = net.features(ws, r0)
                    net.policy_step(n, feat, ld, self.u_act[r0:r0 + n], self.pi[r0 * A:(r0 + n) * A], self.v[r0:r0 + n],
                                    self.actions[r0:r0 + n])
                    nxt = {}
                    if t + 1 < T:
                        nxt = dict(next_idx=ws.frame_idx[r0 + B:r0 + B + n])
                        if self.use_lstm:
                            nxt.update(next_lar=ws.xcat[(r0 + B) * ws.xld:], lar_ld=ws.xld, lar_col0=256, A=A)
                    p["env"].rollout_step(self.actions[r0:r0 + n], self.rewards[r0:r0 + n], self.terminals[r0:r0 + n],
                                          self.active[b0:b0 + n], self.active_log[r0:r0 + n], self.n_steps[b0:b0 + n],
                                          self.terminal_end[b0:b0 + n], index_parent=True, **nxt)
        for p in parts:                                          # join: the learner continues on the caller's stream
            done = torch.cuda.Event()
            done.record(p["stream"])
            main.wait_event(done)
        for p in parts:                                          # the halves' running maxima cover the workspace's rows
            ops.absmax(1, 1, p["slots"][0], 1, s_f2)
            ops.absmax(1, 1, p["slots"][1], 1, s_x)

    def _rollout(self):
        """[Base A3C] n_step_TD lock-step steps, bootstrap value, n-step returns (trainer.py:218-336)."""
        B, T, A, ws, net = self.Bg, self.n_step_TD, self.action_size, self.base_ws, self.local_network
        if self.use_lstm:
            ops.copy_(ws.c0, self.lstm_c)           # start_lstm_state
            ops.copy_(ws.h0, self.lstm_h)
        self.active.fill_(1)
        self.n_steps.zero_()
        self.terminal_end.zero_()
        self.draws.uniform(self.u_act)
        if self.overlap_host:
            self._rollout_steps_overlapped()
        split = getattr(self, "_split", None) is not None
        if split:
            self._rollout_steps_split()
        fused = self.env_type == "maze"      # the maze step kernel also does the loop bookkeeping and prepares step t+1
        fuse_policy = fused and self.fuse_policy_env     # policy head + draw inside the environment step's launch
        for t in range(0 if not (self.overlap_host or split) else T, T):
            s = slice(t * B, (t + 1) * B)
            feat, ld = self._policy_step(ws, t, self.u_act[s], self.actions[s], self.pi[t * B * A:(t + 1) * B * A],
                                         self.v[s], prefilled=fused and t > 0, heads=not fuse_policy)
            if fused:
                nxt = {}
                if t + 1 < T:
                    nxt = dict(next_idx=ws.frame_idx[(t + 1) * B:(t + 2) * B])
                    if self.use_lstm:
                        nxt.update(next_lar=ws.xcat[(t + 1) * B * ws.xld:], lar_ld=ws.xld, lar_col0=256, A=A)
                if fuse_policy:
                    self.environment.policy_rollout_step(net, feat, ld, self.u_act[s], self.pi[t * B * A:(t + 1) * B * A],
                                                         self.v[s], self.actions[s], self.rewards[s], self.terminals[s],
                                                         self.active, self.active_log[s], self.n_steps, self.terminal_end,
                                                         **nxt)
                    continue
                self.environment.rollout_step(self.actions[s], self.rewards[s], self.terminals[s], self.active,
                                              self.active_log[s], self.n_steps, self.terminal_end, **nxt)
                continue
            self.environment.process(self.actions[s], self.active, self.rewards[s], self.terminals[s],
                                     reset_on_terminal=True, track_score=True)
            ops.rollout_advance(B, self.terminals[s], self.active, self.active_log[s], self.n_steps,
                                self.terminal_end)
        if self.use_lstm:
            ops.copy_(self.lstm_c, ws.c[(T - 1) * B * 256:T * B * 256])
            ops.copy_(self.lstm_h, ws.h[(T - 1) * B * 256:T * B * 256])
        # bootstrap R = V(s_T) for actors still running; LSTM state NOT advanced (model.py:687-704)
        bw = self.boot_ws
        if self.use_lstm:
            ops.copy_(bw.c0, self.lstm_c)
            ops.copy_(bw.h0, self.lstm_h)
        self.ring.cur_idx(out=bw.frame_idx[:B])
        feat, ld = net.trunk_forward(self.ring, bw, 1, B, lar_from_ring=False, save_c1=False,
                                     clip_lar=self.rp_mode == 1,   # frame.get_action_reward(): stored (clipped) reward
                                     objective_slot_offset=-1)     # ... and the objective of frame.state (:300)
        net.value_forward(B, feat, ld, self.boot_v)
        if self.use_lstm:                      # episode ended -> reset_state() (trainer.py:293)
            ops.reset_state(B, self.terminal_end, self.lstm_c, self.lstm_h)
        ops.base_returns(B, T, self.rewards, self.v, self.n_steps, self.boot_v, self.terminal_end, self.gamma,
                         self.R, self.adv)

    def _train_base(self):
        B, T, A, ws, net, g, p = self.Bg, self.n_step_TD, self.action_size, self.base_ws, self.local_network, \
            self.local_network.g, self.local_network.p
        rows = T * B
        ops.base_loss_grad(rows, A, self.pi, A, self.v, self.actions, self.adv, self.R, self.active_log,
                           self.entropy_beta, self.grad_scale, self.dlogits, self.dv, self.losses[0:3])
        feat, ld = net.features(ws)
        d_feat = self.gws.d_feat
        ops.linear_small_bwd(rows, 256, A, feat, ld, self.dlogits, A, p["W_base_fc_p"], d_feat, 256, False,
                             g["W_base_fc_p"], g["b_base_fc_p"])
        ops.linear_small_bwd(rows, 256, 1, feat, ld, self.dv, 1, p["W_base_fc_v"], d_feat, 256, True,
                             g["W_base_fc_v"], g["b_base_fc_v"])
        net.trunk_backward(self.ring, ws, self.gws, T, B, d_feat, h0_nonzero=True)

    def _sample_sequence(self):
        """experience.sample_sequence(local_t_max+1) for every actor + the bootstrap frame's features."""
        B, Ta, net = self.Bg, self.local_t_max, self.local_network
        L = Ta + 1
        self.draws.randint(self.experience_history_size - L - 1, self.seq_start)
        ops.replay_sample_seq(self.ring, L, self.seq_start, self.seq_idx, self.seq_len)
        bw = self.boot_ws
        ops.seq_last_idx(B, self.seq_idx, self.seq_len, bw.frame_idx[:B])
        if self.use_lstm:
            bw.c0.zero_()                      # aux networks always start from the zero state (model.py:395,461)
            bw.h0.zero_()
        feat, ld = net.trunk_forward(self.ring, bw, 1, B, lar_from_ring=True, save_c1=False)
        return feat, ld

    def _aux_forward(self):
        B, Ta, net, ws = self.Bg, self.local_t_max, self.local_network, self.aux_ws
        rows = Ta * B
        ops.copy_(ws.frame_idx[:rows], self.seq_idx[:rows])
        if self.use_lstm:
            ws.c0.zero_()
            ws.h0.zero_()
        ops.seq_mask(B, Ta, self.seq_len, self.seq_mask)
        return net.trunk_forward(self.ring, ws, Ta, B, lar_from_ring=True, save_c1=True)

    # Pixel-control head, training pass: loss + backward of the two deconvolutions in ONE launch (unreal_pc_deconv_train:
    # d_dec stays on chip, hp is staged once; 22 KB of HBM traffic per frame instead of 49).  False: the two-launch form.
    fuse_pc_deconv = True
    keep_d_dec = False       # True: the one-launch form also writes d_dec to gws.d_dec (the full-size tests read it)

    def _pc_deconv_loss_backward(self, gws, rows, mask, d_hp, s_hp):
        """gws.hp (relu(pc_fc1), max in slot s_hp), self.seq_act, gws.pc_R -> losses[3], d_hp, the deconv parameter
        gradients; returns the slot holding max |d_hp| (model.py:411-443, 542-557 and their gradients)."""
        A, net = self.action_size, self.local_network
        p, g = net.p, net.g
        s_dhp = net.new_slot()             # max |d_hp|: committed by the deconv backward, read by the pc_fc1 dgrad
        if self.fuse_pc_deconv:
            ops.pc_deconv_train(rows, A, gws.hp, p["W_pc_deconv_v"], p["b_pc_deconv_v"], p["W_pc_deconv_a"], p["b_pc_deconv_a"],
                                self.seq_act, gws.pc_R, mask, self.pixel_change_lambda, self.grad_scale, self.losses[3:4], d_hp,
                                g["W_pc_deconv_v"], g["b_pc_deconv_v"], g["W_pc_deconv_a"], g["b_pc_deconv_a"], dhp_max=s_dhp,
                                hp_max=s_hp, d_dec=gws.ensure_d_dec(rows, A) if self.keep_d_dec else None)
            return s_dhp
        s_dd = net.new_slot()              # bound of max |d_dec|: committed by the deconv forward
        d_dec = gws.ensure_d_dec(rows, A)
        ops.pc_deconv_fwd(rows, A, gws.hp, p["W_pc_deconv_v"], p["b_pc_deconv_v"], p["W_pc_deconv_a"],
                          p["b_pc_deconv_a"], action=self.seq_act, target=gws.pc_R, mask=mask,
                          lam=self.pixel_change_lambda, grad_scale=self.grad_scale, d_dec=d_dec,
                          loss=self.losses[3:4], hp_max=s_hp, ddec_max=s_dd)
        ops.pc_deconv_bwd(rows, A, gws.hp, d_dec, p["W_pc_deconv_v"], p["W_pc_deconv_a"], d_hp,
                          g["W_pc_deconv_v"], g["b_pc_deconv_v"], g["W_pc_deconv_a"], g["b_pc_deconv_a"], dhp_max=s_dhp,
                          hp_max=s_hp, ddec_max=s_dd)
        return s_dhp

    def _train_pc(self):
        """[Pixel change] (trainer.py:339-380, model.py:411-443, 542-557)."""
        B, Ta, A, net = self.Bg, self.local_t_max, self.action_size, self.local_network
        p, g, gws = net.p, net.g, self.gws
        rows = Ta * B
        ws = self.aux_ws                              # (allocates the per-branch workspace on first use)
        gws.ensure_pc(rows, A)
        feat, ld = self._sample_sequence()
        s_bhp = net.new_slot()             # max of the bootstrap frames' hp: committed by the pc_fc1 GEMM
        net.pc_head_forward(B, feat, ld, self.boot_hp, ws=self.boot_ws, hp_max=s_bhp)
        ops.pc_deconv_fwd(B, A, self.boot_hp, p["W_pc_deconv_v"], p["b_pc_deconv_v"], p["W_pc_deconv_a"],
                          p["b_pc_deconv_a"], qmax=self.boot_qmax, hp_max=s_bhp)
        ops.pc_returns(self.ring, Ta + 1, self.seq_idx, self.seq_len, self.boot_qmax, self.gamma_pc, gws.pc_R)
        feat, ld = self._aux_forward()
        s_hp = net.new_slot()              # max hp: committed by the pc_fc1 GEMM
        net.pc_head_forward(rows, feat, ld, gws.hp, ws=self.aux_ws, hp_max=s_hp)
        ops.gather_i32(self.ring.r_action, self.aux_ws.frame_idx[:rows], self.seq_act)
        d_hp = gws.d_f2
        s_dhp = self._pc_deconv_loss_backward(gws, rows, self.seq_mask, d_hp, s_hp)
        from ..model.model import _splitk
        ops.gemm_split_tn(256, 2592, rows, feat, ld, d_hp, 2592, g["W_pc_fc1"], 2592,
                              splitk=_splitk(256, 2592, rows), colsum=g["b_pc_fc1"],
                              a_max=net._one if self.use_lstm else self.aux_ws.s_x, b_max=s_dhp)
        ops.gemm_split_nt(rows, 256, 2592, d_hp, 2592, net.shadow["pc_fc1_dgrad"], gws.d_feat, 256, a_max=s_dhp)
        net.trunk_backward(self.ring, self.aux_ws, gws, Ta, B, gws.d_feat)

    def _train_vr(self):
        """[Value replay] (trainer.py:383-412, model.py:446-470, 559-566)."""
        B, Ta, net = self.Bg, self.local_t_max, self.local_network
        p, g, gws = net.p, net.g, self.gws
        rows = Ta * B
        ws = self.aux_ws                              # (allocates the per-branch workspace on first use)
        feat, ld = self._sample_sequence()
        net.value_forward(B, feat, ld, self.aux_boot_v)
        ops.vr_returns(self.ring, Ta + 1, self.seq_idx, self.seq_len, self.aux_boot_v, self.gamma, self.aux_R)
        feat, ld = self._aux_forward()
        net.value_forward(rows, feat, ld, self.aux_v)
        ops.vr_loss_grad(rows, self.aux_v, self.aux_R, self.seq_mask, self.grad_scale, self.dv, self.losses[4:5])
        ops.linear_small_bwd(rows, 256, 1, feat, ld, self.dv, 1, p["W_base_fc_v"], gws.d_feat, 256, False,
                             g["W_base_fc_v"], g["b_base_fc_v"])
        net.trunk_backward(self.ring, self.aux_ws, gws, Ta, B, gws.d_feat)

    batch_aux_default = True       # class default, read by prepare() only (A/B tools build a second Trainer with it off)

    def _train_aux_batched(self):
        """[Pixel change] + [Value replay] (trainer.py:339-412) as ONE trunk pass over 2B replayed sequences: sequence
        2b is actor b's pixel-control sample, 2b + 1 its value-replay sample, so a branch's rows are every other row of the
        workspace (leading dimension doubled) and every trunk kernel -- encoder, fc, the T recurrent steps, their
        backward -- runs once at twice the rows instead of twice.  Draws, sampling, targets, heads and losses are the
        per-branch ones, in the reference's order (pc, vr)."""
        B, Ta, A, net = self.Bg, self.local_t_max, self.action_size, self.local_network
        p, g, gws, sh = net.p, net.g, self.gws2, net.shadow
        L, rows = Ta + 1, Ta * B
        for s in (0, 1):                                         # experience.sample_sequence(local_t_max + 1), pc then vr
            self.draws.randint(self.experience_history_size - L - 1, self.seq_start2[s])
            ops.replay_sample_seq(self.ring, L, self.seq_start2[s], self.seq_idx2[s], self.seq_len2[s])
            ops.seq_last_idx(B, self.seq_idx2[s], self.seq_len2[s], self.last2[s * B:(s + 1) * B])
            ops.seq_mask(B, Ta, self.seq_len2[s], self.seq_mask2[s])
        # bootstrap frames of both samples (zero LSTM state, model.py:395,461)
        bw = self.boot2_ws
        ops.gather_i32(self.last2, self.map_boot, bw.frame_idx[:2 * B])
        # (boot2_ws / aux2_ws belong to this pass alone and nothing ever writes their c0 / h0: they are the zero state they
        # were allocated as -- model.py:395,461 -- so the four fills per pass the per-branch schedule needs are not issued)
        feat, ld = net.trunk_forward(self.ring, bw, 1, 2 * B, lar_from_ring=True, save_c1=False)
        s_bhp = net.new_slot()             # max of the bootstrap frames' hp: committed by the pc_fc1 GEMM
        net.pc_head_forward(B, feat, 2 * ld, self.boot_hp, ws=self.boot2_ws, hp_max=s_bhp)
        ops.pc_deconv_fwd(B, A, self.boot_hp, p["W_pc_deconv_v"], p["b_pc_deconv_v"], p["W_pc_deconv_a"],
                          p["b_pc_deconv_a"], qmax=self.boot_qmax, hp_max=s_bhp)
        ops.pc_returns(self.ring, L, self.seq_idx2[0], self.seq_len2[0], self.boot_qmax, self.gamma_pc, gws.pc_R)
        net.value_forward(B, feat[ld:], 2 * ld, self.aux_boot_v)
        ops.vr_returns(self.ring, L, self.seq_idx2[1], self.seq_len2[1], self.aux_boot_v, self.gamma, self.aux_R)
        # the 2B sequences through the trunk
        ws = self.aux2_ws
        ops.gather_i32(self.seq_idx_cat, self.map_seq, ws.frame_idx[:2 * rows])
        feat, ld = net.trunk_forward(self.ring, ws, Ta, 2 * B, lar_from_ring=True, save_c1=True)
        d_feat = gws.d_feat
        # pixel-control head on the even rows
        s_hp = net.new_slot()              # max hp: committed by the pc_fc1 GEMM
        net.pc_head_forward(rows, feat, 2 * ld, gws.hp, ws=self.aux2_ws, hp_max=s_hp)
        ops.gather_i32(self.ring.r_action, self.seq_idx2[0][:rows], self.seq_act)
        d_hp = gws.d_hp
        s_dhp = self._pc_deconv_loss_backward(gws, rows, self.seq_mask2[0], d_hp, s_hp)
        from ..model.model import _splitk
        ops.gemm_split_tn(256, 2592, rows, feat, 2 * ld, d_hp, 2592, g["W_pc_fc1"], 2592,
                          splitk=_splitk(256, 2592, rows), colsum=g["b_pc_fc1"],
                          a_max=net._one if self.use_lstm else self.aux2_ws.s_x, b_max=s_dhp)
        ops.gemm_split_nt(rows, 256, 2592, d_hp, 2592, sh["pc_fc1_dgrad"], d_feat, 2 * 256, a_max=s_dhp)
        # value head on the odd rows
        net.value_forward(rows, feat[ld:], 2 * ld, self.aux_v)
        ops.vr_loss_grad(rows, self.aux_v, self.aux_R, self.seq_mask2[1], self.grad_scale, self.aux_dv, self.losses[4:5])
        ops.linear_small_bwd(rows, 256, 1, feat[ld:], 2 * ld, self.aux_dv, 1, p["W_base_fc_v"], d_feat[256:], 2 * 256, False,
                             g["W_base_fc_v"], g["b_base_fc_v"])
        net.trunk_backward(self.ring, ws, gws, Ta, 2 * B, d_feat)

    def _train_rp(self):
        """[Reward prediction] (trainer.py:415-436, model.py:473-488, 569-576)."""
        B, net, ws = self.Bg, self.local_network, self.rp_ws
        p, g, gws = net.p, net.g, self.gws
        self.draws.randint(2, self.rp_coin)
        self.draws.uniform(self.rp_u)
        ops.replay_sample_rp(self.ring, self.rp_coin, self.rp_u, ws.frame_idx[:3 * B], self.rp_class, self.rp_mode)
        s_c1 = net.new_slot()              # max of the conv1 activation: the c1 scale of the conv backward below
        ops.encoder_fwd(self.ring.frames, ws.frame_idx[:3 * B], net.frame_scale, p["W_base_conv1"],
                        p["b_base_conv1"], p["W_base_conv2"], p["b_base_conv2"], ws.f2, ws.c1, c1_max=s_c1,
                        prepared=net.enc_prepared)
        ops.linear_small_fwd(B, 7776, 3, ws.f2, 7776, p["W_rp_fc1"], p["b_rp_fc1"], self.rp_logits, 3)
        ops.rp_loss_grad(B, self.rp_logits, self.rp_class, self.grad_scale, None, self.rp_dlogits,
                         self.losses[5:6])
        ops.linear_small_bwd(B, 7776, 3, ws.f2, 7776, self.rp_dlogits, 3, p["W_rp_fc1"], gws.d_f2, 7776, False,
                             g["W_rp_fc1"], g["b_rp_fc1"])
        ops.relu_mask(3 * B, 2592, gws.d_f2, 2592, ws.f2, 2592)
        ops.encoder_bwd(self.ring.frames, ws.frame_idx[:3 * B], net.frame_scale, p["W_base_conv2"], ws.c1,
                        gws.d_f2, g["W_base_conv1"], g["b_base_conv1"], g["W_base_conv2"], g["b_base_conv2"],
                        c1_max=s_c1)           # (d_f2's maximum: reduced by the wrapper, 127 MB at 4096 actors)

    # ---------------------------------------------------------------------------------------------------
    def compute_gradients(self):
        """Rollout + the four loss branches; leaves the local mean gradient in local_network.grads.flat."""
        net = self.local_network
        net.refresh_shadows()                  # fp16x2 weight planes follow the last optimiser step / load
        net.begin_pass()                       # absmax slots of this pass (fp16x2 GEMM scales)
        self._rollout()
        net.grads.flat.zero_()
        self.losses.zero_()
        self._train_base()
        if self.batch_aux:
            self._train_aux_batched()
        else:
            if self.use_pixel_change:
                self._train_pc()
            if self.use_value_replay:
                self._train_vr()
        if self.use_reward_prediction:
            self._train_rp()

    def process(self, sess=None, global_t=0, summary_writer=None, summary_op_dict=None, score_input=None,
                sr_input=None, eval_input=None, entropy_input=None, term_global_t=None, losses_input=None,
                sync_stats=True):
        """-> (env steps taken by this rank's actors inside the call, mean score of episodes finished or None)."""
        if self.environment is None:
            self.prepare()
        if not self._full:
            self._fill_experience(sess)
            return 0, None
        net = self.local_network
        G = self.groups
        if G > 1:
            self.loss_sum.zero_()
        for g in range(G):                               # G complete actor-learner passes, one after the other
            self._select_group(g)
            # the reference reads global_t when a worker's process() starts (main.py:114-125)
            lr = self._anneal_learning_rate(global_t + g * self.Bg * self.n_step_TD * self.world_size)
            self.compute_gradients()
            if self.grad_sync is not None:
                if self.time_grad_sync:                  # HIP events on the launch stream around the exchange
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self.grad_sync(net.grads.flat)           # RCCL all-reduce (sum of per-rank means / world)
                if self.time_grad_sync:
                    e1.record()
                    self._sync_events.append((e0, e1))
            self.last_grad_norm = self.grad_applier.step(net.params.flat, net.grads.flat, lr)
            net.mark_params_changed()
            ops.rollout_stats(self.Bg, self.n_steps, self.ring.score_valid, self.ring.score_out, self.stats)
            if G > 1:
                ops.axpy(1.0 / G, self.losses, self.loss_sum)
                # bound the un-synchronised dispatch queue like the replay fill does (a group pass is ~300 launches: at
                # G = 64 one call would enqueue ~19 k dispatches before its only host sync; rocprofv3's counter thread
                # did not survive ~24 k in the fill, profiles/r02_pmc_fault.log).  One sync per 16 groups: < 0.1 % of a call.
                if (g + 1) % self.GROUP_SYNC_EVERY == 0 and g + 1 < G:
                    torch.cuda.current_stream().synchronize()
        if G > 1:
            self._select_group(0)
        if not sync_stats:                               # stats keep accumulating on the device
            return None, None
        steps, episodes, score_sum = self.read_stats()
        self._publish_losses()
        return steps, (score_sum / episodes if episodes > 0 else None)

    def grad_sync_ms(self):
        """[ms] of every gradient exchange since the last call (time_grad_sync): from the point the launch stream reaches
        the all-reduce (the backward kernels before it have drained) to the point it may continue with clip + RMSProp."""
        torch.cuda.synchronize()
        out = [e0.elapsed_time(e1) for e0, e1 in self._sync_events]
        self._sync_events = []
        return out

    def read_stats(self):
        """(env steps, finished episodes, sum of their scores) since the last read; one host sync."""
        st = self.stats.cpu().numpy()
        self.stats.zero_()
        steps, episodes, score_sum = int(st[0]), int(st[1]), float(st[2])
        self.local_t += steps
        return steps, episodes, score_sum

    def _publish_losses(self):
        l = (self.loss_sum if self.groups > 1 else self.losses).cpu().numpy()    # groups: mean over the call's updates
        net = self.local_network
        net.policy_loss, net.value_loss, net.entropy = float(l[0]), float(l[1]), float(l[2])
        net.base_loss = net.policy_loss + net.value_loss
        net.pc_loss, net.vr_loss, net.rp_loss = float(l[3]), float(l[4]), float(l[5])
        net.total_loss = net.base_loss + net.pc_loss + net.vr_loss + net.rp_loss
        self.last_losses = dict(total_loss=net.total_loss, base_loss=net.base_loss, policy_loss=net.policy_loss,
                                value_loss=net.value_loss, entropy=net.entropy, pc_loss=net.pc_loss,
                                vr_loss=net.vr_loss, rp_loss=net.rp_loss,
                                grad_norm=float(self.last_grad_norm.cpu()[0]))
        return self.last_losses

    def _print_log(self, global_t):
        if self.thread_index == 0 and self.local_t - self.prev_local_t >= PERFORMANCE_LOG_INTERVAL:
            self.prev_local_t += PERFORMANCE_LOG_INTERVAL
            el = time.time() - self.start_time
            print("### Performance : {} STEPS in {:.0f} sec. {:.0f} STEPS/sec. {:.2f}M STEPS/hour".format(
                global_t, el, global_t / el, global_t / el * 3600 / 1e6))
