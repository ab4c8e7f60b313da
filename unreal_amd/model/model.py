"""UnrealModel on MI355X: the reference's network surface over hand-written gfx950 kernels.

Mirrors /root/reference/model/model.py (vanilla encoder, segnet_mode == 0):
  ctor 49-66, prepare_loss 579-598, run_base_policy_and_value 630-660, run_base_value 687-704,
  run_pc_q_max 707-712, run_vr_value 715-720, run_rp_c 723-728, get_vars 731, sync_from 737-749,
  reset_state 625-628; initialisers 31-42, 752-783.
Parameters live in ONE flat fp32 device buffer (TF layouts, TF creation order, each variable
padded to a 64-float boundary so 16 B vector loads stay aligned); `get_vars()` returns views.
The batched entry points (`trunk_forward`, `trunk_backward`, ...) are what `Trainer` drives; the
batch-1 `run_*` methods keep the reference call shapes for evaluate/display-style callers.

There is no CPU fallback: every method launches kernels of libunreal_hip.so.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from .. import ops

ALIGN = 64
XLD = 264          # row stride of the [fc(256) | last_action_reward | pad] LSTM input buffer (no objective vector)


def xcat_ld(action_size, objective_size=0):
    """Row stride of the [fc(256) | one-hot last action | last reward | objective | pad] buffer (multiple of 8)."""
    return max(XLD, (256 + action_size + 1 + objective_size + 7) // 8 * 8)


def _small_chunks(n):
    """n columns as widths the small-N backward kernel is instantiated for."""
    out = []
    while n > 0:
        w = n if n in (1, 3, 4, 5, 6, 7) else (1 if n == 2 else (7 if n >= 10 else n - 3 if n - 3 in (3, 4, 5, 6) else 4))
        out.append(w)
        n -= w
    return out


def param_spec(action_size, objective_size=0, use_lstm=True, use_pixel_change=True,
               use_value_replay=True, use_reward_prediction=True):
    """(name, shape, fan_in) in the reference's variable-creation order (model.py:106-136)."""
    A = action_size
    lstm_in = 256 + A + 1 + objective_size
    spec = [("W_base_conv1", (8, 8, 3, 16), 192), ("b_base_conv1", (16,), 192),
            ("W_base_conv2", (4, 4, 16, 32), 256), ("b_base_conv2", (32,), 256),
            ("W_base_fc1", (2592, 256), 2592), ("b_base_fc1", (256,), 2592)]
    if use_lstm:
        spec += [("lstm_kernel", (lstm_in + 256, 1024), None), ("lstm_bias", (1024,), 0)]
    spec += [("W_base_fc_p", (256, A), 256), ("b_base_fc_p", (A,), 256),
             ("W_base_fc_v", (256, 1), 256), ("b_base_fc_v", (1,), 256)]
    if use_pixel_change:
        spec += [("W_pc_fc1", (256, 2592), 256), ("b_pc_fc1", (2592,), 256),
                 ("W_pc_deconv_v", (4, 4, 1, 32), 512), ("b_pc_deconv_v", (1,), 512),
                 ("W_pc_deconv_a", (4, 4, A, 32), 512), ("b_pc_deconv_a", (A,), 512)]
    if use_reward_prediction:
        spec += [("W_rp_fc1", (7776, 3), 7776), ("b_rp_fc1", (3,), 7776)]
    return spec


class FlatParams(object):
    """One flat buffer + named views; the same layout is used for grads and RMSProp slots."""

    def __init__(self, spec, device):
        self.spec = spec
        self.offsets = OrderedDict()
        off = 0
        for name, shape, _ in spec:
            n = int(np.prod(shape))
            self.offsets[name] = (off, n, shape)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.size = off
        self.n_params = sum(n for _, n, _ in self.offsets.values())
        self.flat = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.views = self.make_views(self.flat)

    def make_views(self, flat):
        return OrderedDict((k, flat[o:o + n]) for k, (o, n, _) in self.offsets.items())

    def shaped(self, name):
        o, n, shape = self.offsets[name]
        return self.flat[o:o + n].view(shape)


class PathWS(object):
    """Activations of one trunk pass over `rows` frames (time-major rows t*B + b)."""

    def __init__(self, rows, B, device, save_c1=True, lstm=True, xld=XLD):
        f = lambda n: torch.empty(n, dtype=torch.float32, device=device)
        self.rows, self.B, self.xld = rows, B, xld
        self.fc_partials = {}     # stream handle -> K-slab partial products of the fc product at few rows (encode_rows)
        self.s_x = self.s_f2 = self.s_x_cur = self.s_c1 = None     # absmax slots of the rows' LSTM input x / conv output (UnrealModel.encode_rows, per pass)
        self.pass_id = None
        self.frame_idx = torch.zeros(rows, dtype=torch.int32, device=device)
        self.c1 = f(rows * ops.C1_DIM) if save_c1 else None
        self.f2 = f(rows * ops.F2_DIM)
        # ReLU pattern of f2, 1 bit per element (written by the encoder, read by the fc dgrad's epilogue instead of f2)
        self.f2_bits = torch.zeros(rows * ops.RELU_WORDS, dtype=torch.int16, device=device) if save_c1 else None
        self.xcat = torch.zeros(rows * xld, dtype=torch.float32, device=device)
        if lstm:
            self.gates = f(rows * 1024)
            self.c = f(rows * 256)
            self.h = f(rows * 256)
            self.c0 = torch.zeros(B * 256, dtype=torch.float32, device=device)
            self.h0 = torch.zeros(B * 256, dtype=torch.float32, device=device)


class GradWS(object):
    """Gradient temporaries shared by all paths (sized for the largest)."""

    def __init__(self, rows, B, device, lstm=True, pc=True, A=4, pc_rows=None):
        f = lambda n: torch.empty(n, dtype=torch.float32, device=device)
        pc_rows = rows if pc_rows is None else pc_rows
        self.d_feat = f(rows * 256)
        self.d_fc = f(rows * 256)
        self.d_f2 = f(rows * ops.F2_DIM)
        if lstm:
            self.d_gates = f(rows * 1024)
            self.dh_rec = f(B * 256)
            self.dc = f(B * 256)
        self.rows, self.lstm, self.device = rows, lstm, device
        self.hp = self.d_hp = self.d_dec = self.pc_R = None
        if pc:
            self.ensure_pc(pc_rows, A, own_d_hp=pc_rows != rows)

    def ensure_pc(self, pc_rows, A, own_d_hp=False):
        """Pixel-control temporaries for `pc_rows` rows (allocated once; callers without their own d_hp use d_f2)."""
        if self.hp is not None and self.hp.numel() >= pc_rows * ops.F2_DIM:
            return
        f = lambda n: torch.empty(n, dtype=torch.float32, device=self.device)
        self.hp = f(pc_rows * ops.F2_DIM)
        self.d_hp = f(pc_rows * ops.F2_DIM) if own_d_hp else None
        self.d_dec = None            # only the two-launch form of the head's training pass needs it (ensure_d_dec)
        self.pc_R = f(pc_rows * ops.PC_CELLS)

    def ensure_d_dec(self, pc_rows, A):
        """d(loss)/d(deconv pre-activations) [pc_rows][400][1+A]: the hand-over of unreal_pc_deconv_fwd to unreal_pc_deconv_bwd
        (655 MB at 81,920 rows; unreal_pc_deconv_train keeps it on chip)."""
        n = pc_rows * ops.PC_CELLS * (1 + A)
        if self.d_dec is None or self.d_dec.numel() < n:
            self.d_dec = torch.empty(n, dtype=torch.float32, device=self.device)
        return self.d_dec

    def ensure_rows(self, rows, B, device):
        """Grow the row-sized temporaries (a per-branch replay pass on a trainer that was sized for the batched one)."""
        if rows <= self.rows:
            return
        f = lambda n: torch.empty(n, dtype=torch.float32, device=device)
        self.rows = rows
        self.d_feat, self.d_fc, self.d_f2 = f(rows * 256), f(rows * 256), f(rows * ops.F2_DIM)
        if self.lstm:
            self.d_gates = f(rows * 1024)


def _splitk(M, N, K):
    """K slabs for the wgrad GEMMs: ~1024 workgroups in total and a MULTIPLE OF 8 -- the kernel deals whole slabs to the
    8 XCDs, so 25 slabs put 4 on one XCD and 3 on the others (+17 % time on the 2592x256x81920 shape)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    nk = (K + 31) // 32
    if nk < 64:
        return max(1, min(nk // 4 if nk >= 8 else 1, (1024 + tiles - 1) // tiles))
    return max(8, min(8 * int(round(128.0 / tiles)), nk // 4 // 8 * 8))


class UnrealModel(object):
    """UNREAL network (reference ctor signature kept; TF-only arguments are accepted and ignored)."""

    def __init__(self, action_size, objective_size, thread_index, use_lstm, use_pixel_change,
                 use_value_replay, use_reward_prediction, pixel_change_lambda, entropy_beta, device,
                 segnet_param_dict=None, image_shape=(84, 84), is_training=True, n_classes=0,
                 segnet_lambda=1.0, dropout=0.0, for_display=False, frame_scale=None, seed=0):
        if objective_size < 0:
            raise ValueError("objective_size must be >= 0")
        if segnet_param_dict is not None and segnet_param_dict.get("segnet_mode", 0) not in (0, None):
            raise NotImplementedError("only the vanilla encoder (segnet_mode == 0) is on the hot path")
        if tuple(image_shape) != (84, 84):
            raise ValueError("the conv encoder kernels are specialised for 84x84x3 frames")
        self._device = torch.device(device if device not in (None, "/gpu:0", "/cpu:0") else "cuda:0")
        self._action_size = action_size
        self._objective_size = objective_size
        self._thread_index = thread_index
        self._use_lstm = use_lstm
        self.K_x = 256 + action_size + 1 + objective_size      # LSTM input width (model.py:343)
        self.xld = xcat_ld(action_size, objective_size)
        self._use_pixel_change = use_pixel_change
        self._use_value_replay = use_value_replay
        self._use_reward_prediction = use_reward_prediction
        self._pixel_change_lambda = pixel_change_lambda
        self._entropy_beta = entropy_beta
        # value of one stored frame byte: 1 for the maze's 0/1 bytes, 1/255 for host-fed uint8 observations.  None =
        # "whatever the environment stores": Trainer.prepare() / Evaluate set it from environment.frame_scale.
        self._frame_scale_given = frame_scale is not None
        self.frame_scale = 1.0 if frame_scale is None else float(frame_scale)
        self.spec = param_spec(action_size, objective_size, use_lstm, use_pixel_change, use_value_replay,
                               use_reward_prediction)
        self.params = FlatParams(self.spec, self._device)
        self.grads = FlatParams(self.spec, self._device)
        self.p = self.params.views
        self.g = self.grads.views
        self._init_weights(seed)
        self.variables = [self.params.shaped(n) for n, _, _ in self.spec]
        self.reset_state()
        # loss attributes of the reference (populated by Trainer after every update)
        self.total_loss = self.base_loss = self.policy_loss = self.value_loss = None
        self.entropy = self.pc_loss = self.vr_loss = self.rp_loss = None
        self._b1 = None
        self._shadow = None
        self._shadow_stale = True
        # absmax slots (fp16x2 GEMM scales, csrc/gemm_split.hip): a pool zeroed once per pass; a constant 1.0 for tensors
        # bounded by 1 (LSTM outputs); lar_bounded: [last action one-hot | last reward] stays within 1 (maze rewards are
        # -1 / 0 / +1, the Lab contract clips) -- Trainer clears it for environments that feed raw rewards
        self.slots = ops.AbsmaxPool(self._device)
        self.pass_id = 0
        self._one = torch.ones(1, dtype=torch.float32, device=self._device)
        self.lar_bounded = True

    def bind_frame_scale(self, scale):
        """Adopt the byte scale of the environment whose ring this network reads (raises if the caller fixed another)."""
        scale = float(scale)
        if self._frame_scale_given and abs(self.frame_scale - scale) > 1e-12:
            raise ValueError("UnrealModel(frame_scale=%r) but the environment stores frames at scale %r"
                             % (self.frame_scale, scale))
        self.frame_scale = scale

    # -- bf16x3 weight shadows (W operand of ops.gemm_split_nt) ---------------------------------------
    def refresh_shadows(self, only_if_stale=False):
        """Re-split the dense-layer weights into the bf16x3 planes the split-operand GEMM multiplies by (forward:
        transposed, dgrad: natural layout).  Cheap (a few MB); called at the start of every Trainer.process /
        Evaluate.process / batch-1 runner, i.e. after any optimiser step, load or restore.  `only_if_stale`: skip
        when no load / sync / optimiser step was announced (mark_params_changed) since the last split -- the replay
        fill makes 2000 policy steps on frozen weights."""
        if only_if_stale and self._shadow is not None and not self._shadow_stale:
            return
        self._shadow_stale = False
        if self._shadow is None:
            p, A = self.p, self._action_size
            K_x = self.K_x
            # every shadow's absmax slot lives in ONE tensor and all of them are refreshed by one fill + two launches
            # (ops.ShadowSet; per matrix it was fill + maximum + split: ~28 launches in front of every learner pass)
            ss = ops.ShadowSet(self._device, 12)
            S = lambda *a, **k: ss.add(ops.SplitWeights(*a, wmax=ss.slot(), defer=True, **k))
            sh = dict(fc1_fwd=S(p["W_base_fc1"], 2592, 256, 256, True),
                      fc1_dgrad=S(p["W_base_fc1"], 2592, 256, 256, False))
            if self._use_lstm:
                W = p["lstm_kernel"]
                sh.update(lstm_xh_fwd=ss.add(ops.LstmKernelShadow(W, K_x, wmax=ss.slot(), defer=True)),   # whole kernel, single-step launches
                          lstm_h_dgrad=S(W, 256, 1024, 1024, False, offset=K_x * 1024),
                          lstm_fc_dgrad=S(W, 256, 1024, 1024, False))
                if self.hoist_lstm_x:        # the hoisted schedule's input / recurrent halves (off: nothing reads them)
                    sh.update(lstm_x_fwd=S(W, K_x, 1024, 1024, True),
                              lstm_h_fwd=S(W, 256, 1024, 1024, True, offset=K_x * 1024, row_perm=1))   # gate-interleaved
            if self._use_pixel_change:
                sh.update(pc_fc1_fwd=S(p["W_pc_fc1"], 256, 2592, 2592, True),
                          pc_fc1_dgrad=S(p["W_pc_fc1"], 256, 2592, 2592, False))
            self._shadow, self._shadow_set = sh, ss
        self._shadow_set.refresh()
        if self.prepare_encoder:
            # the conv weights' share of encoder_fwd's prologue (scales + MFMA operand fragments), once per update instead
            # of once per workgroup of each of the ~24 encoder launches that follow
            p = self.p
            self._enc_prep = ops.encoder_prepare(p["W_base_conv1"], p["b_base_conv1"], p["W_base_conv2"], self.frame_scale,
                                                 self._enc_prep)

    # False: every encoder_fwd workgroup derives scales and fragments itself (identical results; A/B: r04_ab_summary.md 3f)
    prepare_encoder = True
    _enc_prep = None

    @property
    def enc_prepared(self):
        """Block of ops.encoder_prepare for the current conv weights and self.frame_scale (None: prepare_encoder off)."""
        if not self.prepare_encoder:
            return None
        if self._enc_prep is None or self._shadow_stale:
            self.refresh_shadows()
        return self._enc_prep

    @property
    def shadow(self):
        if self._shadow is None:
            self.refresh_shadows()
        return self._shadow

    def _hoist_shadows(self):
        """The hoisted LSTM schedule's two shadows, made on first use (hoist_lstm_x flipped on after construction, or a caller
        that runs lstm_step(fused_x=False) by hand)."""
        sh = self.shadow
        if "lstm_x_fwd" not in sh:
            ss, W, K_x = self._shadow_set, self.p["lstm_kernel"], self.K_x
            S = lambda *a, **k: ss.add(ops.SplitWeights(*a, wmax=ss.slot(), defer=True, **k))
            sh.update(lstm_x_fwd=S(W, K_x, 1024, 1024, True),
                      lstm_h_fwd=S(W, 256, 1024, 1024, True, offset=K_x * 1024, row_perm=1))
            ss.refresh()
        return sh

    # -- parameters ---------------------------------------------------------------------------------
    def _init_weights(self, seed):
        """U(+-1/sqrt(fan_in)) for W and b (model.py:31-42,752-783); LSTM kernel glorot_uniform, bias 0."""
        rs = np.random.RandomState(seed)
        for name, shape, fan_in in self.spec:
            if fan_in is None:
                lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                v = rs.uniform(-lim, lim, size=shape)
            elif fan_in == 0:
                v = np.zeros(shape)
            else:
                d = 1.0 / math.sqrt(fan_in)
                v = rs.uniform(-d, d, size=shape)
            self.p[name].copy_(torch.as_tensor(v.reshape(-1), dtype=torch.float32))

    def mark_params_changed(self):
        self._shadow_stale = True

    def load_named(self, named):
        """Load {name: array} (TF layouts), e.g. parameters exported by another implementation."""
        self._shadow_stale = True
        for k, v in named.items():
            self.p[k].copy_(torch.as_tensor(np.asarray(v, dtype=np.float32).reshape(-1)))

    def export_named(self):
        return OrderedDict((n, self.params.shaped(n).detach().cpu().numpy().copy()) for n, _, _ in self.spec)

    def get_vars(self):
        return self.variables

    def get_global_vars(self):
        return self.variables

    def sync_from(self, src_network, name=None):
        """global -> local copy (model.py:737-749).  One parameter copy per GPU: nothing to do."""
        if src_network is not self:
            self.params.flat.copy_(src_network.params.flat)
            self._shadow_stale = True
        return None

    def prepare_loss(self):
        return None

    def reset_state(self):
        z = torch.zeros(256, dtype=torch.float32, device=self._device)
        self.base_lstm_state_out = (z, z.clone())     # (c, h) like LSTMStateTuple

    # -- batched building blocks -----------------------------------------------------------------------
    def new_slot(self):
        """An absmax slot of the current pass (ops.AbsmaxPool; Trainer / Evaluate / the batch-1 runners call
        begin_pass() once per pass: the pool is zeroed with one fill)."""
        return self.slots.new()

    def begin_pass(self):
        self.slots.reset()
        self.pass_id += 1

    def ws_slots(self, ws):
        """The workspace's absmax slots of the current pass (max f2, max LSTM input x, max c1), taken on first use."""
        if getattr(ws, "pass_id", None) != self.pass_id:
            ws.s_f2, ws.s_x, ws.s_c1, ws.pass_id = self.new_slot(), self.new_slot(), self.new_slot(), self.pass_id
        return ws.s_f2, ws.s_x, ws.s_c1

    def encode_rows(self, ring, ws, row0, nrows, lar_from_ring=True, save_c1=True, clip_lar=False,
                    objective_slot_offset=0, actor_ring=None, lar_prefilled=False, lstm_x=True, slots=None):
        """conv encoder -> fc (+ last_action_reward[_objective] columns and the input half of the LSTM gates) for
        rows [row0, row0+nrows) of a path workspace.  `objective_slot_offset` = -1 reproduces trainer.py:300, where the
        bootstrap value is fed the objective of the previous frame's state.  `lstm_x` False: a single time step follows
        whose lstm_step(fused_x=True) multiplies [x | h] by the whole kernel, so the input half is not hoisted."""
        p = self.p
        idx = ws.frame_idx[row0:row0 + nrows]
        f2 = ws.f2[row0 * ops.F2_DIM:]
        xcat = ws.xcat[row0 * self.xld:]
        c1 = ws.c1[row0 * ops.C1_DIM:] if (save_c1 and ws.c1 is not None) else None
        bits = ws.f2_bits[row0 * ops.RELU_WORDS:] if (c1 is not None and ws.f2_bits is not None) else None
        # absmax slots: the encoder commits max f2, the fc GEMM reads it and commits max of its own output -- the scale of
        # the product that multiplies the fc row next (LSTM step: [fc | last action, reward, objective | h])
        # The slots belong to the WORKSPACE for the pass: a rollout encodes its rows block by block, the wgrad products of
        # the backward read all of them, so every block maxes into the same two slots (a block's own products then use the
        # running maximum: >= its rows', deterministic because the launches are stream-ordered).  Blocks encoded on
        # separate streams (host-fed half-batches, `actor_ring`) take fresh slots and merge them into the workspace's.
        # `slots` = (s_f2, s_x): a caller that runs row blocks of one workspace on several streams (Trainer's half-batch
        # rollout) hands every stream its own running pair and merges them into the workspace's once, after the join.
        self.ws_slots(ws)
        own = actor_ring is not None and slots is None
        s_f2, s_fc = slots if slots is not None else ((self.new_slot(), self.new_slot()) if own else (ws.s_f2, ws.s_x))
        ops.encoder_fwd(ring.frames, idx, self.frame_scale, p["W_base_conv1"], p["b_base_conv1"],
                        p["W_base_conv2"], p["b_base_conv2"], f2, c1, relu_bits=bits, f2_max=s_f2,
                        c1_max=ws.s_c1 if c1 is not None else None, prepared=self.enc_prepared)
        sh = self.shadow
        nslab = ops.slab_count(nrows, 256, 2592) if self.fc_few_rows_slabs else 0
        if nslab:
            # few rows (a group's rollout step, a small update's replay pass): 4-64 tiles of 81 dependent K steps would
            # leave most of the chip idle -- 2-8 K slabs in separate workgroups + an ordered sum (one partials buffer per stream:
            # blocks of one workspace may be encoded on several streams at once)
            key = torch.cuda.current_stream(f2.device).cuda_stream
            part = ws.fc_partials.get(key)
            if part is None or part.numel() < nslab * nrows * 256:
                part = ws.fc_partials[key] = torch.empty(nslab * nrows * 256, dtype=torch.float32, device=f2.device)
            ops.gemm_split_nt_slabs(nrows, 256, 2592, f2, 2592, sh["fc1_fwd"], xcat, self.xld, part, nslab,
                                    bias=p["b_base_fc1"], flags=ops.GEMM_RELU, a_max=s_f2, c_max=s_fc)
        else:
            ops.gemm_split_nt(nrows, 256, 2592, f2, 2592, sh["fc1_fwd"], xcat, self.xld, bias=p["b_base_fc1"],
                              flags=ops.GEMM_RELU, a_max=s_f2, c_max=s_fc)
        ws.s_x_cur = s_fc          # max over the fc columns; the other columns of x are added below where they can exceed 1
        if own:
            ops.absmax(1, 1, s_f2, 1, ws.s_f2)
        if not self._use_lstm:
            if own:
                ops.absmax(1, 1, s_fc, 1, ws.s_x)
            return
        A = self._action_size
        if lar_prefilled:
            pass                   # the environment step kernel has written the [last action | last reward] columns
        elif lar_from_ring:
            ops.lar_fill(nrows, A, ring.r_last_action, ring.r_last_reward, idx, xcat, self.xld)
        else:
            ar = ring if actor_ring is None else actor_ring     # per-actor state of a sub-range of the ring's actors
            ops.lar_fill(nrows, A, ar.last_action, ar.last_reward, None, xcat, self.xld, clip=clip_lar)
        if self._objective_size:
            ops.objective_fill(ring, nrows, idx, xcat, self.xld, 256 + A + 1, slot_offset=objective_slot_offset)
        # [one-hot last action | last reward | objective]: the one-hot and a clipped reward stay within the kernel's own
        # floor of 1 (|h| < 1); raw rewards / measurement vectors (host-fed actors) are reduced into the slot
        if self._objective_size or not self.lar_bounded:
            ops.absmax(nrows, self.K_x - 256, xcat[256:], self.xld, s_fc)
        if own:
            ops.absmax(1, 1, s_fc, 1, ws.s_x)
        if lstm_x:
            ops.gemm_split_nt(nrows, 1024, self.K_x, xcat, self.xld, self._hoist_shadows()["lstm_x_fwd"], ws.gates[row0 * 1024:], 1024,
                              a_max=ops.absmax(nrows, self.K_x, xcat, self.xld, self.new_slot()))

    def lstm_step(self, ws, t, B, b0=0, nrows=None, fused_x=False):
        """One BasicLSTMCell step for time row-block t; `b0`, `nrows`: only the actors [b0, b0 + nrows) of the block.
        fused_x False: ws.gates holds the hoisted input half, the recurrent half + gate math run here; True (one step at
        a time: rollout, bootstrap, batch-1 runners): [x | h] @ kernel in this one launch."""
        p = self.p
        n = B if nrows is None else nrows
        h_prev = ws.h0[b0 * 256:] if t == 0 else ws.h[((t - 1) * B + b0) * 256:]
        c_prev = ws.c0[b0 * 256:] if t == 0 else ws.c[((t - 1) * B + b0) * 256:]
        g_t = ws.gates[(t * B + b0) * 1024:]
        if fused_x:
            ops.lstm_step_fwd(n, h_prev, self.shadow["lstm_xh_fwd"], g_t, p["lstm_bias"], c_prev,
                              ws.c[(t * B + b0) * 256:], ws.h[(t * B + b0) * 256:],
                              x=ws.xcat[(t * B + b0) * self.xld:], ldx=self.xld, Kx=self.K_x, x_max=ws.s_x_cur)
            return
        ops.lstm_step_fwd(n, h_prev, self._hoist_shadows()["lstm_h_fwd"], g_t, p["lstm_bias"], c_prev, ws.c[(t * B + b0) * 256:],
                          ws.h[(t * B + b0) * 256:])

    def features(self, ws, row0=0):
        """(tensor, ld) of the features the heads read: LSTM outputs, or the fc output in FF mode."""
        if self._use_lstm:
            return ws.h[row0 * 256:], 256
        return ws.xcat[row0 * self.xld:], self.xld

    # True: the input half of the gates of a T-step training sequence is one [T*B, K_x] x [K_x, 1024] product ahead of the
    # recurrence and each step multiplies only h (the round-1 schedule).  False: every step multiplies [x | h] by the
    # whole kernel -- at 4096 rows a step is bound by operand bytes, not FLOPs, and the hoisted form writes and re-reads
    # 16 MB of pre-activations per step on top (measured: 52 -> 30 us per step of a sequence, tools/bench_kernels.py lstm).
    # fc 2592 -> 256 at <= ops.FEW_ROWS (2048) rows as K slabs + ordered sum (unreal_gemm_f32_split_nt_slabs); False: the
    # one-launch kernel at every row count (A/B: profiles/r04_ab_summary.md 3e)
    fc_few_rows_slabs = True
    hoist_lstm_x = False
    relu_bits = True           # fc dgrad masks with the encoder's 1-bit ReLU pattern instead of re-reading f2 (849 MB per branch)
    fuse_bptt = True           # BPTT: recurrent dgrad + the earlier step's gate backward in one launch (lstm_bptt_step)

    def trunk_forward(self, ring, ws, T, B, lar_from_ring=True, save_c1=True, clip_lar=False,
                      objective_slot_offset=0):
        """conv encoder -> fc -> (LSTM over T steps from ws.c0/ws.h0); rows = T*B listed in ws.frame_idx."""
        hoist = self.hoist_lstm_x and T > 1
        self.encode_rows(ring, ws, 0, T * B, lar_from_ring, save_c1, clip_lar, objective_slot_offset, lstm_x=hoist)
        if self._use_lstm:
            for t in range(T):
                self.lstm_step(ws, t, B, fused_x=not hoist)
        return self.features(ws)

    def trunk_backward(self, ring, ws, gws, T, B, d_feat, h0_nonzero=False):
        """Back-propagate d_feat [T*B,256] through LSTM, fc and the conv encoder into self.g."""
        p, g, sh = self.p, self.g, self.shadow
        rows = T * B
        if self._use_lstm:
            A = self._action_size
            K_x = self.K_x
            W = p["lstm_kernel"]
            Wh = W[K_x * 1024:]
            gws.dc.zero_()
            # absmax slots of d_gates: one per time step (the A scale of the step's recurrent product) and one over the
            # whole sequence (the A scale of the fc dgrad below); the gate-backward epilogues commit into both
            s_all = self.new_slot()
            s_step = self.new_slot()
            if self.fuse_bptt:
                # the last step's gate backward stands alone; every earlier step's runs in the epilogue of the product
                # that yields its dh_rec (one launch per step instead of two, dh_rec never written)
                c_prev = ws.c0 if T == 1 else ws.c[(T - 2) * B * 256:]
                ops.lstm_gates_bwd(B, d_feat[(T - 1) * B * 256:], None, gws.dc, ws.gates[(T - 1) * B * 1024:], c_prev,
                                   ws.c[(T - 1) * B * 256:], gws.d_gates[(T - 1) * B * 1024:], c_max0=s_step, c_max1=s_all)
                for t in reversed(range(1, T)):
                    c_prev = ws.c0 if t == 1 else ws.c[(t - 2) * B * 256:]
                    s_next = self.new_slot()
                    ops.lstm_bptt_step(B, gws.d_gates[t * B * 1024:], sh["lstm_h_dgrad"], d_feat[(t - 1) * B * 256:],
                                       gws.dc, ws.gates[(t - 1) * B * 1024:], c_prev, ws.c[(t - 1) * B * 256:],
                                       gws.d_gates[(t - 1) * B * 1024:], a_max=s_step, c_max0=s_next, c_max1=s_all)
                    s_step = s_next
            else:                                      # the two-kernel form (bit-identical; kept for A/B timing)
                for t in reversed(range(T)):
                    c_prev = ws.c0 if t == 0 else ws.c[(t - 1) * B * 256:]
                    s_step = self.new_slot()
                    ops.lstm_gates_bwd(B, d_feat[t * B * 256:], gws.dh_rec if t < T - 1 else None, gws.dc,
                                       ws.gates[t * B * 1024:], c_prev, ws.c[t * B * 256:], gws.d_gates[t * B * 1024:],
                                       c_max0=s_step, c_max1=s_all)
                    if t > 0:
                        ops.gemm_split_nt(B, 256, 1024, gws.d_gates[t * B * 1024:], 1024, sh["lstm_h_dgrad"], gws.dh_rec, 256,
                                          a_max=s_step)
            dW = g["lstm_kernel"]
            # input half of the kernel gradient: the 256 fc rows as two exact 128-row MFMA tiles, the A+1
            # last_action_reward rows by the small-N outer-product kernel (no padded third tile)
            ops.gemm_split_tn(256, 1024, rows, ws.xcat, self.xld, gws.d_gates, 1024, dW, 1024,
                              splitk=_splitk(256, 1024, rows), colsum=g["lstm_bias"],     # + bias gradient
                              a_max=ws.s_x, b_max=s_all)
            c0 = 256
            for w in _small_chunks(K_x - 256):         # last action, last reward (and objective) rows
                ops.linear_small_bwd(rows, 1024, w, gws.d_gates, 1024, ws.xcat[c0:], self.xld, None, None, 0, False,
                                     dW[c0 * 1024:], None, dw_stride_k=1, dw_stride_n=1024)
                c0 += w
            if T > 1:
                r1 = (T - 1) * B
                ops.gemm_split_tn(256, 1024, r1, ws.h, 256, gws.d_gates[B * 1024:], 1024, dW[K_x * 1024:], 1024,
                              splitk=_splitk(256, 1024, r1), a_max=self._one, b_max=s_all)      # |h| < 1
            if h0_nonzero:
                ops.gemm_split_tn(256, 1024, B, ws.h0, 256, gws.d_gates, 1024, dW[K_x * 1024:], 1024,
                              splitk=_splitk(256, 1024, B), a_max=self._one, b_max=s_all)
            s_dfc = self.new_slot()
            ops.gemm_split_nt(rows, 256, 1024, gws.d_gates, 1024, sh["lstm_fc_dgrad"], gws.d_fc, 256, mask=ws.xcat,
                              ldm=self.xld, flags=ops.GEMM_RELU_MASK, a_max=s_all, c_max=s_dfc)
            d_fc = gws.d_fc
        else:
            ops.relu_mask(rows, 256, d_feat, 256, ws.xcat, self.xld)
            d_fc = d_feat
            s_dfc = ops.absmax(rows, 256, d_fc, 256, self.new_slot())
        ops.gemm_split_tn(2592, 256, rows, ws.f2, 2592, d_fc, 256, g["W_base_fc1"], 256,
                              splitk=_splitk(2592, 256, rows), colsum=g["b_base_fc1"], a_max=ws.s_f2, b_max=s_dfc)
        s_df2 = self.new_slot()            # max |d_f2|: committed by the fc dgrad's epilogue, the d2 scale of the conv backward
        if ws.f2_bits is not None and self.relu_bits:
            ops.gemm_split_nt(rows, 2592, 256, d_fc, 256, sh["fc1_dgrad"], gws.d_f2, 2592, mask=ws.f2_bits,
                              ldm=ops.RELU_WORDS, flags=ops.GEMM_RELU_BITS, a_max=s_dfc, c_max=s_df2)
        else:
            ops.gemm_split_nt(rows, 2592, 256, d_fc, 256, sh["fc1_dgrad"], gws.d_f2, 2592, mask=ws.f2, ldm=2592,
                              flags=ops.GEMM_RELU_MASK, a_max=s_dfc, c_max=s_df2)
        ops.encoder_bwd(ring.frames, ws.frame_idx[:rows], self.frame_scale, p["W_base_conv2"], ws.c1, gws.d_f2,
                        g["W_base_conv1"], g["b_base_conv1"], g["W_base_conv2"], g["b_base_conv2"],
                        c1_max=ws.s_c1, d2_max=s_df2)

    def heads_forward(self, rows, feat, ld, pi_out, v_out):
        p, A = self.p, self._action_size
        ops.linear_small_fwd(rows, 256, A, feat, ld, p["W_base_fc_p"], p["b_base_fc_p"], pi_out, A)
        ops.linear_small_fwd(rows, 256, 1, feat, ld, p["W_base_fc_v"], p["b_base_fc_v"], v_out, 1)

    def policy_step(self, rows, feat, ld, u, pi_out, v_out, actions_out):
        """heads_forward + softmax / action draw in one launch (u None: greedy)."""
        p = self.p
        ops.policy_step(rows, self._action_size, feat, ld, p["W_base_fc_p"], p["b_base_fc_p"], p["W_base_fc_v"],
                        p["b_base_fc_v"], u, pi_out, v_out, actions_out)

    def value_forward(self, rows, feat, ld, v_out):
        p = self.p
        ops.linear_small_fwd(rows, 256, 1, feat, ld, p["W_base_fc_v"], p["b_base_fc_v"], v_out, 1)

    def pc_head_forward(self, rows, feat, ld, hp, ws=None, hp_max=None):
        """hp = relu(feat @ W_pc_fc1 + b).  `ws`: the workspace the features live in (its absmax slot of the fc rows is the
        A scale in FF mode; LSTM outputs are bounded by 1: a slot holding 1.0).  `hp_max`: absmax slot that receives
        max hp -- the scale of hp's fp16 planes in the deconvolution kernels."""
        p = self.p
        if self._use_lstm:
            a_max = self._one
        else:
            a_max = ws.s_x if ws is not None and getattr(ws, "s_x", None) is not None else None
        ops.gemm_split_nt(rows, 2592, 256, feat, ld, self.shadow["pc_fc1_fwd"], hp, 2592, bias=p["b_pc_fc1"],
                          flags=ops.GEMM_RELU, a_max=a_max, c_max=hp_max)

    # -- reference batch-1 runners (model.py:630-728) ---------------------------------------------------
    def _b1_ws(self):
        if self._b1 is None:
            dev = self._device
            self._b1 = dict(ring=ops.Ring(3, 1, dev, objective_size=self._objective_size), ws=PathWS(3, 3, dev, save_c1=False, lstm=self._use_lstm, xld=self.xld),
                            pi=torch.zeros(self._action_size, device=dev), v=torch.zeros(1, device=dev),
                            hp=torch.zeros(2592, device=dev), q=torch.zeros(400, device=dev),
                            z=torch.zeros(3, device=dev))
        return self._b1

    def _stage(self, images, last_action_reward=None):
        """Quantise float images in [0,1] to the uint8 frame pool (frame_scale 1/255) of a scratch ring."""
        b1 = self._b1_ws()
        ring, ws = b1["ring"], b1["ws"]
        n = len(images)
        for k, img in enumerate(images):
            a = np.asarray(img, dtype=np.float64)
            u8 = np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8).reshape(-1)
            ring.frames[k * ops.FRAME_BYTES:(k + 1) * ops.FRAME_BYTES].copy_(torch.from_numpy(u8))
        ws.frame_idx[:n].copy_(torch.arange(n, dtype=torch.int32))
        if last_action_reward is not None:
            lar = np.asarray(last_action_reward, dtype=np.float32)
            ring.r_last_action[0] = int(np.argmax(lar[:self._action_size]))
            ring.r_last_reward[0] = float(lar[self._action_size])
            if self._objective_size:                   # [one-hot action | reward | objective] (experience.py:42-44)
                ring.r_objective[:self._objective_size].copy_(torch.from_numpy(lar[self._action_size + 1:]))
        return ring, ws

    def _run_trunk1(self, s_t, last_action_reward, state):
        self.refresh_shadows()                 # the caller may have changed the weights since the last call
        self.begin_pass()
        ring, ws = self._stage([s_t['image']], last_action_reward)
        scale, bounded = self.frame_scale, self.lar_bounded
        self.frame_scale = 1.0 / 255.0
        self.lar_bounded = False               # the caller's last_action_reward vector is arbitrary
        try:
            if self._use_lstm:
                ws.c0[:256].copy_(state[0].reshape(-1))
                ws.h0[:256].copy_(state[1].reshape(-1))
            feat, ld = self.trunk_forward(ring, ws, 1, 1, lar_from_ring=True, save_c1=False)
        finally:
            self.frame_scale, self.lar_bounded = scale, bounded
        return ws, feat, ld

    def run_base_policy_and_value(self, sess, s_t, last_action_reward, mode=""):
        b1 = self._b1_ws()
        ws, feat, ld = self._run_trunk1(s_t, last_action_reward, self.base_lstm_state_out)
        self.heads_forward(1, feat, ld, b1["pi"], b1["v"])
        ops.softmax_sample(1, self._action_size, b1["pi"], self._action_size)
        if self._use_lstm:
            self.base_lstm_state_out = (ws.c[:256].clone(), ws.h[:256].clone())
        return b1["pi"].cpu().numpy(), float(b1["v"].cpu()[0]), None

    def run_base_value(self, sess, s_t, last_action_reward):
        b1 = self._b1_ws()
        ws, feat, ld = self._run_trunk1(s_t, last_action_reward, self.base_lstm_state_out)   # state NOT advanced
        self.value_forward(1, feat, ld, b1["v"])
        return float(b1["v"].cpu()[0])

    def _zero_state(self):
        z = torch.zeros(256, dtype=torch.float32, device=self._device)
        return (z, z)

    def run_pc_q_max(self, sess, s_t, last_action_reward):
        b1 = self._b1_ws()
        ws, feat, ld = self._run_trunk1(s_t, last_action_reward, self._zero_state())
        self.pc_head_forward(1, feat, ld, b1["hp"], ws=ws)
        p = self.p
        ops.pc_deconv_fwd(1, self._action_size, b1["hp"], p["W_pc_deconv_v"], p["b_pc_deconv_v"],
                          p["W_pc_deconv_a"], p["b_pc_deconv_a"], qmax=b1["q"])
        return b1["q"].cpu().numpy().reshape(20, 20)

    def run_vr_value(self, sess, s_t, last_action_reward):
        b1 = self._b1_ws()
        ws, feat, ld = self._run_trunk1(s_t, last_action_reward, self._zero_state())
        self.value_forward(1, feat, ld, b1["v"])
        return float(b1["v"].cpu()[0])

    def run_rp_c(self, sess, state_history):
        b1 = self._b1_ws()
        ring, ws = self._stage([s['image'] for s in state_history])
        p = self.p
        ops.encoder_fwd(ring.frames, ws.frame_idx[:3], 1.0 / 255.0, p["W_base_conv1"], p["b_base_conv1"],
                        p["W_base_conv2"], p["b_base_conv2"], ws.f2, None)
        ops.linear_small_fwd(1, 7776, 3, ws.f2, 7776, p["W_rp_fc1"], p["b_rp_fc1"], b1["z"], 3)
        ops.softmax_sample(1, 3, b1["z"], 3)
        return b1["z"].cpu().numpy()
