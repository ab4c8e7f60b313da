"""N > 1 path on CPU: world_size 2 over gloo (127.0.0.1).  Each rank owns its shard of the actors,
produces its local mean gradient in the product's flat layout (the per-actor gradients come from the
oracle here -- the HIP kernels need a GPU), and the product's exchange step (`parallel.all_reduce_sum`,
the one collective of the path) must reproduce the single-process mean over ALL actors."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle.trainer import OracleTrainer, ExplicitDraws

CFG = dict(action_size=4, use_lstm=True, use_pixel_change=True, use_value_replay=True,
           use_reward_prediction=True, pixel_change_lambda=0.05, entropy_beta=0.001, local_t_max=5,
           n_step_TD=5, gamma=0.99, gamma_pc=0.9, experience_history_size=12, max_time_step=10 ** 6,
           rmsp_alpha=0.99, rmsp_epsilon=0.1, grad_norm_clip=40.0, initial_alpha_low=1e-4,
           initial_alpha_high=5e-3, initial_alpha_log_rate=0.5)


def _actor_draws(gid):
    rs = np.random.RandomState(1000 + gid)
    d = ExplicitDraws()
    d.action_u = list(rs.random_sample(CFG["experience_history_size"] + CFG["n_step_TD"]))
    d.seq_starts = list(rs.randint(0, CFG["experience_history_size"] - CFG["local_t_max"] - 2, size=2))
    d.rp_coin = [int(rs.randint(2))]
    d.rp_u = [float(rs.random_sample())]
    return d


def _local_mean_grad(actor_ids, scale):
    """Sum over this shard's actors of (per-actor gradient * scale), flattened in the product layout."""
    from unreal_amd.model.model import FlatParams, param_spec
    tr = OracleTrainer(CFG, n_actors=len(actor_ids), draws=[_actor_draws(g) for g in actor_ids], seed=7,
                       dtype=torch.float64)
    tr.fill()
    fp = FlatParams(param_spec(4), "cpu")
    flat = torch.zeros(fp.size, dtype=torch.float64)
    views = fp.make_views(flat)
    steps = 0
    for a in tr.actors:
        t0 = a.local_t
        a.draws.action_u = a.draws.action_u[:CFG["n_step_TD"]]
        g, _, _, _ = tr.actor_grad(a)
        steps += a.local_t - t0
        for (name, _), gi in zip(tr.params.items(), g):
            views[name] += gi.reshape(-1) * scale
    return flat, steps


def _worker(rank, world, port, per_rank, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from unreal_amd import parallel
    r, lr, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = parallel.actor_range(rank, per_rank)
    flat, steps = _local_mean_grad(list(range(lo, hi)), 1.0 / (per_rank * world))
    parallel.all_reduce_sum(flat)
    tot_steps = parallel.sum_over_ranks([steps], "cpu")[0]
    mx = parallel.max_over_ranks(float(rank), "cpu")
    parallel.barrier()
    q.put((rank, flat.numpy(), tot_steps, mx))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_exchange_equals_global_mean():
    world, per_rank = 2, 1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref, steps = _local_mean_grad(list(range(world * per_rank)), 1.0 / (world * per_rank))
    for rank, flat, tot_steps, mx in res:
        np.testing.assert_allclose(flat, ref.numpy(), rtol=1e-12, atol=1e-14)
        assert tot_steps == steps and mx == world - 1
    np.testing.assert_array_equal(res[0][1], res[1][1])       # replicas stay bit-identical
    assert np.abs(ref.numpy()).max() > 0
