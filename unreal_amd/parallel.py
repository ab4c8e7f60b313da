"""Multi-GPU: one process per GPU, actors sharded by rank, ONE exchange per update.

The reference has no collective (threads race on one variable set: main.py:455,
rmsprop_applier.py:86-93).  Here rank r owns actors [r*B, (r+1)*B) with their env state, LSTM state
and replay ring entirely in its own HBM; the only cross-GPU step is a SUM all-reduce of the flat
gradient buffer (each rank's contribution is already scaled by 1/(B*world), so the sum is the mean
over all actors).  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests."""
import os

import torch
import torch.distributed as dist


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend=None):
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("UNREAL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def actor_range(rank, per_rank_actors):
    return rank * per_rank_actors, (rank + 1) * per_rank_actors


def all_reduce_sum(flat):
    """In-place SUM all-reduce of one flat buffer (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
