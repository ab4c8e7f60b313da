"""Multi-GPU: one process per GPU, actors sharded by rank, ONE exchange per update.

The reference has no collective (threads race on one variable set: main.py:455,
rmsprop_applier.py:86-93).  Here rank r owns actors [r*B, (r+1)*B) with their env state, LSTM state
and replay ring entirely in its own HBM; the only cross-GPU step is a SUM all-reduce of the flat
gradient buffer (each rank's contribution is already scaled by 1/(B*world), so the sum is the mean
over all actors).  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests and by
the rehearsal of several ranks on ONE device (RCCL refuses two ranks on the same GPU).

`launch_ranks` is the self-launcher behind `python bench.py --gpus N`: it starts N fresh child
processes with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set (never re-executing a process that has touched
the GPU), relays rank 0's stdout and returns the worst exit code."""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def device_index(local_rank=None):
    """GPU of this rank: LOCAL_RANK, or UNREAL_FORCE_DEVICE when several ranks rehearse on one device."""
    if local_rank is None:
        local_rank = dist_env()[1]
    return int(os.environ.get("UNREAL_FORCE_DEVICE", local_rank))


def default_backend():
    """nccl (= RCCL) for one rank per GPU; gloo when there is no GPU or ranks share a device."""
    forced = os.environ.get("UNREAL_DIST_BACKEND")
    if forced:
        return forced
    if not torch.cuda.is_available():
        return "gloo"
    if "UNREAL_FORCE_DEVICE" in os.environ and dist_env()[2] > 1:
        return "gloo"            # RCCL: "Duplicate GPU detected" for two ranks on one device
    return "nccl"


def init_distributed(backend=None, force=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (no-op for a single process unless
    `force`, which builds a 1-rank group: that is how the RCCL path is exercised on a one-GPU box)."""
    rank, local_rank, world = dist_env()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = default_backend()
        kw = {}
        if backend == "nccl":
            dev = device_index(local_rank)
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def backend_name():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def _active():
    return dist.is_available() and dist.is_initialized()


def actor_range(rank, per_rank_actors):
    return rank * per_rank_actors, (rank + 1) * per_rank_actors


def all_reduce_sum(flat):
    """In-place SUM all-reduce of one flat buffer (no-op without a process group)."""
    if _active():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def all_true(flag, device):
    """True iff `flag` holds on EVERY rank (MIN all-reduce): the replay-full phase switch must be taken by all ranks
    in the same process() call, or their gradient all-reduces would pair up off by one call and hang."""
    if not (_active() and dist.get_world_size() > 1):
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _active() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if _active() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def gather_over_ranks(value, device):
    """[value of rank 0, rank 1, ...] on every rank (one float per rank).  A SUM all-reduce of a one-hot-placed vector:
    gloo has no all_gather for device tensors, all_reduce works on both backends."""
    if not (_active() and dist.get_world_size() > 1):
        return [float(value)]
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def world_size():
    """Ranks of the process group that actually formed (1 without one)."""
    return dist.get_world_size() if _active() else 1


def barrier():
    if _active() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown():
    if _active():
        dist.destroy_process_group()


def visible_gpus():
    """GPUs this process would see, counted WITHOUT touching the HIP / HSA runtime (the self-launcher must stay a process
    that never initialised the GPU): KFD topology nodes with SIMDs, cut by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES when set.  None if the topology is not readable (then the ranks find out themselves)."""
    import glob
    n = 0
    try:
        nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
        if not nodes:
            return None
        for f in nodes:
            for line in open(f):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


# ---- self-launcher ----------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Terminated(Exception):
    pass


def launch_ranks(script, argv, world, extra_env=None, poll_s=0.5, grace_s=20.0):
    """Start `world` children `python script argv...`, one per rank, from a parent that has not touched the GPU.
    Rank 0's stdout is relayed to this process's stdout, the other ranks' stdout goes to stderr (so the caller
    still sees exactly one JSON line).  If a rank dies, the survivors get `grace_s` seconds, then are terminated by
    PID (they would otherwise wait in a collective forever).  The same happens when THIS process is interrupted
    (KeyboardInterrupt) or receives SIGTERM / SIGHUP (an outer `timeout`): no rank outlives its launcher holding a
    GPU.  Returns the worst exit code (130 / 143 after an interrupt / a termination signal)."""
    import signal
    procs = []
    interrupted = 0
    state = {"reaping": False, "pending": 0}

    def _on_signal(signum, frame):
        # while the children are being reaped (or one is being started) a second TERM / HUP -- an outer `timeout -k`
        # sends TERM, then more -- must not abort that work: it is remembered, not raised
        if state["reaping"]:
            state["pending"] = signum
            return
        raise _Terminated(signum)

    old = {}
    for sig in (signal.SIGTERM, signal.SIGHUP):
        try:
            old[sig] = signal.signal(sig, _on_signal)
        except ValueError:               # not the main thread: the caller's own handlers stay
            pass

    def _reap():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10.0
        for p in procs:
            try:
                p.wait(max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        for p in procs:
            p.wait()

    try:
        # free_port() closes its probe socket before rank 0 binds the rendezvous port (a check-then-use window of a few
        # ms on 127.0.0.1); a collision makes rank 0 exit with "address already in use" and the launcher returns its code
        port = free_port()
        for r in range(world):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(world))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this host driver
            env.setdefault("OMP_NUM_THREADS", "4")
            if extra_env:
                env.update(extra_env)
            out = None if r == 0 else sys.stderr
            state["reaping"] = True                    # a signal between Popen() returning and append() would leak the child
            try:
                procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=out))
            finally:
                state["reaping"] = False
            if state["pending"]:
                raise _Terminated(state["pending"])
        first_fail = None
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if first_fail is None and any(c not in (None, 0) for c in codes):
                first_fail = time.time()
            if first_fail is not None and time.time() - first_fail > grace_s:
                break                                  # the finally block terminates the survivors
            time.sleep(poll_s)
    except KeyboardInterrupt:
        interrupted = 130
    except _Terminated as e:
        interrupted = 128 + int(e.args[0])
    finally:
        state["reaping"] = True                        # from here on signals are recorded, never raised: _reap() completes
        try:
            _reap()
        except KeyboardInterrupt:                      # Ctrl-C during the reap: finish it (children are killed by PID)
            interrupted = interrupted or 130
            _reap()
        for sig, h in old.items():
            signal.signal(sig, h)
        if state["pending"] and not interrupted:
            interrupted = 128 + int(state["pending"])
    if interrupted:
        return interrupted
    worst = 0
    for c in [p.returncode for p in procs]:
        if c != 0:
            worst = c if c > 0 else 128 - c
            break
    return worst
