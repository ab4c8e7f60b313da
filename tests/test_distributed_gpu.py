"""N > 1 path on the device: the REAL `Trainer.process()` with its gradient exchange, several ranks on ONE MI355X.

Replaces the reference's thread parallelism (main.py:453-464: `parallel_size` threads on one variable set, hogwild
RMSProp rmsprop_applier.py:86-93) by ranks that own disjoint actor shards and all-reduce the flat gradient.

* 2 ranks x B actors share cuda:0 (UNREAL_FORCE_DEVICE=0).  RCCL refuses two ranks on one GPU ("Duplicate GPU
  detected"), so this rehearsal exchanges the device gradient buffer over gloo; the RCCL ("nccl") path itself is
  exercised by `test_rccl_single_rank_group` below (a 1-rank RCCL communicator on the device) and by bench.py on
  multi-GPU nodes.  Asserted: parameters bit-identical across the ranks after 2 updates, and equal (fp32 summation
  order differs: 2e-6 abs + 2e-5 rel) to ONE process holding all 2B actors -- the device draws are sharding-invariant
  (PhiloxDraws), so both runs see the same trajectories.
* `python bench.py --gpus 2` typed as is (no outer torchrun) must self-launch its ranks and print one JSON line.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, H, T, UPDATES = 6, 40, 5, 2


def _make_trainer(rank, world, batch, device, grad_sync):
    from unreal_amd.environment.environment import Environment
    from unreal_amd.model.model import UnrealModel
    from unreal_amd.train.rmsprop_applier import RMSPropApplier
    from unreal_amd.train.trainer import Trainer
    Environment.action_size = -1
    A = Environment.get_action_size("maze", "")
    net = UnrealModel(A, 0, -1, True, True, True, True, 0.05, 0.001, device, seed=5)
    applier = RMSPropApplier(None, decay=0.99, momentum=0.0, epsilon=0.1, clip_norm=40.0, device=device)
    tr = Trainer(rank, net, 7.0711e-4, None, applier, "maze", "", True, True, True, True, 0.05, 0.001, T, T, 0.99, 0.9,
                 H, 10 ** 6, device, batch_size=batch, world_size=world, rank=rank, seed=0xA3C, grad_sync=grad_sync)
    tr.prepare()
    return net, tr


def _run(tr, net, world, batch):
    while not tr._full:
        tr.process(None, 0)
    g = 0
    steps = []
    for _ in range(UPDATES):
        s, _ = tr.process(None, g)
        steps.append(s)
        g += batch * T * world
    torch.cuda.synchronize()
    return net.params.flat.detach().cpu().numpy().copy(), steps


def _worker(rank, world, port, backend, outdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), UNREAL_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from unreal_amd import parallel
    r, lr, w = parallel.init_distributed(backend=backend, force=True)
    dev = torch.device("cuda", parallel.device_index(lr))
    torch.cuda.set_device(dev)
    net, tr = _make_trainer(rank, world, B, dev, parallel.all_reduce_sum)
    flat, steps = _run(tr, net, world, B)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), params=flat, steps=np.asarray(steps),
             backend=np.asarray(parallel.backend_name()))
    parallel.barrier()
    parallel.shutdown()


def _spawn(world, backend, outdir):
    from unreal_amd import parallel
    ctx = mp.get_context("spawn")
    port = parallel.free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, outdir)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.terminate()
    return codes


def test_two_ranks_on_one_device_match_one_process_with_all_actors(tmp_path):
    codes = _spawn(2, "gloo", str(tmp_path))
    assert codes == [0, 0], codes
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert str(r0["backend"]) == "gloo"
    np.testing.assert_array_equal(r0["params"], r1["params"])            # replicas never drift: bit-identical
    net, tr = _make_trainer(0, 1, 2 * B, torch.device("cuda", 0), None)
    ref, steps = _run(tr, net, 1, 2 * B)
    assert list(r0["steps"] + r1["steps"]) == steps                      # same trajectories (sharding-invariant draws)
    init = _make_trainer(0, 1, 1, torch.device("cuda", 0), None)[0].params.flat.cpu().numpy()
    assert np.abs(ref - init).max() > 1e-5                               # the updates did move the parameters
    np.testing.assert_allclose(r0["params"], ref, rtol=2e-5, atol=2e-6)


def test_rccl_single_rank_group(tmp_path):
    """backend "nccl" IS RCCL: a 1-rank communicator on the device runs the same all_reduce call on the flat gradient
    buffer that the N-GPU job issues; the update must equal the run without any process group (the weight-gradient
    kernels accumulate with float atomics, so two runs agree to summation order, not bit for bit)."""
    codes = _spawn(1, "nccl", str(tmp_path))
    assert codes == [0], codes
    r0 = np.load(tmp_path / "rank0.npz")
    assert str(r0["backend"]) == "nccl"
    net, tr = _make_trainer(0, 1, B, torch.device("cuda", 0), None)
    ref, steps = _run(tr, net, 1, B)
    assert list(r0["steps"]) == steps
    np.testing.assert_allclose(r0["params"], ref, rtol=2e-5, atol=2e-6)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` as typed: the parent starts 2 fresh rank processes before touching the GPU."""
    env = dict(os.environ, UNREAL_FORCE_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--actors", "64", "--history", "60"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["global_actors"] == 128 and "gloo" in out["config"]["parallelism"]
    assert "cpu_baseline" not in out                                    # rank 0 at N = 1 only
    # the N > 1 line is self-evidencing: what the exchange ran on, how many ranks formed, what each rank did, what the
    # exchange cost per update (HIP events around grad_sync inside the timed region) and how big the message is
    c = out["config"]
    assert c["backend"] == "gloo" and c["ranks"] == 2 and c["rccl_ranks"] == 0          # rehearsal on one device
    assert len(c["env_steps_per_rank"]) == 2 and sum(c["env_steps_per_rank"]) == round(c["env_steps_per_call"] * 2)
    assert all(0 < x <= 64 * 20 * 2 for x in c["env_steps_per_rank"])
    assert c["grad_message_bytes"] >= 1898877 * 4 and c["comm_ms_per_update"] > 0
    assert len(c["comm_ms_per_update_per_rank"]) == 2


def test_bench_refuses_more_ranks_than_devices():
    """`--gpus N` on a node with fewer GPUs (and no UNREAL_FORCE_DEVICE rehearsal) must fail fast with a clear message
    instead of hanging in the rendezvous or stacking ranks on one device."""
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("UNREAL_FORCE_DEVICE", "RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"UNREAL_FORCE_DEVICE" in r.stderr and not r.stdout.strip()
