"""hipGraph A/B for grouped updates -- TIMING ONLY.

One group's actor-learner pass (`compute_gradients` + clip/RMSProp + stats, ~300 dependent launches of B/G rows) is
captured once per group into a hipGraph (stream capture through torch.cuda.CUDAGraph: every launch of this library goes
to torch's current raw stream, so the capture sees all of them) and a `process()` call becomes G graph launches.

What the captured graph freezes: the Philox stream ids of the call's draws, the learning rate and the absmax slot
indices are host scalars baked into the kernel arguments, so every replay repeats the SAME draws -- the replayed job is
not the product's job.  The kernels, shapes, dependencies and byte counts are the product's, which is all a launch-path
A/B needs; adopting graphs would need those scalars moved to device words first (which is why this measures before
building that).

usage: python tools/exp/graph_ab.py [--actors 4096] [--groups 8 64] [--calls 5]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def one_group_pass(tr, g, lr):
    net = tr.local_network
    tr._select_group(g)
    tr.compute_gradients()
    tr.last_grad_norm = tr.grad_applier.step(net.params.flat, net.grads.flat, lr)
    net.mark_params_changed()
    from unreal_amd import ops
    ops.rollout_stats(tr.Bg, tr.n_steps, tr.ring.score_valid, tr.ring.score_out, tr.stats)


def timed(fn, calls):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls * 1e3


def run(actors, history, G, calls):
    args = argparse.Namespace(actors=actors, history=history, groups=G)
    flags, net, tr = bench.build_trainer(args, 0, 1, torch.device("cuda:0"))
    while not tr._full:
        tr.process(None, 0)
    torch.cuda.synchronize()
    lr = tr._anneal_learning_rate(0)
    for _ in range(2):
        tr.process(None, 0)
    eager = timed(lambda: tr.process(None, 0, sync_stats=False), calls)

    # host time of the eager launch path alone (no device wait): how long the host needs to ENQUEUE one call
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.process(None, 0, sync_stats=False)
    enqueue = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()

    graphs = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):               # warm every lazily allocated buffer outside the capture
        for g in range(G):
            one_group_pass(tr, g, lr)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    pool = None
    for g in range(G):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, pool=pool):
            one_group_pass(tr, g, lr)
        pool = gr.pool()
        graphs.append(gr)
    torch.cuda.synchronize()

    def replay():
        for gr in graphs:
            gr.replay()

    for _ in range(2):
        replay()
    graph = timed(replay, calls)
    env_steps = actors * tr.n_step_TD
    return dict(actors=actors, groups=G, rows_per_launch=actors // G, eager_ms=round(eager, 3),
                eager_enqueue_ms=round(enqueue, 3), graph_ms=round(graph, 3),
                eager_Msteps=round(env_steps / eager / 1e3, 3), graph_Msteps=round(env_steps / graph / 1e3, 3),
                speedup=round(eager / graph, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--actors", type=int, default=4096)
    ap.add_argument("--history", type=int, default=500, help="replay length: enters the fill time only, not the launches")
    ap.add_argument("--groups", type=int, nargs="+", default=[8, 64])
    ap.add_argument("--calls", type=int, default=5)
    a = ap.parse_args()
    for G in a.groups:
        r = run(a.actors, a.history, G, a.calls if G <= 8 else max(2, a.calls // 2))
        print(json.dumps(r), flush=True)
        import gc
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
