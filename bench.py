#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of full-UNREAL `Trainer.process()` on maze 84x84 (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under an outer `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE already set: this process IS one rank), or typed as is -- then this process is
only a launcher: before anything touches the GPU it starts N fresh rank processes of itself
(unreal_amd.parallel.launch_ranks), relays rank 0's JSON line and exits with the worst child code.

A "step" is one `Trainer.process()` call: every actor of the rank is advanced by n_step_TD = 20
environment steps (HIP env kernel writing straight into the HBM replay ring), then one UNREAL update
(base A3C + pixel control + value replay + reward prediction, hand-written forward/backward kernels,
fused clip + RMSProp; gradient all-reduce over RCCL when N > 1).  Workload = BASELINE.json configs[1]:
4096 batched actors per MI355X, replay history 2000 frames per actor (173 GB uint8 ring), fp32 compute.
value = env steps taken by ALL ranks inside the K timed calls / max-over-ranks wall time.  The replay
fill (2000 policy steps per actor, no learning -- the reference's global_t does not advance there
either, trainer.py:446-448) and W warm-up calls are untimed.

Extra objects on the JSON line: `roofline` for the dominant kernel (HIP-event timed inside the timed
region) and `cpu_baseline` (the oracle's threaded restatement of the reference loop on host cores).
"""
import argparse
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per launch unit (DESIGN.md section "Kernels"); MACs per frame
ENC_FWD_MAC = 400 * 192 * 16 + 81 * 256 * 32                     # conv1 + conv2 forward
ENC_BWD_MAC = 81 * 256 * 32 * 2 + 400 * 192 * 16                 # conv2 wgrad + dgrad, conv1 wgrad
# encoder_bwd runs every product on the 16-bit matrix cores with split operands.  Round 3: fp16 hi + lo planes -- conv2
# wgrad / dgrad take 3 fp16 passes per fp32 product, conv1 wgrad 2 (the uint8 pixel is exact in one term).  (Round 2:
# three bf16 terms, 6 / 6 / 3 passes, blended ceiling 549 TF.)  Its ceiling is the dense fp16 MFMA peak over those passes:
# BLENDED fp32-equivalent peak = algorithmic MACs / (16-bit pass-MACs / 2500 TF)
ENC_BWD_PASSES = (3, 3, 2)                                       # conv2 wgrad, conv2 dgrad, conv1 wgrad
ENC_BWD_BF16_PASS_MAC = 81 * 256 * 32 * (ENC_BWD_PASSES[0] + ENC_BWD_PASSES[1]) + 400 * 192 * 16 * ENC_BWD_PASSES[2]
ENC_BWD_R2_PASS_MAC = 81 * 256 * 32 * 2 * 6 + 400 * 192 * 16 * 3 # last round's pass count (for the comparable fraction)
BF16_MFMA_PEAK_TFLOPS = 2500.0                                   # MI355X_MICROARCH.md, Peak BF16 / FP16 MFMA (dense)
ENC_BWD_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS * ENC_BWD_MAC / ENC_BWD_BF16_PASS_MAC      # ~992 fp32-equivalent TFLOP/s
ENC_BWD_R2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS * ENC_BWD_MAC / ENC_BWD_R2_PASS_MAC     # ~549
FP32_MFMA_PEAK_TFLOPS = 157.3                                    # MI355X_MICROARCH.md, Peak FP32 (matrix)
HBM_PEAK_GBS = 8000.0
# What the chip sustains (measured on this pool, DESIGN.md section 9 / profiles/r02_kernel_roofline.md): a pure streaming
# kernel reaches ~6.2 TB/s, and dense 16-bit MFMA runs at ~1.6 - 2.1 GHz instead of the 2.4 GHz the 2.5 PF figure assumes
# (power-limited).  Reported beside the nominal peaks as roofline.peak_sustained; `frac` stays against the nominal ones.
HBM_SUSTAINED_GBS = 6200.0
MFMA_SUSTAINED_CLOCK_FRACTION = 1.85 / 2.4
PMC_FILE = "r04_pmc_bench.json"                                  # in-situ rocprofv3 --pmc passes of THIS program


def build_trainer(args, rank, world, device):
    from unreal_amd.environment.environment import Environment
    from unreal_amd.model.model import UnrealModel
    from unreal_amd.options import get_options
    from unreal_amd.train.rmsprop_applier import RMSPropApplier
    from unreal_amd.train.trainer import Trainer, log_uniform
    from unreal_amd import parallel
    flags = get_options("training", preset="lab", argv=["--env_type", "maze", "--env_name", ""])
    Environment.action_size = -1
    A = Environment.get_action_size("maze", "")
    net = UnrealModel(A, 0, -1, flags.use_lstm, flags.use_pixel_change, flags.use_value_replay,
                      flags.use_reward_prediction, flags.pixel_change_lambda, flags.entropy_beta, device, seed=1)
    lr0 = log_uniform(flags.initial_alpha_low, flags.initial_alpha_high, flags.initial_alpha_log_rate)
    applier = RMSPropApplier(None, decay=flags.rmsp_alpha, momentum=0.0, epsilon=flags.rmsp_epsilon,
                             clip_norm=flags.grad_norm_clip, device=device)
    tr = Trainer(0 if rank == 0 else rank, net, lr0, None, applier, "maze", "", flags.use_lstm,
                 flags.use_pixel_change, flags.use_value_replay, flags.use_reward_prediction,
                 flags.pixel_change_lambda, flags.entropy_beta, flags.local_t_max, flags.n_step_TD, flags.gamma,
                 flags.gamma_pc, args.history, flags.max_time_step, device, batch_size=args.actors,
                 world_size=world, rank=rank, seed=0xA3C,
                 grad_sync=parallel.all_reduce_sum if world > 1 else None, groups=getattr(args, "groups", 1))
    tr.prepare()
    return flags, net, tr


def host_cpus():
    """CPUs this process may use: the affinity mask, cut to the cgroup's CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) // int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(legs, history, extra_history=100):
    """The oracle's restatement of the reference's threaded loop (main.py:72-162 + trainer.py:438-636):
    `threads` Python threads, each one actor with batch-1 forwards, numpy maze, deque-like replay, its own
    gradient and a hogwild RMSProp step on shared parameters; PyTorch-CPU fp32, one intra-op thread.
    `legs` = [(threads, seconds)]: timed one after the other on ONE trainer (parallel_size = 8, also 1 and
    nproc).  The first 8 actors carry the workload's replay history; actors beyond them (the nproc leg) a short one
    (`extra_history`) so the untimed fill of hundreds of actors stays bounded -- replay length does not enter the cost
    of a step.  -> [(threads, env-steps/s, env-steps, seconds)]"""
    from oracle.trainer import OracleTrainer
    torch.set_num_threads(1)
    cfg = dict(action_size=4, use_lstm=True, use_pixel_change=True, use_value_replay=True,
               use_reward_prediction=True, pixel_change_lambda=0.05, entropy_beta=0.001, local_t_max=20,
               n_step_TD=20, gamma=0.99, gamma_pc=0.9, experience_history_size=history,
               max_time_step=int(13.2e6), rmsp_alpha=0.99, rmsp_epsilon=0.1, grad_norm_clip=40.0,
               initial_alpha_low=1e-4, initial_alpha_high=5e-3, initial_alpha_log_rate=0.5)
    n_max = max(t for t, _ in legs)
    tr = OracleTrainer(cfg, n_actors=min(8, n_max), seed=1)
    if n_max > 8:                                      # the extra actors of the nproc leg: same network, short replay
        from oracle.trainer import OracleActor
        cfg2 = dict(cfg, experience_history_size=extra_history)
        tr.actors += [OracleActor(cfg2, tr.actors[0].draws, tr.dtype) for _ in range(n_max - 8)]

    def fill(i):                                       # replay fill, untimed (weights frozen: actors are independent)
        a = tr.actors[i]
        while not a.exp.is_full():
            a.fill_step(tr.params)

    ths = [threading.Thread(target=fill, args=(i,)) for i in range(n_max)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    out = []
    for threads, seconds in legs:
        steps = [0] * threads
        stop = time.time() + seconds
        t0 = time.time()

        def work(i):
            while time.time() < stop:
                d, _, _ = tr.process_async(i, sum(steps))
                steps[i] += d

        ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        el = time.time() - t0
        out.append((threads, sum(steps) / el, sum(steps), el))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # 100 x 33 ms: a timed region long enough for a 5-s utilisation
    ap.add_argument("--warmup", type=int, default=5)        # sampler to see (round 1: 10 steps = 0.4 s of GPU work)
    ap.add_argument("--actors", type=int, default=4096, help="actors per GPU (BASELINE.json configs[1])")
    ap.add_argument("--history", type=int, default=2000, help="experience_history_size per actor")
    ap.add_argument("--groups", type=int, default=1, help="sequential updates per process() call (actors are dealt into "
                    "this many groups; 1 = one update from all actors, the headline configuration)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="total timed CPU-baseline budget: half of it for the "
                    "parallel_size leg, a quarter each for the 1-thread and the nproc-thread legs")
    ap.add_argument("--cpu-threads", type=int, default=8, help="parallel_size of the reference (options.py:37)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-kernel", default="unreal_encoder_bwd")
    ap.add_argument("--progress", action="store_true", help="phase markers on stderr (diagnosing a run under a profiler)")
    args = ap.parse_args()

    from unreal_amd import parallel
    n_dev = parallel.visible_gpus()                    # from sysfs: the launcher below never initialises HIP / HSA
    if args.gpus > 1 and "UNREAL_FORCE_DEVICE" not in os.environ and n_dev is not None and n_dev < args.gpus:
        raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s).  One rank per GPU over RCCL needs %d devices; "
                         "to rehearse several ranks on one device set UNREAL_FORCE_DEVICE=0 (gloo, not a measurement)."
                         % (args.gpus, n_dev, args.gpus))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launcher only: nothing in this process has touched (or will touch) the GPU
        sys.exit(parallel.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    from unreal_amd import ops
    if args.gpus > 1 and "UNREAL_FORCE_DEVICE" not in os.environ and torch.cuda.device_count() < args.gpus:
        # the authoritative check, in the RANK process (the sysfs count above sees the host's topology, which a container
        # may show in full while exposing one device)
        raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s).  One rank per GPU over RCCL needs %d devices; "
                         "to rehearse several ranks on one device set UNREAL_FORCE_DEVICE=0 (gloo, not a measurement)."
                         % (args.gpus, torch.cuda.device_count(), args.gpus))
    rank, local_rank, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    dev_index = parallel.device_index(local_rank)      # UNREAL_FORCE_DEVICE: several ranks rehearse on one GPU
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    def mark(msg):
        if args.progress:
            torch.cuda.synchronize()
            print("[bench %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    mark("device ready")
    flags, net, tr = build_trainer(args, rank, world, device)
    if os.environ.get("UNREAL_SAVE_MAPS"):             # tools/pmc_bench.sh: lets a crash under the profiler be symbolised
        try:
            open(os.environ["UNREAL_SAVE_MAPS"], "w").write(open("/proc/self/maps").read())
        except OSError:
            pass
    T = flags.n_step_TD
    mark("trainer built")

    t_fill = time.time()
    k = 0
    while not tr._full:                      # replay warm-up: untimed, global_t frozen
        tr.process(None, 0)
        k += 1
        # (Trainer._fill_experience bounds the un-synchronised dispatch queue itself: one stream sync every 64 calls)
        if k in (1, 2, 10, 100, 1000):
            mark("fill call %d" % k)
    torch.cuda.synchronize()
    t_fill = time.time() - t_fill
    mark("replay full")

    global_t = 0
    for _ in range(args.warmup):
        tr.process(None, global_t, sync_stats=False)
        global_t += args.actors * T * world
    tr.read_stats()
    mark("warm-up done")
    ops.kernel_timer_start(args.timed_kernel)
    tr.time_grad_sync = world > 1                     # HIP events around the gradient exchange of every timed update
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.process(None, global_t, sync_stats=False)
        global_t += args.actors * T * world
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    kt = ops.kernel_timer_stop()
    comm_ms = tr.grad_sync_ms()                       # [ms per update] of this rank (empty at world == 1)
    steps_local, episodes, score_sum = tr.read_stats()
    elapsed = parallel.max_over_ranks(elapsed, device)
    tot_steps, tot_eps, tot_score = parallel.sum_over_ranks([steps_local, episodes, score_sum], device)
    losses = tr._publish_losses()
    per_rank_steps = parallel.gather_over_ranks(steps_local, device)
    per_rank_comm = parallel.gather_over_ranks(sum(comm_ms) / len(comm_ms) if comm_ms else 0.0, device)

    backend = parallel.backend_name()
    ranks_formed = parallel.world_size()
    if rank != 0:
        parallel.shutdown()
        return
    value = tot_steps / elapsed
    frames_per_launch = kt["units"] / max(kt["launches"], 1)
    mac = ENC_BWD_MAC if "bwd" in args.timed_kernel else ENC_FWD_MAC
    avg_ms = kt["ms"] / max(kt["launches"], 1)
    achieved = (2.0 * mac * frames_per_launch) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    # HBM traffic per launch: PMC counters cannot be read from inside this process; the figure comes from the committed
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS program (tools/pmc_bench.sh -> profiles/r04_pmc_bench.json,
    # corrected as MI355X_MICROARCH.md prescribes: FETCH doubled), mean over the launches of its timed calls.
    # The PMC file records the launch it measured (frames per launch, actors, groups); the figure is scaled to THIS run's
    # frames per launch when the schedule is the profiled one (same groups: same mix of launches) and null otherwise.
    traffic, traffic_note = None, "no PMC summary for this kernel"
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        base = args.timed_kernel.replace("unreal_", "")
        k = next(doc["kernels"][n] for n in (base + "_kernel", base + "_roles_kernel") if n in doc["kernels"])
        prof = doc.get("profiled_run", {"actors": 4096, "groups": 1, "frames_per_launch": 86016.0})
        if prof.get("groups", 1) == args.groups and frames_per_launch > 0:
            traffic = k["hbm_bytes_per_launch"] * frames_per_launch / float(prof["frames_per_launch"])
            traffic_note = "HBM bytes per launch: in-situ rocprofv3 --pmc passes of bench.py (profiles/%s, measured at " \
                           "%d frames per launch, scaled by frames to this run's %d)" % (
                               PMC_FILE, prof["frames_per_launch"], frames_per_launch)
        else:
            traffic_note = "profiles/%s was measured with groups=%s: not comparable with this run" % (PMC_FILE, prof.get("groups"))
    except Exception:
        pass
    peak = ENC_BWD_PEAK_TFLOPS if "bwd" in args.timed_kernel else FP32_MFMA_PEAK_TFLOPS
    bytes_per_frame = 57136.0 if "bwd" in args.timed_kernel else 57168.0      # DESIGN.md section 4 (fwd: frame in, f2 + c1 out)
    alg_bytes = bytes_per_frame * frames_per_launch
    ai = 2.0 * mac / bytes_per_frame
    ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    hbm_bound = ai < ridge
    out = {
        "metric": "env-steps/sec (whole node), UNREAL maze 84x84",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (fp16x2-split MFMA operands, fp32 accumulate)", "data": "synthetic (device maze environments, random-init weights)",
        "config": {"workload": "maze_environment full UNREAL (PC+RP+VR), %d batched actors per MI355X, "
                               "n_step_TD=%d, replay history %d/actor (uint8 HBM ring)" % (args.actors, T, args.history),
                   "actors_per_gpu": args.actors, "global_actors": args.actors * world,
                   "updates_per_call": args.groups,
                   "env_steps_per_call": tot_steps / args.steps, "parallelism": "actors sharded x%d, flat-gradient "
                   "all-reduce (%s)" % (world, backend) if world > 1 else "single GPU", "replay_fill_s": t_fill,
                   # what the exchange actually ran on (the driver, not the builder, runs N > 1 on real GPUs)
                   "backend": backend, "rccl_ranks": ranks_formed if backend == "nccl" else 0, "ranks": ranks_formed,
                   "devices_visible": torch.cuda.device_count(),
                   "env_steps_per_rank": [int(x) for x in per_rank_steps],
                   "grad_message_bytes": int(net.grads.flat.numel()) * 4,
                   "comm_ms_per_update": (sum(per_rank_comm) / len(per_rank_comm)) if world > 1 else 0.0,
                   "comm_ms_per_update_per_rank": [round(x, 4) for x in per_rank_comm] if world > 1 else [],
                   "total_loss": losses["total_loss"], "grad_norm": losses["grad_norm"]},
        # Which roof binds the dominant kernel: its arithmetic intensity (fp32-equivalent FLOP per algorithmic HBM byte)
        # against the ridge of ITS ceilings (fp32-equivalent MFMA ceiling / 8 TB/s).  Since round 3 runs encoder_bwd's
        # products in 3 / 3 / 2 fp16 passes (ceiling 992 TF) the kernel sits LEFT of the ridge (89 < 124 FLOP/B): HBM binds.
        # Both fractions are reported; `bound` / `achieved` / `peak` / `frac` are the binding roof's.
        "roofline": dict(
            {"kernel": args.timed_kernel},
            **({"bound": "hbm", "achieved": alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": (alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms > 0 else 0.0}
               if hbm_bound else
               {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak}),
            **{"arithmetic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
               "mfma_achieved_tflops": achieved, "mfma_peak_tflops": peak, "mfma_frac": achieved / peak,
               "hbm_achieved_gbs": alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0, "hbm_peak_gbs": HBM_PEAK_GBS,
               "hbm_frac": (alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms > 0 else 0.0,
               "peak_note": "mfma_peak = fp32-equivalent: dense fp16 MFMA peak (2500 TF) over the kernel's 16-bit passes "
                            "(conv2 wgrad + dgrad x3, conv1 wgrad x2 since round 3: fp16 hi + lo planes).  The same achieved "
                            "rate against round 2's ceiling (6 / 6 / 3 bf16 passes, 549 TF) is %.3f; against the fp32 MFMA "
                            "peak (157.3) %.3f.  hbm: algorithmic bytes (57,136 B per frame: uint8 frame + saved conv1 "
                            "activation + d_f2) over the HIP-event launch time, against 8 TB/s"
                            % (achieved / ENC_BWD_R2_PEAK_TFLOPS, achieved / FP32_MFMA_PEAK_TFLOPS),
               "frac_vs_round2_ceiling": achieved / ENC_BWD_R2_PEAK_TFLOPS,
               "peak_sustained": {"hbm_gbs": HBM_SUSTAINED_GBS, "mfma_tflops": peak * MFMA_SUSTAINED_CLOCK_FRACTION,
                                  "hbm_frac": (alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_SUSTAINED_GBS) if avg_ms > 0 else 0.0,
                                  "mfma_frac": achieved / (peak * MFMA_SUSTAINED_CLOCK_FRACTION),
                                  "note": "streaming-kernel HBM rate and the MFMA peak at the ~1.85 GHz the chip holds "
                                          "under dense 16-bit MFMA (measured, DESIGN.md section 9); not used for `frac`"},
               "traffic": traffic, "traffic_unit": traffic_note,
               "algorithmic_bytes_per_launch": alg_bytes,
               "launches": kt["launches"], "avg_launch_ms": avg_ms,
               "frames_per_launch": frames_per_launch, "share_of_step": kt["ms"] / (elapsed * 1e3),
               "whole_path_frac_fp32_mfma": value * 69.67e6 / (FP32_MFMA_PEAK_TFLOPS * 1e12 * world),
               "whole_path_frac_hbm_u8": value * 114396.0 / (HBM_PEAK_GBS * 1e9 * world)}),
    }
    if world == 1 and not args.no_cpu_baseline:
        hist_cpu = args.history          # the workload's own replay history (its fill is untimed, like the GPU's)
        nproc = host_cpus()
        legs = [(1, args.cpu_seconds / 4.0), (args.cpu_threads, args.cpu_seconds / 2.0)]
        if nproc not in (1, args.cpu_threads):
            legs.append((nproc, args.cpu_seconds / 4.0))
        res = cpu_baseline(legs, hist_cpu)
        main_leg = [r for r in res if r[0] == args.cpu_threads][0]
        desc = lambda r: "%d env-steps in %.1f s on %d Python thread%s" % (r[2], r[3], r[0], "" if r[0] == 1 else "s")
        out["cpu_baseline"] = {"value": main_leg[1], "unit": "env-steps/s", "cores": args.cpu_threads, "kind": "port",
                               "host_cores": nproc, "host_cores_online": os.cpu_count(),
                               "by_threads": [{"threads": r[0], "value": r[1], "env_steps": r[2], "seconds": r[3]} for r in res],
                               "sample": "%s (value; %s): each thread = one actor running full-UNREAL process() of "
                                         "oracle/trainer.py on PyTorch-CPU fp32, 1 intra-op thread each, shared hogwild "
                                         "RMSProp; replay history %d per thread for the first 8 actors, 100 for the extra "
                                         "actors of the nproc leg; fills untimed" % (
                                             desc(main_leg), "; ".join(desc(r) for r in res if r is not main_leg), hist_cpu)}
    side = os.environ.get("UNREAL_BENCH_SIDECAR")      # tools/pmc_bench.sh: what the counter passes measured
    if side:
        json.dump({"actors": args.actors, "groups": args.groups, "steps": args.steps, "warmup": args.warmup,
                   "history": args.history, "timed_kernel": args.timed_kernel, "frames_per_launch": frames_per_launch,
                   "launches_timed": kt["launches"], "ms_per_step": elapsed / args.steps * 1e3}, open(side, "w"))
    print(json.dumps(out), flush=True)
    parallel.shutdown()


if __name__ == "__main__":
    main()
