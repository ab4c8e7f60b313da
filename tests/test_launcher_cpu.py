"""The self-launcher behind `python bench.py --gpus N` (unreal_amd.parallel.launch_ranks), on CPU over gloo: it starts
N fresh rank processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's stdout only (one JSON line
for the caller), and returns the worst exit code -- a dead rank must not leave the others hanging in a collective."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import torch
    from unreal_amd import parallel
    rank, local_rank, world = parallel.init_distributed(backend="gloo")
    assert world == int(sys.argv[1]) and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if len(sys.argv) > 2 and rank == int(sys.argv[2]):
        sys.exit(7)                                   # this rank dies before the collective
    flat = torch.full((1000,), float(rank + 1))
    parallel.all_reduce_sum(flat)
    ok = parallel.all_true(rank != 99, "cpu")
    mx = parallel.max_over_ranks(float(rank), "cpu")
    parallel.barrier()
    print(json.dumps({"rank": rank, "sum": float(flat[0]), "all_true": ok, "max": mx, "n_gpus": world}), flush=True)
    parallel.shutdown()
""") % ROOT

LAUNCH = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    from unreal_amd import parallel
    sys.exit(parallel.launch_ranks(sys.argv[1], sys.argv[2:], int(sys.argv[2]), grace_s=3.0))
""") % ROOT


def _run(tmp_path, *args):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    launcher = tmp_path / "launch.py"
    launcher.write_text(LAUNCH)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(launcher), str(script)] + list(args), env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=240)


def test_launcher_relays_rank0_and_reduces(tmp_path):
    r = _run(tmp_path, "3")
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()                 # ranks 1.. print to stderr
    out = json.loads(lines[0])
    assert out == {"rank": 0, "sum": 6.0, "all_true": True, "max": 2.0, "n_gpus": 3}
    others = [json.loads(l) for l in r.stderr.decode().splitlines() if l.startswith("{")]
    assert sorted(o["rank"] for o in others) == [1, 2] and all(o["sum"] == 6.0 for o in others)


def test_launcher_returns_the_failure_and_reaps_the_survivors(tmp_path):
    r = _run(tmp_path, "2", "1")                              # rank 1 exits with code 7 before the all-reduce
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


def test_launcher_terminated_by_signal_leaves_no_rank_behind(tmp_path):
    """SIGTERM to the launcher (an outer `timeout`) must take the ranks down with it: a rank stuck in a collective would
    otherwise hold its GPU for ever.  The ranks here sleep; each writes its PID first."""
    import signal
    import time
    rank_script = tmp_path / "sleeper.py"
    rank_script.write_text(textwrap.dedent("""
        import os, sys, time
        open(os.path.join(%r, "pid.%%s" %% os.environ["RANK"]), "w").write(str(os.getpid()))
        time.sleep(600)
    """) % str(tmp_path))
    launcher = tmp_path / "launch.py"
    launcher.write_text(LAUNCH)
    p = subprocess.Popen([sys.executable, str(launcher), str(rank_script), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    deadline = time.time() + 60
    while time.time() < deadline and not all((tmp_path / ("pid.%d" % r)).exists() for r in (0, 1)):
        time.sleep(0.1)
    pids = [int((tmp_path / ("pid.%d" % r)).read_text()) for r in (0, 1)]
    p.send_signal(signal.SIGTERM)
    assert p.wait(60) == 128 + signal.SIGTERM
    time.sleep(0.2)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = open("/proc/%d/stat" % pid).read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, "rank process %d survived its launcher" % pid
