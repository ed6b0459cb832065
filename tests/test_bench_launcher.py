"""bench.py as the driver starts it: `python bench.py --gpus N` must run N rank processes (not one), rendezvous them on
127.0.0.1, take the barrier / max-over-ranks reduce on the host control plane (gloo) and hand back ONE JSON line from rank
0 -- and a rank that dies must show up in the exit code.  `--dry-run` exercises exactly that path without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, timeout=240, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, BENCH, *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    p = _run("--gpus", "2", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert out["max_over_ranks"] == 2.0            # the all-reduce(MAX) really saw rank 1's contribution


def test_a_slow_barrier_is_not_charged_to_the_timed_region():
    """The region of `value` ends when this rank's work has returned and its device is idle; the trailing barrier (gloo, the
    control plane) and the MAX over ranks come after it.  Stand-in work of 20 ms, a barrier made 60 ms slow on purpose."""
    for world in ("1", "2"):
        p = _run("--gpus", world, "--dry-run", VBA_BENCH_BARRIER_SLEEP_MS="60")
        assert p.returncode == 0, p.stderr[-2000:]
        ms = json.loads(p.stdout.strip())["timed_region_ms"]
        assert 19.0 <= ms < 70.0, ms          # (charged, the trailing barrier alone would add 60)


def test_single_rank_needs_no_process_group():
    p = _run("--gpus", "1", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip())["n_gpus"] == 1


def test_a_dead_rank_fails_the_run():
    p = _run("--gpus", "3", "--dry-run", "--dry-run-fail-rank", "1", "--rank-timeout", "120")
    assert p.returncode != 0
    assert "rank exit codes" in p.stderr


def test_under_torchrun_the_environment_decides_the_rank():
    # what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` gives each process: no launcher then
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert json.loads(outs[0][0].strip())["n_gpus"] == 2
    assert outs[1][0].strip() == ""                # only rank 0 prints


def test_cpu_baseline_worker_protocol():
    # one worker of the all-cores CPU baseline on the smallest window: ready -> go -> {"calls", "seconds"}
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    p = subprocess.Popen([sys.executable, BENCH, "--cpu-worker", "3", "--cpu-seconds", "0.2", "--config", "C1"], env=env,
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "ready"
    p.stdin.write("go\n")
    p.stdin.flush()
    res = json.loads(p.stdout.readline())
    p.stdin.close()
    assert p.wait(timeout=60) == 0
    assert res["calls"] >= 20 and res["calls"] % 20 == 0 and res["seconds"] > 0
