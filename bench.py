#!/usr/bin/env python3
"""Benchmark of the BA hot path on MI355X: BA iterations/sec on the 500-pose / 50 000-observation window.

    python bench.py --gpus N --steps K --warmup W

One "step" is one call of BA() (reference estimation/BA/BA_filtering.py:4-98) through the C ABI with every
input already resident in HBM.  The steps walk the reference driver's own schedule (od_pipe.py:1036-1040):
20 calls per window -- 10 landmark-only (initialize=True) then 10 full -- restarting from the perturbed
initial guess after each 20, so the mix of kernels is the one the reference's 20-iteration loop executes.
The 20 calls of a schedule are issued with one host call (vba_run_schedule) and chained on the device.

Ranks.  N = 1: this process measures.  N > 1: one process per GPU.  Either the driver starts them
(`python -m torch.distributed.run ... bench.py --gpus N`: RANK / WORLD_SIZE / LOCAL_RANK in the environment), or --
when bench.py is started plainly with --gpus N > 1 -- THIS process becomes a launcher: before anything touches a
GPU it starts N fresh rank processes (python bench.py, RANK = 0..N-1, rendezvous on 127.0.0.1), relays rank 0's
JSON line and exits with the worst rank's code.  The control plane (barriers, max-over-ranks time) is gloo on the
host, so the replica measurement needs no RCCL and can be rehearsed with --gpus 2 on a one-GPU box (ranks map to
LOCAL_RANK % device_count).  Every rank runs its own 500/50k window (the reference's outer loop over sequences,
od_pipe.py:1063-1086): weak scaling, no data-path collective, value = N windows' iterations / max-over-ranks time.
The observation-sharded mode (rows of ONE window split over the ranks, three all-gathers per call over RCCL) is
measured in the same run on a separate nccl group and reported under "sharded" with the number of ranks RCCL saw; it
is skipped, and says so, when two ranks share a device.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel class of the timed run, HIP-event
timings on the library's stream; "whole_call" = SURVEY's algorithmic bytes of a call / ms_per_step; "mfma" =
matrix-core counters of the solve kernels from the committed PMC pass), "cpu_baseline" (the NumPy oracle on ALL host
cores of this box, one single-threaded worker process per core over independent windows, bounded sample),
"python_BA_call" (the drop-in vinsat_amd.ba.BA in the reference's loop shape), "host_roundtrip" (vba_iterate),
"batched" (4096 windows per launch: the HBM-bound regime), "batched_sweep" (W = 1 .. 4096 windows per handle, kernel set and
solver chosen by the handle), "python_BA_batch" (22 windows through BA / BA_window on lists), "configs" (C2, C4, C5 and the
two-pass window with either integrator), "chain_classes_ms" / "phase_ms" (class and phase times of the chained schedule).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# fp64 matrix peak: the guide's MFMA table has no f64 row; the vendor figure for MI355X is 78.6 TFLOP/s (dense fp64
# matrix = fp64 vector rate), = 256 CUs x 4 SIMDs x 32 FMA/clk x 2.4 GHz
MFMA_F64_PEAK_TFLOPS = 78.6
METRIC = "BA iterations/sec (500 poses, 50k landmarks) + final pose RMSE vs ref"

# Algorithmic bytes per unit, per kernel class (SURVEY.md section 8d itemisation):
#   per observation: inputs 56 B (xyz 24 + uv 16 + conf 8 + ii 8), |r| 16 B, weight 8 B
#   "residual" runs only on a call whose states were replaced by the host; otherwise the previous call's trial
#   kernel has left the keys behind (trial = 56 in + 8 weight + 16 keys out)
#   per pose: see DESIGN.md table
ALG_BYTES = {
    "residual": lambda n, m: 72 * m,
    "select": lambda n, m: 16 * m,
    "accumulate": lambda n, m: 64 * m + 216 * n,
    "dynamics": lambda n, m: (80 + 32 + 8 + 288 + 48 + 48 + 8 + 24 + 216) * n,
    "assemble": lambda n, m: (216 + 2 * 288 + 96 + 24 + 216 + 1944 + 72) * n,
    # landmark-only call of the batched mode: the fused assembly + 6x6 solve reads the per-pose sums and the states and
    # writes the step and the trial states; the diagonal blocks are not written (a later trial forms them)
    "assemble_init": lambda n, m: (216 + 80 + 72 + 80) * n,
    "solve": lambda n, m: (1944 + 72 + 2 * 1440 + 144 + 160) * n,
    # batched mode since round 3: the walk forms its blocks from the per-pose inputs itself (792 B per pose) -- no assembly
    # launch, the bands never go through memory; X / z of the forward sweep still make their round trip
    "solve_forming": lambda n, m: (792 + 2 * 1440 + 144 + 160) * n,
    "trial": lambda n, m: 80 * m + (80 + 32 + 8) * n,
    # the accept test reads a few hundred block partials and writes the window's scalars; "begin" has no kernel at all (two
    # back-to-back event records: the state ping-pong removed the commit copy) -- neither moves per-pose data any more
    "decide": lambda n, m: 8 * ((m + 255) // 256 + (n + 31) // 32) + 512,
    "begin": lambda n, m: 0,
}


def survey_bytes_per_call(n, m):
    """B_alg = 208 m + 5000 n for a one-trial call."""
    return 208.0 * m + 5000.0 * n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--windows", type=int, default=4096, help="windows per launch of the batched series (0 = skip)")
    ap.add_argument("--batched-steps", type=int, default=40)
    ap.add_argument("--sweep", default="1,2,4,8,15,16,22,32,64,128,256,1024", help="window counts of the batched_sweep series (the "
                    "--windows handle joins it); empty = skip")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config series (C2, C4, C5, gap window)")
    ap.add_argument("--profile-tag", default="r05", help="profiles/<tag>_w1_traffic.json / _mfma.json supply roofline.traffic / .mfma")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the CPU baseline (0 = every core of this box)")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--strict-sharded", action="store_true", help="a failed or hung sharded leg fails the run (exit code 3); default: the line "
                    "carries sharded.error, the headline measurement (replicas) and the exit code are not affected")
    ap.add_argument("--no-driver", action="store_true", help="skip the end-to-end series of the drop-in driver (streaming_version, run_folder)")
    ap.add_argument("--no-schur", action="store_true", help="skip the free-landmark Schur add-on leg (parity unpinned, not part of the metric)")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="launcher: seconds before hung rank processes are ended")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)     # internal: CPU baseline worker
    ap.add_argument("--dry-run", action="store_true", help="launcher / control-plane self-test without a GPU: ranks rendezvous "
                    "over gloo, take the barrier and the max-reduce, rank 0 prints a line marked dry_run (not a measurement)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)    # self-test: this rank exits 7
    return ap.parse_args()


def schedule(k):
    j = k % 20
    return j, j < 10


def run_steps(eng, st0, nsteps, windows=1):
    """nsteps BA() calls walking the 20-call schedule; each schedule is issued through vba_run_schedule, i.e. the
    driver's loop `for iter in range(20): BA(iter, ...)` (od_pipe.py:1036-1040) as one host call whose calls are
    chained on the device (bit-identical to call-by-call stepping, tests/test_gpu_parity.py)."""
    k = 0
    while k < nsteps:
        cnt = min(20, nsteps - k)
        eng.set_states(st0, 1e-4, window=-1 if windows > 1 else 0)
        calls = [schedule(j) for j in range(cnt)]
        eng.run_schedule([c[0] for c in calls], [c[1] for c in calls])
        k += cnt


def load_windows(eng, win, n, W):
    for w in range(W):
        eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
        eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)


def sweep_point(BAEngine, win, st0, n, m, W, device, sync):
    """W independent C3 windows on ONE handle with the handle's own choice of kernel set and solver: the chained 20-call
    schedule, warmed once, then timed."""
    e = BAEngine(n, m, windows=W, device=device)
    load_windows(e, win, n, W)
    run_steps(e, st0, 20, windows=W)
    reps = 3 if W <= 256 else 1
    sync()
    t0 = time.perf_counter()
    run_steps(e, st0, 20 * reps, windows=W)
    sync()
    dt = time.perf_counter() - t0
    lat, chunk = e.mode()
    e.close()
    return {"windows": W, "value": 20 * reps * W / dt, "ms_per_step": 1e3 * dt / (20 * reps),
            "kernel_set": "latency" if lat else "bandwidth", "solver": f"chunks of {chunk} + cyclic reduction" if chunk else "sequential walk, four windows per wavefront",
            "whole_step_frac_survey_bytes": survey_bytes_per_call(n, m) * W / (dt / (20 * reps)) / 1e9 / HBM_PEAK_GBS}


def config_series(BAEngine, od_pipe, synth, device, sync):
    """The other BASELINE configs on one GPU (same function, other sizes: BA_filtering.py:4-98): one warmed chained 20-call
    schedule each, plus the two-pass window whose dynamics factor spans a ~945 s gap with either integrator."""
    out = {}
    for name in ("C2", "C4", "C5"):
        det, orb = synth.make_sequence(name)
        win = od_pipe.prepare_window(det, orb)
        st0 = od_pipe.initial_guess(win)
        n, m = win.time_idx.size, win.ii.size
        e = BAEngine(n, m, device=device)
        load_windows(e, win, n, 1)
        run_steps(e, st0, 20)
        reps = 3
        sync()
        t0 = time.perf_counter()
        run_steps(e, st0, 20 * reps)
        sync()
        ms = 1e3 * (time.perf_counter() - t0) / (20 * reps)
        e.set_chain_profile(True)
        run_steps(e, st0, 20)
        cls = e.chain_profile()
        e.close()
        whole = survey_bytes_per_call(n, m)
        out[name] = {"poses": int(n), "observations": int(m), "value": 1e3 / ms, "unit": "BA iterations/s", "ms_per_step": ms,
                     "whole_call_bytes": whole, "whole_call_frac": whole / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "kernels_per_call": {"landmark_only": 2, "full": 5},
                     "chain_classes_ms": {k: v[0] for k, v in cls.items()}}
    # second batch of the two-pass sequence: 25 poses, one gap of ~945 s -- the RK4 chain of the dynamics factor IS the call
    win = od_pipe.prepare_window(*synth.make_two_pass_sequence())
    gap_states = od_pipe.initial_guess(win)
    n, m = win.time_idx.size, win.ii.size
    gap = {"poses": int(n), "observations": int(m), "longest_gap_s": int(np.diff(win.time_idx).max())}
    for hop in (False, True):
        e = BAEngine(n, m, device=device)
        e.set_integrator(hop)
        load_windows(e, win, n, 1)
        iters, inits = list(range(20)), [False] * 20
        e.set_states(gap_states, 1e-4)
        e.run_schedule(iters, inits)
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            e.set_states(gap_states, 1e-4)
            e.run_schedule(iters, inits)
        sync()
        ms = 1e3 * (time.perf_counter() - t0) / 60
        e.close()
        gap["hop (predict_gpu, the reference's GPU default)" if hop else "rk4 (predict, the parity target)"] = {"value": 1e3 / ms, "ms_per_step": ms}
    out["GAP"] = gap
    # the window of a sequence's LAST batch after six passes: 127 poses / 6 000 rows, ten gaps of 395 .. 1000 s (a knot every 1000 s)
    win = od_pipe.prepare_window(*synth.make_multi_pass_sequence())
    mp_states = od_pipe.initial_guess(win)
    n, m = win.time_idx.size, win.ii.size
    gaps = np.diff(win.time_idx)
    mp = {"poses": int(n), "observations": int(m), "long_gaps": int((gaps > 64).sum()), "longest_gap_s": int(gaps.max())}
    for hop in (False, True):
        e = BAEngine(n, m, device=device)
        e.set_integrator(hop)
        load_windows(e, win, n, 1)
        iters, inits = list(range(20)), [False] * 20
        e.set_states(mp_states, 1e-4)
        e.run_schedule(iters, inits)
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            e.set_states(mp_states, 1e-4)
            e.run_schedule(iters, inits)
        sync()
        ms = 1e3 * (time.perf_counter() - t0) / 60
        e.close()
        mp["hop (predict_gpu, the reference's GPU default)" if hop else "rk4 (predict, the parity target)"] = {"value": 1e3 / ms, "ms_per_step": ms}
    out["SIXTH_PASS"] = mp
    return out


def timed_region(work, barrier, sync, reduce_max):
    """The contract's timed region: barrier + device synchronisation, t0, `work` (synchronous on return), device synchronisation, t1,
    THEN the trailing barrier and the MAX over ranks.  The trailing barrier is the control plane (gloo over the host: 0.1 - 0.5 ms
    with eight local ranks) and must not be charged to a region that is 0.86 ms long at the driver's --steps 20; `work` having
    returned and the device being idle is what ends this rank's time, the MAX over ranks takes care of the slowest."""
    barrier()
    t0 = time.perf_counter()
    work()
    sync()
    dt = time.perf_counter() - t0
    barrier()
    return reduce_max(dt)


def driver_series(od_pipe, synth, errors_eval, device):
    """The drop-in pipeline end to end: wall time of `streaming_version` (reference od_pipe.py:911-1062) on
    the C2, C3 and two-pass sequences and of `errors_eval.run_folder` (the reference's loop over 22 sequence files,
    od_pipe.py:1063-1086) sequentially and batched, split into data preparation / BA calls (uploads and result copies included)
    / bookkeeping (batch cut, dead reckoning across a gap, error records), beside the reference's own wall time for the same
    sequence where a fixture recorded it (tests/golden/*.npz ref_wall_seconds: build container, 8 vCPU)."""
    import shutil
    import tempfile
    out = {"unit": "ms", "sequences": {}}
    for name, fixture in (("C2", "c2"), ("C3", "c3"), ("two-pass", "gap"), ("six-pass", None)):
        det, orb = (synth.make_two_pass_sequence() if name == "two-pass" else synth.make_multi_pass_sequence() if name == "six-pass"
                    else synth.make_sequence(name))
        od_pipe.streaming_version(detections=det.copy(), orbit_np=orb.copy())          # engines created, kernels loaded
        best = None
        for _ in range(3):
            t = {}
            t0 = time.perf_counter()
            od_pipe.streaming_version(detections=det.copy(), orbit_np=orb.copy(), timing=t)
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, t)
        wall, t = best
        e = {"wall": 1e3 * wall, "prep": 1e3 * t["prep"], "ba": 1e3 * t["ba"], "bookkeeping": 1e3 * t["bookkeeping"],
             "ba_calls": int(t["ba_calls"]), "rows": int(det.shape[0]), "host_over_ba": (t["prep"] + t["bookkeeping"]) / t["ba"]}
        gpath = os.path.join(ROOT, "tests", "golden", (fixture or "none") + ".npz")
        if os.path.exists(gpath):
            g = np.load(gpath)
            if "ref_wall_seconds" in g:
                e["reference_wall"] = 1e3 * float(g["ref_wall_seconds"])
                e["reference_threads"] = int(g["ref_threads"]) if "ref_threads" in g else None
        out["sequences"][name] = e
    # 22 synthetic sequences on disk, as the reference's __main__ finds them
    tmp = tempfile.mkdtemp(prefix="vba_bench_folder_")
    try:
        for sub in ("tmp_dets", "tmp_pose"):
            os.makedirs(os.path.join(tmp, sub))
        rows = 0
        for k in range(22):
            det, orb = synth.make_sequence("C3", seed=100 + k)
            rows += det.shape[0]
            np.save(os.path.join(tmp, "tmp_dets", f"{k:05d}_all_detections.npy"), det)
            np.save(os.path.join(tmp, "tmp_pose", f"{k:05d}_orbit_eci_zyxvecs.npy"), orb)
        folder = {"sequences": 22, "rows": int(rows), "window": "C3 (500 poses / 50 000 rows each)"}
        for key, kw in (("batched", dict(batched=True)), ("sequential", dict())):
            errors_eval.run_folder(tmp, **kw)
            st = []
            t0 = time.perf_counter()
            errors_eval.run_folder(tmp, stats=st, **kw)
            wall = time.perf_counter() - t0
            s0 = st[0]
            folder[key] = {"wall": 1e3 * wall, "prep": 1e3 * s0["prep"], "ba": 1e3 * s0["ba"], "bookkeeping": 1e3 * s0["bookkeeping"],
                           "ba_calls": int(s0["ba_calls"]), "ba_iterations_per_s_end_to_end": s0["ba_calls"] / wall,
                           "host_over_ba": (s0["prep"] + s0["bookkeeping"]) / s0["ba"]}
        out["folder"] = folder
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["note"] = ("wall = one call of the driver with its inputs in memory (sequences) / on disk (folder); prep = the per-row part on the device "
                   "(vba_prepare_rows: lat / lon -> ECI, reprojection at ground truth, outlier mask) + NumPy for the per-pose part + the library's "
                   "host helpers for the serial chains (vba_host_*); the sequential folder run prepares the next sequences on host threads "
                   "beside the BA calls; replicas over GPUs and several workers per GPU (errors_eval.run_folder(gpus=N, workers_per_gpu=K)) "
                   "run the preparation of different sequences in parallel")
    return out


# ------------------------------------------------------------------------------------------------ launcher (N > 1)
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Start args.gpus fresh rank processes (this process has not touched, and never touches, a GPU), relay rank 0's
    line, return the worst exit code.  A rank that outlives --rank-timeout is ended by PID and the run fails."""
    n = args.gpus
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = {}

    def drain():
        out["line"] = procs[0].stdout.read()

    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    deadline = time.monotonic() + args.rank_timeout
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        failed = any(c not in (None, 0) for c in codes)
        if failed or time.monotonic() > deadline:
            # a dead rank leaves the others waiting in a collective: give them a moment, then end exactly these PIDs
            grace = time.monotonic() + (10.0 if failed else 0.0)
            while time.monotonic() < grace and any(p.poll() is None for p in procs):
                time.sleep(0.2)
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()
                    p.wait()
                    codes[r] = 124
                else:
                    codes[r] = p.returncode
            break
        time.sleep(0.05)
    reader.join(timeout=5.0)
    line = (out.get("line") or b"").decode(errors="replace").strip()
    if line:
        sys.stdout.write(line.splitlines()[-1] + "\n")
        sys.stdout.flush()
    worst = max(abs(c) if c is not None else 125 for c in codes)
    if worst:
        sys.stderr.write(f"bench.py launcher: rank exit codes {codes}\n")
    return worst


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_worker(seed, budget_s, config):
    """One single-threaded worker of the CPU baseline: the NumPy oracle over the 20-call schedule of its own window."""
    from oracle import ba_oracle as O
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence(synth.CONFIGS[config], seed=seed)
    win = od_pipe.prepare_window(det, orb)
    st0 = od_pipe.initial_guess(win)
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    sys.stdin.readline()                # all workers start together
    calls = 0
    t0 = time.perf_counter()
    t_end = t0 + budget_s
    while True:
        st, lam = st0, 1e-4
        for it in range(20):
            st, lam, _, _ = O.ba_iteration(it, st, win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii, win.time_idx,
                                           win.intrinsics, win.confidences, lam, initialize=it < 10)
            calls += 1
        if time.perf_counter() > t_end:
            break
    sys.stdout.write(json.dumps({"calls": calls, "seconds": time.perf_counter() - t0}) + "\n")
    sys.stdout.flush()


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup CPU quota when there is one (a
    GPU box shows every core of the host in the mask but grants a share of them)."""
    cores = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    cores = min(cores, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


def cpu_baseline(args):
    """The NumPy oracle (a structure-aware CPU port, NOT the reference's dense autograd path) on every host core of
    this box: one fresh single-threaded worker process per core, each over its own 500/50k window (the reference's
    outer loop over sequences is embarrassingly parallel).  Runs before this process touches the GPU."""
    cores = args.cpu_cores
    if not cores:
        cores = usable_cores()
        try:        # a GPU box grants about 16 host cores per GPU whatever its affinity mask shows; counting devices does not touch the GPU
            import torch
            cores = min(cores, 16 * max(1, torch.cuda.device_count()))
        except Exception:
            pass
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", NUMEXPR_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(100 + w), "--cpu-seconds",
                               str(args.cpu_seconds), "--config", args.config], env=env, stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, text=True) for w in range(cores)]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("CPU baseline worker failed to start")
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
        wall = time.perf_counter() - t0
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    calls = sum(r["calls"] for r in res)
    rate = sum(r["calls"] / r["seconds"] for r in res)
    return {"value": rate, "unit": "BA iterations/s", "cores": cores, "kind": "port",
            "per_core": rate / cores, "cores_in_affinity_mask": len(os.sched_getaffinity(0)),
            "sample": f"{calls} BA calls ({calls // 20} x the 20-call schedule) over {cores} independent 500/50k windows, one "
                      f"single-threaded NumPy fp64 oracle process per host core, {wall:.1f} s wall; the reference's own "
                      "dense-autograd path measured in the build container (8 vCPU, torch intra-op threads): 0.25 it/s over "
                      "the same schedule (tests/golden/c3.npz ref_wall_seconds)"}


# ------------------------------------------------------------------------------------------------ one rank
class Emitter:
    """Rank 0 prints the ONE JSON line exactly once, whichever thread gets there first."""

    def __init__(self, fd, rank):
        self.fd, self.rank = fd, rank
        self.lock = threading.Lock()
        self.done = False
        self.payload = None

    def emit(self, **extra):
        with self.lock:
            if self.done or self.rank != 0 or self.payload is None:
                return
            self.done = True
            out = dict(self.payload)
            out.update(extra)
            os.write(self.fd, (json.dumps(out) + "\n").encode())


def dry_run(args, em, rank, world):
    """The control plane of a multi-rank run and nothing else (no GPU, no kernels): used by the CPU test-suite to check
    the launcher, the rendezvous, the barrier / max-over-ranks reduce and the exit-code propagation."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    if rank == args.dry_run_fail_rank:
        os._exit(7)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # the timed region of the real run (timed_region) around a stand-in for the work: 20 ms of sleep.  VBA_BENCH_BARRIER_SLEEP_MS
    # makes the barrier slow on purpose -- the region must not see it (tests/test_bench_launcher.py)
    slow = float(os.environ.get("VBA_BENCH_BARRIER_SLEEP_MS", "0")) * 1e-3

    def barrier():
        if slow:
            time.sleep(slow)
        if world > 1:
            dist.barrier()

    def reduce_max(dt):
        if world == 1:
            return dt
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    dt = timed_region(lambda: time.sleep(0.02), barrier, lambda: None, reduce_max)
    em.payload = {"metric": METRIC, "value": 0.0, "unit": "BA iterations/s", "n_gpus": world, "dry_run": True,
                  "max_over_ranks": float(t.item()), "timed_region_ms": 1e3 * dt}
    em.emit()
    if world > 1:
        dist.destroy_process_group()


def run_rank(args):
    # Libraries (RCCL prints a version banner at init) must not add lines to stdout: everything written to fd 1
    # during the run goes to stderr, the ONE JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and "RANK" in os.environ:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the environment wins\n")
    em = Emitter(real_stdout, rank)

    # CPU baseline first: its worker processes are started (and gone) before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and not args.dry_run:      # (N = 1 only: at N > 1 the other ranks would wait for it)
        try:
            cpu = cpu_baseline(args)
        except Exception as exc:
            cpu = {"error": repr(exc)[:300]}

    import torch
    if args.dry_run:
        return dry_run(args, em, rank, world)
    ndev = torch.cuda.device_count()            # does not initialise the GPU
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible (there is no CPU fallback)")
    device = local % ndev
    # VBA_BENCH_FORCE_DIST=1 takes the multi-rank code path (process groups, collectives, sharded window) even with a
    # single rank -- the way to rehearse the RCCL leg on a one-GPU box
    force_dist = os.environ.get("VBA_BENCH_FORCE_DIST") == "1"
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("gloo")         # control plane on the host: barrier, max-over-ranks time
    torch.cuda.set_device(device)
    from vinsat_amd import od_pipe, synth
    from vinsat_amd.engine import BAEngine

    cfg = synth.CONFIGS[args.config]
    det, orb = synth.make_sequence(cfg, seed=rank)          # every replica gets its own sequence
    win = od_pipe.prepare_window(det, orb)
    st0 = od_pipe.initial_guess(win)
    n, m = win.time_idx.size, win.ii.size
    eng = BAEngine(n, m, windows=1, device=device)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # initialisation, not measurement: five full schedules (~5 ms), so that every kernel of both phases has been loaded and
    # launched and the clocks are up before the W warm-up steps (which, for small W, would only ever reach the landmark-only phase)
    run_steps(eng, st0, 100)
    # (the library replays a chained schedule as a hipGraph captured per schedule AND state-buffer parity; W warm-up steps that are
    # not a multiple of 20 end in an odd-length schedule and leave the timed 20-call schedules on the other parity -- both are
    # captured here, ~0.2 ms each once per process, so that neither falls into a timed region)
    run_steps(eng, st0, 1)
    run_steps(eng, st0, 40)
    run_steps(eng, st0, 1)
    run_steps(eng, st0, args.warmup)
    # EXACTLY args.steps steps between a barrier + device synchronisation and a device synchronisation; the trailing barrier and
    # the MAX over ranks follow outside the region (timed_region)
    dt = timed_region(lambda: run_steps(eng, st0, args.steps), barrier, torch.cuda.synchronize, reduce_max)
    value = world * args.steps / dt
    # companion of the driver-sized sample (20 steps = 1 ms): the same loop over 200 steps, same bracketing
    dt200 = timed_region(lambda: run_steps(eng, st0, 200), barrier, torch.cuda.synchronize, reduce_max)

    # ---- class times of the CHAINED schedule (the one `value` is timed on), HIP events on the library's stream at the class
    # boundaries of every call (VBA_OPT_CHAIN_PROFILE): accumulate (select and the previous call's accept test folded in),
    # solve (chunk elimination + cyclic reduction), trial.  The markers cost ~1 us each, so this pass follows the timed one.
    eng.set_chain_profile(True)
    eng.chain_profile(reset=True)
    run_steps(eng, st0, max(40, min(args.steps, 200) // 20 * 20))
    torch.cuda.synchronize()
    cls = eng.chain_profile(reset=True)
    eng.set_chain_profile(False)
    classes_ms = {k: v[0] for k, v in cls.items()}
    class_calls = {k: v[1] for k, v in cls.items()}
    # ... and the two phases of the schedule, each chained and timed by the wall clock around its ten calls
    phase = {"landmark_only": [], "full": []}
    for _ in range(5):
        eng.set_states(st0, 1e-4)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        eng.run_schedule(list(range(10)), [True] * 10)
        tb = time.perf_counter()
        eng.run_schedule(list(range(10, 20)), [False] * 10)
        tc = time.perf_counter()
        phase["landmark_only"].append(1e3 * (tb - ta) / 10)
        phase["full"].append(1e3 * (tc - tb) / 10)
    # the dominant class by total time in the schedule
    share = {k: classes_ms[k] * class_calls[k] for k in classes_ms}
    dom = max(share, key=share.get)
    alg = float(ALG_BYTES[dom](n, m))
    achieved = alg / (classes_ms[dom] * 1e-3) / 1e9
    # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/profile_pmc.sh + tools/summarize_pmc.py:
    # 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md); a kernel class sums its kernels
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"{args.profile_tag}_w1_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            pref = {"solve": ("k_solve", "k_cr_level0"), "select": ("k_select",)}.get(dom, ("k_" + dom, "k_obs_" + dom))
            names = [k for k in tj if k.startswith(pref)]
            calls = max([tj[k].get("calls", 0) for k in names] or [0])
            vals = [tj[k]["hbm_bytes_per_launch"] * tj[k].get("calls", calls) / max(calls, 1) for k in names if "hbm_bytes_per_launch" in tj[k]]
            traffic = float(sum(vals)) if vals else None
        except Exception:
            traffic = None
    mfma = None
    mpath = os.path.join(ROOT, "profiles", f"{args.profile_tag}_w1_mfma.json")
    if os.path.exists(mpath):
        try:
            mfma = json.load(open(mpath))
            mfma["peak_TFLOPs"] = MFMA_F64_PEAK_TFLOPS
        except Exception:
            mfma = None
    ms_per_step = 1e3 * dt / args.steps
    whole = survey_bytes_per_call(n, m)
    roofline = {"kernel": "solve class (chunk elimination + cyclic reduction of the separators + recovery)" if dom == "solve" else "k_" + dom,
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg,
                "avg_launch_ms": classes_ms[dom], "launches_timed": class_calls[dom],
                # machine-readable: `value` / ms_per_step are timed on the hipGraph replay of the schedule, the class times (and with
                # them achieved / frac) on a kernel-by-kernel pass with event markers -- ~15 % conservative, they do not sum to ms_per_step
                "timed_on_graph_replay": False, "value_timed_on_graph_replay": True,
                "timing_source": "HIP events on the library's stream at the class boundaries of the chained schedule (VBA_OPT_CHAIN_PROFILE); "
                                 "recorded on a pass that launches kernel by kernel with ~1 us of marker per boundary -- events cannot be recorded "
                                 "inside the graph replay that `value` is timed on, whose kernels run ~0.5 us shorter each "
                                 "(profiles/r05_w1_kernel_stats.csv: 14.6 + 7.8 + 14.6 us for the three kernels of the class)",
                "whole_call": {"bytes": whole, "achieved": whole / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                               "frac": whole / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "note": "SURVEY 8(d) B_alg = 208 m + 5000 n over the timed (chained) ms_per_step"},
                "mfma": mfma,
                "note": "single window = a latency-bound chain of dependent kernels; see 'batched' for the bandwidth regime"}

    # ---- the same calls through vba_iterate: states cross PCIe both ways on every call (never the headline value)
    host_roundtrip = python_ba = None
    if rank == 0:
        stt, lam = st0, 1e-4
        eng.iterate(0, True, lam, stt)
        th = time.perf_counter()
        nh = 100
        for k in range(nh):
            it, init = schedule(k)
            if it == 0:
                stt, lam = st0, 1e-4
            stt, lam, _, _, _ = eng.iterate(it, init, lam, stt)
        dth = time.perf_counter() - th
        host_roundtrip = {"value": nh / dth, "unit": "BA iterations/s", "ms_per_call": 1e3 * dth / nh,
                          "note": "vba_iterate: 40 kB of states host->device and back, one host synchronisation per call"}
        # ---- the drop-in Python call in the reference's loop shape (od_pipe.py:1036-1040), torch tensors in and out
        from vinsat_amd.ba import BA
        imu = torch.zeros((1, n, 1, 10), dtype=torch.float64)
        imu[0, :, 0, 6:10] = torch.from_numpy(win.cumrot_last)
        uv_t, xyz_t = torch.from_numpy(win.landmarks_uv)[None], torch.from_numpy(win.landmarks_xyz)[None]
        intr_t, conf_t = torch.from_numpy(win.intrinsics)[None], torch.from_numpy(win.confidences)
        gt_t, vel_t = torch.from_numpy(win.poses_gt), torch.from_numpy(win.velocities)[None]
        s0_t = torch.from_numpy(st0)[None]

        def ref_loop(reps):
            for _ in range(reps):
                states_t, lam_ = s0_t, 1e-4
                for it in range(20):
                    states_t, _, lam_, _ = BA(it, states_t, vel_t, imu, uv_t, xyz_t, win.ii, win.time_idx, intr_t, conf_t,
                                              1e-3, 1e-3, lam_, gt_t, initialize=it < 10, device=device)
        ref_loop(1)
        tp = time.perf_counter()
        ref_loop(5)
        dtp = time.perf_counter() - tp
        from vinsat_amd import ba as ba_mod
        ba_mod.release()            # its handle (memory, stream) is not needed any more
        python_ba = {"value": 100 / dtp, "unit": "BA iterations/s", "ms_per_call": 1e3 * dtp / 100,
                     "note": "vinsat_amd.ba.BA called as the reference's driver calls BA (for iter in range(20): states, ... = "
                             "BA(iter, states, ...)): window uploaded once (identity check; ndarray arguments compared byte for byte "
                             "by the library while the device works), states fed back stay on the device, and behind every call the "
                             "next one is enqueued speculatively (VBA_OPT_PIPELINE): the host waits for the kernel that decides the "
                             "call and reads the result from mapped host memory"}

    # ---- batched windows: W independent windows per launch
    batched = None
    if args.windows > 0 and rank == 0 and world == 1 and not force_dist:
        W = args.windows
        be = BAEngine(n, m, windows=W, device=device)
        load_windows(be, win, n, W)
        run_steps(be, st0, 20, windows=W)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        run_steps(be, st0, args.batched_steps, windows=W)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - tb
        bk = {k: [] for k in BAEngine.KERNELS}
        bbytes = {k: [] for k in BAEngine.KERNELS}      # algorithmic bytes of each of those launches (per window)
        for k in range(20):
            it, init = schedule(k)
            if it == 0:
                be.set_states(st0, 1e-4, window=-1)
            ms_k = be.step_profiled(it, init)
            for name, v in ms_k.items():
                if v > 0:
                    bk[name].append(v)
                    key = "assemble_init" if (name == "assemble" and init) else name
                    if name == "solve" and not init and ms_k.get("assemble", 0.0) == 0.0:
                        key = "solve_forming"
                    bbytes[name].append(float(ALG_BYTES[key](n, m)))
        bms = {k: (float(np.mean(v)) if v else 0.0) for k, v in bk.items()}
        per_kernel = {k: {"ms": bms[k], "GBps": (float(np.mean(bbytes[k])) * W / (bms[k] * 1e-3) / 1e9) if bms[k] > 0 else 0.0}
                      for k in bms}
        bdom = max(bms, key=lambda k: bms[k] * len(bk[k]))
        # bytes of the average call of the 20-call schedule (a class counts for the calls it ran in) over the TIMED
        # (chained) ms per call -- not over the sum of the serialised per-class times
        step_bytes = sum(sum(v) for v in bbytes.values()) / 20.0 * W
        bms_step = 1e3 * dtb / args.batched_steps
        blat, bchunk = be.mode()
        batched = {"windows": W, "value": W * args.batched_steps / dtb, "unit": "BA iterations/s", "steps": args.batched_steps,
                   "kernel_set": "latency" if blat else "bandwidth",
                   "solver": f"chunks of {bchunk} + cyclic reduction" if bchunk else "sequential walk, four windows per wavefront",
                   "ms_per_step": bms_step, "dominant_kernel": "k_" + bdom,
                   "roofline": {"kernel": "k_" + bdom, "bound": "hbm", "achieved": per_kernel[bdom]["GBps"], "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": per_kernel[bdom]["GBps"] / HBM_PEAK_GBS},
                   "whole_step_GBps": step_bytes / (1e-3 * bms_step) / 1e9,
                   "whole_step_frac": step_bytes / (1e-3 * bms_step) / 1e9 / HBM_PEAK_GBS,
                   "whole_step_GBps_serialised": step_bytes / (1e-3 * sum(bms[k] * len(bk[k]) / 20.0 for k in bms)) / 1e9,
                   "whole_step_bytes_per_window": step_bytes / W,
                   # the same time against SURVEY 8(d)'s per-unit figure B_alg = 208 m + 5000 n (which charges a re-read of the
                   # observations for the residual pass and the bands' round trip through memory: both are gone here)
                   "whole_step_frac_survey_bytes": survey_bytes_per_call(n, m) * W / (1e-3 * bms_step) / 1e9 / HBM_PEAK_GBS,
                   "kernels": per_kernel}
        # measured HBM traffic of the same chained schedule from the committed rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE
        # per kernel, tools/profile_pmc.sh / summarize_pmc.py), if that profile was taken at this many windows
        tpath4 = os.path.join(ROOT, "profiles", f"{args.profile_tag}_w{W}_traffic.json")
        if os.path.exists(tpath4):
            try:
                tj4 = json.load(open(tpath4))
                tot = sum(v["hbm_bytes_per_launch"] * v["calls"] for k, v in tj4.items()
                          if k.startswith("k_") and k not in ("k_broadcast_states", "k_set_counts", "k_reset_calls", "k_clear_hist")
                          and "hbm_bytes_per_launch" in v)
                calls = max(v["calls"] for k, v in tj4.items() if k.startswith("k_trial"))
                per_step = tot / max(calls, 1)
                batched["hbm_traffic"] = {"bytes_per_step": per_step, "GBps": per_step / (1e-3 * bms_step) / 1e9,
                                          "frac": per_step / (1e-3 * bms_step) / 1e9 / HBM_PEAK_GBS,
                                          "source": os.path.relpath(tpath4, ROOT) + " (PMC counters of one chained schedule; rate over this run's timed ms_per_step)"}
            except Exception:
                pass
        be.close()

    # ---- the regime between one window and 4096: W windows per handle, the handle choosing kernel set and solver itself
    sync = torch.cuda.synchronize
    batched_sweep = configs = python_batch = None
    if rank == 0 and world == 1 and not force_dist and args.sweep:
        pts = []
        for W in sorted({int(x) for x in args.sweep.split(",") if x}):
            if W == 1:
                pts.append({"windows": 1, "value": value, "ms_per_step": ms_per_step, "kernel_set": "latency",
                            "solver": f"chunks of {eng.mode()[1]} + cyclic reduction",
                            "whole_step_frac_survey_bytes": whole / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS})
            else:
                pts.append(sweep_point(BAEngine, win, st0, n, m, W, device, sync))
        if batched is not None and batched["windows"] not in [q["windows"] for q in pts]:
            pts.append({"windows": batched["windows"], "value": batched["value"], "ms_per_step": batched["ms_per_step"],
                        "kernel_set": batched["kernel_set"], "solver": batched["solver"],
                        "whole_step_frac_survey_bytes": batched["whole_step_frac_survey_bytes"]})
        pts.sort(key=lambda q: q["windows"])
        vals = [q["value"] for q in pts]
        batched_sweep = {"unit": "BA iterations/s", "workload": "W independent C3 windows on one handle, chained 20-call schedule",
                         "points": pts, "monotone": all(b >= 0.97 * a for a, b in zip(vals, vals[1:])),
                         "note": "kernel set and solver are the handle's own choice (vba_create_mode -1): latency-mode kernels up to "
                                 "38 (50000 / rows)^0.7 windows (exponent 0.46 beyond 50000 rows), partitioned solve up to 1023 windows, see DESIGN.md section 3"}
        # the same batch behind the reference's call surface: BA_window on lists of 22 windows (the reference's 22 sequences,
        # od_pipe.py:1069-1077), arguments as the driver holds them
        from vinsat_amd import ba as ba_mod
        B = 22
        imu1 = torch.zeros((1, n, 1, 10), dtype=torch.float64)
        imu1[0, :, 0, 6:10] = torch.from_numpy(win.cumrot_last)
        one = dict(states=torch.from_numpy(st0)[None], vel=torch.from_numpy(win.velocities)[None], imu=imu1,
                   uv=torch.from_numpy(win.landmarks_uv)[None], xyz=torch.from_numpy(win.landmarks_xyz)[None],
                   intr=torch.from_numpy(win.intrinsics)[None], conf=torch.from_numpy(win.confidences))
        rep = lambda key: [one[key]] * B
        sched_i, sched_b = list(range(20)), [k < 10 for k in range(20)]

        def window_call():
            return ba_mod.BA_window(sched_i, sched_b, rep("states"), rep("vel"), rep("imu"), rep("uv"), rep("xyz"), [win.ii] * B,
                                    [win.time_idx] * B, rep("intr"), rep("conf"), [1e-4] * B, device=device)
        window_call()
        tq = time.perf_counter()
        for _ in range(3):
            window_call()
        dq = (time.perf_counter() - tq) / 3
        st_l, lam_l = rep("states"), [1e-4] * B
        tq2 = time.perf_counter()
        for it in range(20):
            st_l, _, lam_l, _ = ba_mod.BA(it, st_l, rep("vel"), rep("imu"), rep("uv"), rep("xyz"), [win.ii] * B, [win.time_idx] * B,
                                          rep("intr"), rep("conf"), 1e-3, 1e-3, lam_l, None, initialize=it < 10, device=device)
        dq2 = time.perf_counter() - tq2
        ba_mod.release()
        python_batch = {"windows": B, "BA_window": {"value": 20 * B / dq, "unit": "BA iterations/s", "ms_per_schedule": 1e3 * dq},
                        "BA_per_call": {"value": 20 * B / dq2, "unit": "BA iterations/s", "ms_per_call": 1e3 * dq2 / 20},
                        "note": "vinsat_amd.ba.BA / BA_window on lists of 22 windows (ragged batch, one handle); BA_window = the "
                                "20-call loop as one chained device call incl. upload check, states up and results back; BA per call "
                                "= states fed back stay on the device, every ndarray argument compared with its uploaded copy per call"}
    if rank == 0 and world == 1 and not force_dist and not args.no_configs:
        try:
            configs = config_series(BAEngine, od_pipe, synth, device, sync)
        except Exception as exc:
            configs = {"error": repr(exc)[:300]}

    # ---- the drop-in pipeline end to end
    driver = None
    if rank == 0 and world == 1 and not force_dist and not args.no_driver:
        try:
            from vinsat_amd import errors_eval
            driver = driver_series(od_pipe, synth, errors_eval, device)
        except Exception as exc:
            driver = {"error": repr(exc)[:300]}

    # ---- accuracy: the 20-call schedule once more from the initial guess, against the reference's final states
    accuracy = None
    gpath = os.path.join(ROOT, "tests", "golden", f"{cfg.name.lower()}.npz")
    if rank == 0 and os.path.exists(gpath):
        g = np.load(gpath)
        eng.set_states(g["states0"][0], 1e-4)           # the timed path: the 20 calls as ONE chained schedule
        eng.run_schedule([schedule(k)[0] for k in range(20)], [schedule(k)[1] for k in range(20)])
        s_fin = eng.get_states()[0]
        ref = g["states_out_19"][0]
        dpos = np.linalg.norm(s_fin[:, :3] - ref[:, :3], axis=1)
        ang = 2 * np.arccos(np.clip(np.abs((s_fin[:, 3:7] * ref[:, 3:7]).sum(-1)), 0, 1))
        gt = g["in_poses_gt_eci"]
        accuracy = {"pose_rmse_vs_ref_km": float(np.sqrt((dpos ** 2).mean())), "max_rel_pos_err_vs_ref": float(np.abs(s_fin[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max()),
                    "max_attitude_err_vs_ref_rad": float(ang.max()),
                    "pos_rmse_vs_truth_km": float(np.sqrt(((s_fin[:, :3] - gt[:, :3]) ** 2).sum(1).mean())),
                    "path": "set_states + vba_run_schedule(20 calls), the path `value` is timed on",
                    "reference": "states after call 19 of the reference's own run on the same inputs (tests/golden)"}

    # ---- free-landmark Schur-complement add-on (PARITY UNPINNED: no counterpart in the reference; not part of the metric)
    schur = None
    if rank == 0 and world == 1 and not args.no_schur and not force_dist:
        try:
            from vinsat_amd.schur import SchurBA
            d = synth.make_tracked_landmarks(n_poses=500, n_landmarks=20000, seed=0)
            rng = np.random.default_rng(1)
            st_s = d["states_gt"].copy()
            st_s[:, :3] += rng.normal(0, 2.0, st_s[:, :3].shape)
            sb = SchurBA(st_s, d["X0"], d["uv"], np.full(d["uv"].shape[0], 0.95), d["pose_of_row"], d["landmark_of_row"], d["intrinsics"],
                         sigma_prior=d["sigma"], device=device)
            lam_s, ms_s, costs = 1e-4, [], []
            for it in range(6):
                c0, c1, ok = sb.iterate(lam_s)
                ms_s.append(sb.last_ms())
                costs.append(c0)
                lam_s = max(lam_s * 0.1, 1e-9) if ok else lam_s * 10
            Np = (6 * 500 + 255) // 256 * 256
            fac = float(np.mean([x["factor"] for x in ms_s[1:]]))
            schur = {"parity": "unpinned (the reference keeps landmarks fixed, BA_filtering.py:32-37: no counterpart)",
                     "poses": 500, "landmarks": int(d["X_true"].shape[0]), "rows": int(d["uv"].shape[0]), "reduced_system": 3000,
                     "ms_per_trial": {k: float(np.mean([x[k] for x in ms_s[1:]])) for k in ms_s[0]},
                     "trials_per_s": 1e3 / float(np.mean([sum(x.values()) for x in ms_s[1:]])),
                     "cholesky_TFLOPs": Np ** 3 / 3.0 / (fac * 1e-3) / 1e12, "fp64_matrix_peak_TFLOPs": MFMA_F64_PEAK_TFLOPS,
                     "cost_first_last": [costs[0], costs[-1]],
                     "note": "dense reduced camera system factorised on the matrix cores by panels of 256 columns with look-ahead (round 4); "
                             "MFMA counters of the 12000 x 12000 case: profiles/r04_schur_mfma.json (trailing update 29.3 TFLOP/s = 37 % of the "
                             "fp64 matrix peak over all its launches, whole factorisation 25.4 TFLOP/s = 32 %)"}
            sb.close()
            # the size at which a dense reduced camera system feeds the matrix cores: 2000 poses / 60 000 landmarks, 12 000 x 12 000
            d = synth.make_tracked_landmarks(n_poses=2000, n_landmarks=60000, seed=0)
            st_s = d["states_gt"].copy()
            st_s[:, :3] += rng.normal(0, 2.0, st_s[:, :3].shape)
            sb = SchurBA(st_s, d["X0"], d["uv"], np.full(d["uv"].shape[0], 0.95), d["pose_of_row"], d["landmark_of_row"], d["intrinsics"],
                         sigma_prior=d["sigma"], device=device)
            lam_s, facs = 1e-4, []
            for it in range(4):
                c0, c1, ok = sb.iterate(lam_s)
                facs.append(sb.last_ms()["factor"])
                lam_s = max(lam_s * 0.1, 1e-9) if ok else lam_s * 10
            Np2 = (6 * 2000 + 255) // 256 * 256
            fac2 = float(np.mean(facs[1:]))
            schur["large"] = {"poses": 2000, "landmarks": int(d["X_true"].shape[0]), "rows": int(d["uv"].shape[0]), "reduced_system": 12000,
                              "factor_ms": fac2, "cholesky_TFLOPs": Np2 ** 3 / 3.0 / (fac2 * 1e-3) / 1e12,
                              "frac_of_fp64_matrix_peak": Np2 ** 3 / 3.0 / (fac2 * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS}
            sb.close()
        except Exception as exc:
            schur = {"error": repr(exc)[:300]} if schur is None else dict(schur, large_error=repr(exc)[:300])

    em.payload = {
        "metric": METRIC,
        "value": value, "unit": "BA iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "value_200": world * 200 / dt200, "ms_per_step_200": 1e3 * dt200 / 200,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{cfg.name}: {n}-pose / {m}-observation window, 20-call schedule (10 landmark-only + 10 full)"
                               + (f", {world} independent windows (one per rank)" if world > 1 else ""),
                   "poses": n, "observations": m, "windows_per_gpu": 1, "parallelism": f"replicas{world}",
                   "devices_visible": ndev, "ranks_per_device": (world + ndev - 1) // ndev},
        "phase_ms": {k: float(np.median(v)) if v else None for k, v in phase.items()},
        "phase_ms_source": "ten chained calls of the phase (vba_run_schedule), wall clock; their mean is ms_per_step plus one host synchronisation per ten calls",
        "chain_classes_ms": classes_ms,
        "chain_classes_source": "HIP events at the class boundaries of the chained schedule; the dynamics factor rides in the accumulation's grid, select and accept test are folded into it",
        "roofline": roofline,
        "host_roundtrip": host_roundtrip,
        "python_BA_call": python_ba,
        "accuracy": accuracy,
        "cpu_baseline": cpu,
        "batched": batched,
        "batched_sweep": batched_sweep,
        "python_BA_batch": python_batch,
        "configs": configs,
        "driver": driver,
        "schur_addon": schur,
    }

    # ---- observation-sharded mode: ONE window whose rows are split over the ranks, collectives over RCCL
    sharded = None
    hung = threading.Event()
    if (world > 1 or force_dist) and not args.no_sharded:
        fake_rccl = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
        rehearse = ndev < world and os.environ.get("VBA_BENCH_FAKE_RCCL") == "1" and os.path.exists(fake_rccl)
        if ndev < world and not rehearse:
            sharded = {"skipped": f"{world} ranks share {ndev} device(s): RCCL needs one device per rank (rehearsal run)"}
        else:
            # A hung collective must neither cost the headline line nor read as success: the watchdog prints the line with the
            # error ("sharded_failed": true) and ends this rank (every rank runs its own watchdog) -- with exit code 0 by default,
            # since the replicas were timed before this leg began; --strict-sharded makes it exit code 3.  Nothing is retried from
            # a process that has touched the GPU.
            got = {}        # what has been measured so far (the caller-dispatched leg runs first)

            def bail():
                hung.set()
                if got:     # a later leg hung: the line keeps what was measured and says so ...
                    em.emit(sharded=dict(got, error="the sharded leg timed out (collective hung); entries present were measured before"),
                            sharded_failed=True)
                else:
                    em.emit(sharded={"error": "sharded measurement timed out (collective hung)"}, sharded_failed=True)
                # a hung collective is a defect, never a result: the line says so (sharded.error / native_error, "sharded_failed": true).
                # The headline measurement -- the replicas above, timed and complete before this leg started -- stands; with
                # --strict-sharded the run fails as well
                os._exit(3 if args.strict_sharded else 0)

            watchdog = threading.Timer(420.0, bail)
            watchdog.daemon = True
            watchdog.start()
            try:
                from vinsat_amd.dist import ShardedBA
                # VBA_BENCH_FAKE_RCCL=1 on a box with fewer devices than ranks: a REHEARSAL of the library-issued leg on the test
                # double of RCCL (tests/fake_rccl: host shared memory) -- every rank process drives the real kernels and the real
                # schedule; the rate means nothing (the exchanges synchronise the stream), the leg running through does
                nccl = None if rehearse else (dist.new_group(backend="nccl", device_id=torch.device("cuda", device)) if hasattr(dist, "new_group") else None)
                ns = min(args.steps, 100)
                rccl_ranks = 0 if rehearse else dist.get_world_size(nccl)

                def run(sba, st_s, count):
                    """`count` BA() calls walking the 20-call schedule: one chained device call per schedule where the library
                    issues the exchanges (vba_sh_run_schedule), call by call through torch.distributed otherwise."""
                    k = 0
                    while k < count:
                        cnt = min(20, count - k)
                        sba.set_states(st_s, 1e-4)
                        calls = [schedule(j) for j in range(cnt)]
                        if getattr(sba.engine, "native", False):
                            sba.run_schedule([c[0] for c in calls], [c[1] for c in calls])
                        else:
                            for it, init in calls:
                                sba.step(it, init)
                        k += cnt

                def rate(sba, st_s):
                    run(sba, st_s, 20)
                    return ns / timed_region(lambda: run(sba, st_s, ns), barrier, torch.cuda.synchronize, reduce_max)

                def leg(win_s, st_s, what):
                    """ONE window of FIXED size, its rows split over the ranks: the rate with the exchanges dispatched by the caller
                    (torch.distributed), then issued by the library, beside the same window unsharded on one GPU (rank 0's device)."""
                    ns_, ms_ = win_s.time_idx.size, win_s.ii.size
                    e1 = BAEngine(ns_, ms_, device=device)
                    load_windows(e1, win_s, ns_, 1)
                    run_steps(e1, st_s, 20)
                    one = ns / timed_region(lambda: run_steps(e1, st_s, ns), barrier, torch.cuda.synchronize, reduce_max)
                    e1.close()
                    out = {"window": what, "poses": int(ns_), "observations_total": int(ms_), "observations_per_rank": int(ms_ // world),
                           "unit": "BA iterations/s", "rccl_ranks": rccl_ranks, "one_gpu_unsharded": one}
                    v_torch = None
                    if not rehearse:
                        sba = ShardedBA.from_window(win_s, device=device, group=nccl)
                        v_torch = rate(sba, st_s)
                        sba.close()
                    out.update(value=v_torch, value_torch_dispatched=v_torch, vs_one_gpu=(v_torch / one) if v_torch else None)
                    got[what.split(":")[0]] = dict(out)
                    # ... and issued by the library (RCCL on its own stream, one host call per schedule; the id of its communicator
                    # travels over the gloo control group).  Whether the leg runs is decided by ALL ranks together (a rank that could
                    # not join must not leave the others inside a collective).
                    sba, ok = None, 1.0
                    try:
                        sba = ShardedBA.from_window(win_s, device=device, native=True, rccl_path=fake_rccl if rehearse else None)
                    except Exception as exc:
                        ok = 0.0
                        out["native_error"] = repr(exc)[:300]
                    agree = torch.tensor([ok], dtype=torch.float64)
                    dist.all_reduce(agree, op=dist.ReduceOp.MIN)
                    if float(agree.item()) > 0.5:
                        v_native = rate(sba, st_s)
                        first_b, n_miss, n_lm = sba.engine.stats()
                        out.update(value_library_issued=v_native, library_issued_vs_one_gpu=v_native / one,
                                   first_exchange_bytes_per_rank=first_b, calls_repeated_after_a_missed_select=n_miss,
                                   calls_finished_by_the_lm_loop=n_lm, rccl_library=sba.engine.rccl_path)
                    elif "native_error" not in out:
                        out["native_error"] = "another rank could not join the library's communicator"
                    if sba is not None:
                        sba.close()
                    got[what.split(":")[0]] = dict(out)
                    return out

                got.update({"unit": "BA iterations/s", "rccl_ranks": rccl_ranks,
                            **({"rehearsal": "library-issued leg on tests/fake_rccl (ranks share a device): not a measurement"} if rehearse else {}),
                            "collectives_per_call": "3 all-gathers (|r| keys, per-pose blocks, trial sums) over RCCL",
                            "transport": "torch.distributed all_gather_into_tensor between the stage calls (value); ncclAllGather issued by "
                                         "libvinsat_ba.so on its stream, calls chained on the device (value_library_issued)",
                            "protocol_library_issued": "carried keys: all-gather of [warm histogram | block sums], of the median bin's bucket, of "
                                                       "the per-pose normal equations (vba_sh_run_schedule)",
                            "note": "`value` of every entry is the caller-dispatched rate: the library-issued path has been compared with it bit "
                                    "for bit at one rank and on the test double of tests/fake_rccl only, not yet on two real devices"})
                # the configs BASELINE.json names as sharded, FIXED totals whatever the rank count: C4 = 500 poses / 200 000 rows,
                # C5 = 2004 poses / 500 000 rows (strong scaling of one window) ...
                for name in ("C4", "C5"):
                    win_s = od_pipe.prepare_window(*synth.make_sequence(name))
                    leg(win_s, od_pipe.initial_guess(win_s), f"{name}: fixed total, rows split over {world} rank(s)")
                # ... and the headline window with its rows multiplied by the rank count (per-rank rows fixed)
                cfg_s = synth.WindowConfig("sharded", cfg.n_poses, cfg.obs_per_pose * world, cfg.stride)
                win_s = od_pipe.prepare_window(*synth.make_sequence(cfg_s, seed=0))
                weak = leg(win_s, od_pipe.initial_guess(win_s), f"weak: {cfg.name} with {cfg.obs_per_pose} x {world} rows per pose")
                got.update(value=weak["value"], value_library_issued=weak.get("value_library_issued"), poses=weak["poses"],
                           observations_total=weak["observations_total"], observations_per_rank=weak["observations_per_rank"],
                           value_is="the weak-scaled window's caller-dispatched rate (the entry of rounds 1-4); C4 / C5 are the fixed-size windows")
                sharded = dict(got)
            except Exception as exc:      # never lose the headline line to the secondary measurement
                sharded = {"error": repr(exc)[:300]}
            watchdog.cancel()

    em.emit(sharded=sharded, sharded_failed=bool(isinstance(sharded, dict) and "error" in sharded))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if isinstance(sharded, dict) and "error" in sharded and args.strict_sharded:
        sys.exit(3)


def main():
    args = parse()
    if args.cpu_worker >= 0:
        cpu_worker(args.cpu_worker, args.cpu_seconds, args.config)
        return
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
