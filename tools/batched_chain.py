#!/usr/bin/env python3
"""W batched C3 windows, the 20-call schedule chained (vba_run_schedule) `reps` times: the run to put under
rocprofv3 --kernel-trace for the per-kernel times of the chained batched regime (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
det, orb = synth.make_sequence(os.environ.get("VBA_CONFIG", "C3"))       # VBA_CONFIG: another BASELINE config
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
# VBA_MODE: lat / bw force the kernel set (default: the handle's own choice); VBA_SOLVER: part / seq force the solver
mode = {"lat": 1, "bw": 0}.get(os.environ.get("VBA_MODE", ""), -1)
e = BAEngine(n, m, windows=W, mode=mode)
if os.environ.get("VBA_SOLVER") == "part":
    e.set_solver(8, -1)
elif os.environ.get("VBA_SOLVER") == "seq":
    e.set_solver(0)
for w in range(W):
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
if os.environ.get("VBA_FUSION"):
    e.set_fusion(int(os.environ["VBA_FUSION"]))
iters, inits = list(range(20)), [k < 10 for k in range(20)]
e.set_states(st0, 1e-4, window=-1)
e.run_schedule(iters, inits)
t0 = time.perf_counter()
for r in range(reps):
    e.set_states(st0, 1e-4, window=-1)
    e.run_schedule(iters, inits)
dt = (time.perf_counter() - t0) / (20 * reps)
print(f"W {W}: {1e3 * dt:.3f} ms per step, {W / dt:.0f} it/s, misses {e.warm_select_misses()}", flush=True)
e.close()
