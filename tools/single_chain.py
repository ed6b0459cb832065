#!/usr/bin/env python3
"""One C3 window, the 20-call schedule chained `reps` times: the run to put under rocprofv3 --kernel-trace for the kernel
times and gaps of the headline path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m)
e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
iters, inits = list(range(20)), [k < 10 for k in range(20)]
for r in range(3):
    e.set_states(st0, 1e-4); e.run_schedule(iters, inits)
t0 = time.perf_counter()
for r in range(reps):
    e.set_states(st0, 1e-4); e.run_schedule(iters, inits)
dt = (time.perf_counter() - t0) / (20 * reps)
print(f"{1e6 * dt:.2f} us per call", flush=True)
e.close()
