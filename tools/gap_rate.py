#!/usr/bin/env python3
"""bench.py's GAP leg alone: the two-pass window (25 poses, one gap of ~945 s), 20 full calls of a chained schedule -- diagnostic."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
win = od_pipe.prepare_window(*synth.make_two_pass_sequence())
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
print("poses", n, "rows", m, "gaps", np.diff(win.time_idx)[np.diff(win.time_idx) > 64])
for hop in ((False,) if "--rk4" in sys.argv else (False, True)):
    e = BAEngine(n, m)
    e.set_integrator(hop)
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    iters, inits = list(range(20)), [False] * 20
    e.set_states(st0, 1e-4)
    e.run_schedule(iters, inits)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        e.set_states(st0, 1e-4)
        e.run_schedule(iters, inits)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 100
    print("hop" if hop else "rk4", f"{ms:.4f} ms per call, {1e3 / ms:.0f} it/s")
    e.close()
