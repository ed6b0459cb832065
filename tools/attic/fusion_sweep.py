#!/usr/bin/env python3
"""Time of the 20-call schedule of one window for the fusion masks of the latency mode (VBA_OPT_FUSION; diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
det, orb = synth.make_sequence(cfg)
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m)
e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
iters, inits = list(range(20)), [k < 10 for k in range(20)]
for mask in [int(x) for x in sys.argv[2:]] or [0, 1, 2, 3, 0]:
    e.set_fusion(mask)
    for rep in range(3):
        e.set_states(st0, 1e-4)
        e.run_schedule(iters, inits)
    t0 = time.perf_counter()
    for rep in range(20):
        e.set_states(st0, 1e-4)
        e.run_schedule(iters, inits)
    dt = (time.perf_counter() - t0) / 400
    print(f"{cfg} fusion {mask}: {1e6 * dt:7.1f} us per call, {1 / dt:8.0f} it/s, warm misses {e.warm_select_misses()}", flush=True)
e.close()
