#!/usr/bin/env python3
"""W batched C3 windows: ms per step of the chained 20-call schedule for several warm-bin widths (and the exact select)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["exact", "48", "49", "50"]
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m, windows=W)
for w in range(W):
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
iters, inits = list(range(20)), [k < 10 for k in range(20)]
ref = None
for mode in modes:
    if mode == "exact":
        e.set_warm_select(0)
    else:
        e.set_warm_select(1)
        e.set_warm_shift(int(mode))
    e.set_states(st0, 1e-4, window=-1)
    e.run_schedule(iters, inits)
    t0 = time.perf_counter()
    for r in range(2):
        e.set_states(st0, 1e-4, window=-1)
        e.run_schedule(iters, inits)
    dt = (time.perf_counter() - t0) / 40
    got = e.get_states(window=W - 1)
    if ref is None:
        ref = got
    same = (got[0] == ref[0]).all() and got[1] == ref[1]
    print(f"W {W} mode {mode}: {1e3 * dt:.3f} ms per step, {W / dt:.0f} it/s, misses {e.warm_select_misses()}, bits equal to first mode: {same}", flush=True)
e.close()
