#!/usr/bin/env python3
"""Batched chain (W windows of C3, 20 calls) for a list of accumulate lane counts (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
W = int(sys.argv[1]); lanes = [int(x) for x in sys.argv[2:]] or [8, 4, 16]
det, orb = synth.make_sequence("C3"); win = od_pipe.prepare_window(det, orb); st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m, windows=W)
for w in range(W):
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
iters, inits = list(range(20)), [k < 10 for k in range(20)]
for rep in range(2):
    for G in lanes:
        e.set_accumulate_lanes(G)
        e.set_states(st0, 1e-4, window=-1); e.run_schedule(iters, inits)
        t0 = time.perf_counter()
        for r in range(3):
            e.set_states(st0, 1e-4, window=-1); e.run_schedule(iters, inits)
        dt = (time.perf_counter() - t0) / 60
        print(f"W {W} lanes {G}: {1e3 * dt:.3f} ms per step, {W / dt:.0f} it/s", flush=True)
e.close()
