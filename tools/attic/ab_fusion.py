#!/usr/bin/env python3
"""One C3 window: us per call of the chained 20-call schedule for several VBA_OPT_FUSION masks, and whether the final states
are bit-equal to the first mask's."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
masks = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["13", "15"])]
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
iters, inits = list(range(20)), [k < 10 for k in range(20)]
ref = None
for rep in range(2):
    for mask in masks:
        e = BAEngine(n, m)
        e.set_fusion(mask)
        e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
        e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
        for r in range(5):
            e.set_states(st0, 1e-4); e.run_schedule(iters, inits)
        t0 = time.perf_counter()
        for r in range(50):
            e.set_states(st0, 1e-4); e.run_schedule(iters, inits)
        dt = (time.perf_counter() - t0) / 1000
        st = e.get_states()[0]
        if ref is None:
            ref = st
        print(f"mask {mask}: {1e6 * dt:.2f} us per call, bits equal to first: {np.array_equal(st, ref)}", flush=True)
        e.close()
