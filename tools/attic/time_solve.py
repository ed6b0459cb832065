#!/usr/bin/env python3
"""W batched C3 windows: HIP-event time of the solve class (and the others) of full-phase calls (vba_step_profiled)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m, windows=W)
for w in range(W):
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
e.set_states(st0, 1e-4, window=-1)
e.run_schedule(list(range(10)), [True] * 10)
acc = {}
for it in range(10, 16):
    ms = e.step_profiled(it, False)
    for k, v in ms.items():
        acc.setdefault(k, []).append(v)
print(os.environ.get("VBA_LIB", "default"), {k: round(float(np.mean(v[1:])), 3) for k, v in acc.items()}, flush=True)
e.close()
