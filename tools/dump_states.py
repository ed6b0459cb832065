#!/usr/bin/env python3
"""Final states, lamda and trial counts of the C3 20-call schedule (and of a 300-pose window with chunks of 5) to argv[1] (.npz):
for comparing two builds of the library bit by bit (VBA_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
out = {}
for name, cfg, chunk in (("c3", "C3", 0), ("w300", synth.WindowConfig("w", 300, 20, 5), 5)):
    win = od_pipe.prepare_window(*synth.make_sequence(cfg, seed=3))
    n, m = win.time_idx.size, win.ii.size
    e = BAEngine(n, m)
    if chunk:
        e.set_solver(chunk, -1)
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    e.set_states(od_pipe.initial_guess(win, seed=3), 1e-4)
    tr = e.run_schedule(list(range(20)), [k < 10 for k in range(20)])
    st = e.get_states()
    out[name + "_states"], out[name + "_lam"], out[name + "_trials"] = st[0], np.array(st[1]), np.array(tr)
    e.close()
np.savez(sys.argv[1], **out)
