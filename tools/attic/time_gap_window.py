#!/usr/bin/env python3
"""Time the two-pass window (gaps of 945 / 555 s: long RK4 chains) call by call -- diagnostic."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
det, orb = synth.make_two_pass_sequence()
win = od_pipe.prepare_window(det, orb)
n, m = win.time_idx.size, win.ii.size
for hop in (False, True):
    e = BAEngine(n, m)
    e.set_integrator(hop)
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    e.set_states(od_pipe.initial_guess(win), 1e-4)
    for it in range(10):
        e.step(it, True)
    acc = {}
    for it in range(10, 20):
        for k, v in e.step_profiled(it, False).items():
            acc.setdefault(k, []).append(v)
    print("hop100" if hop else "1s-steps", {k: round(float(np.mean(v)), 4) for k, v in acc.items() if np.mean(v) > 0.001})
    e.close()
