#!/usr/bin/env python3
"""Diagnostic (CPU): conditioning of the long edges of a random window at the state the oracle reaches before the last call of the
stress schedule -- radius, speed, |Phi| and the change of the end state for a 1e-15 relative change of the start state.
usage: diag_conditioning.py SEED"""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from random_windows import SCHEDULE, make
from oracle import ba_oracle as O
win, xyz, uv, ii, conf, t, st0 = make(int(sys.argv[1]), long_gaps=True)
ref, lam = st0.copy(), 1e-4
for it, init in SCHEDULE[:-1]:
    ref, lam, _, ntr = O.ba_iteration(it, ref, win.cumrot_last, uv, xyz, ii, t, win.intrinsics, conf, lam, initialize=init)
x = np.concatenate([ref[:, :3], ref[:, 7:10]], -1)
steps = O.step_counts(t)
print("radius km", np.linalg.norm(x[:, :3], axis=1).round(0))
print("speed km/s", np.linalg.norm(x[:, 3:], axis=1).round(2))
xh, Phi = O.propagate_orbit(x, steps, stm=True)
for i in np.nonzero(steps > 64)[0]:
    # sensitivity: relative perturbation of 1e-15 in the start state
    xp = x.copy(); xp[i] *= (1 + 1e-15)
    xhp = O.propagate_orbit(xp[i:i+1], steps[i:i+1], stm=False)
    print("edge", i, "steps", steps[i], "|Phi|max", np.abs(Phi[i]).max().round(1), "radius along? end", np.linalg.norm(xh[i,:3]).round(0),
          "rel change of end state for 1e-15 rel change of start:", np.abs(xhp[0]-xh[i]).max()/np.abs(xh[i]).max())
