#!/usr/bin/env python3
"""Diagnostic: one random window (tests/random_windows.py, with long gaps) through a one-window handle, small and big bandwidth-mode
handles and the oracle, free running over the stress schedule -- who agrees with whom.  usage: diag_window.py SEED"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from random_windows import SCHEDULE, make
from oracle import ba_oracle as O
from vinsat_amd.engine import BAEngine
seed = int(sys.argv[1])
win, xyz, uv, ii, conf, t, st0 = make(seed, long_gaps=True)
n, m = t.size, ii.size
print("n", n, "m", m, "gaps", np.diff(t))
def rel(a, b): return np.abs(a - b).max() / np.abs(b).max()
def run(W, mode, chained):
    e = BAEngine(max(n, 2), max(m, 1), windows=W, mode=mode)
    for k in range(W):
        e.upload_observations(xyz, uv, conf, ii, n, window=k); e.upload_window(win.intrinsics, win.cumrot_last, t, window=k)
        e.set_states(st0, 1e-4, window=k)
    hist = []
    if chained:
        e.run_schedule([s[0] for s in SCHEDULE], [s[1] for s in SCHEDULE])
    else:
        for it, init in SCHEDULE:
            e.step(it, init)
            r = e.get_states(0)
            hist.append((r[1], r[3]))
    out = e.get_states(0)
    print(f"W {W} mode {e.mode()} chained {chained}: lam {out[1]} ntr {out[3]} hist {hist}")
    e.close()
    return out[0]
a = run(1, -1, False)
b = run(2, 0, False)
c = run(300, 0, False)
d = run(300, 0, True)
ref, lam = st0.copy(), 1e-4
oh = []
for it, init in SCHEDULE:
    ref, lam, _, ntr = O.ba_iteration(it, ref, win.cumrot_last, uv, xyz, ii, t, win.intrinsics, conf, lam, initialize=init)
    oh.append((lam, ntr))
print("oracle hist", oh)
print("single vs oracle", rel(a, ref), "bw2 vs oracle", rel(b, ref), "bw300 vs oracle", rel(c, ref), "bw300 chained vs stepped", rel(d, c))
print("single vs bw2", rel(a, b), "bw2 vs bw300", rel(b, c))
