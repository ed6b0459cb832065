#!/usr/bin/env python3
"""Diagnostic: the calls of the stress schedule on one random window (with long gaps), each from the oracle's state: where does a
GPU call leave the oracle -- factor (Phi, r_pred), system, step?  usage: diag_call.py SEED"""
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
def rel(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
e = BAEngine(n, m)
if os.environ.get("DIAG_SOLVER"):
    e.set_solver(int(os.environ["DIAG_SOLVER"]))
e.upload_observations(xyz, uv, conf, ii, n); e.upload_window(win.intrinsics, win.cumrot_last, t)
ref, lam = st0.copy(), 1e-4
D = np.array([1, 1, 1, 100.0, 100, 100])
for it, init in SCHEDULE:
    out, lam_g, hess, ntr, flags = e.iterate(it, init, lam, ref)
    new, lam_o, hess_o, ntr_o = O.ba_iteration(it, ref, win.cumrot_last, uv, xyz, ii, t, win.intrinsics, conf, lam, initialize=init)
    line = f"iter {it} init {init}: ntr {ntr}/{ntr_o} lam {lam_g}/{lam_o} flags {flags} states {rel(out, new):.2e}"
    if not init:
        r, E, F = O.orbit_factor(ref, t, jacobian=True)
        Phi = e.debug("Phi")[:-1]
        ph = max(rel(D[:, None] * Phi[i], np.concatenate([E[i][:, 0:3], E[i][:, 6:9]], -1)) for i in range(n - 1))
        rp = np.abs(e.debug("r_pred")[:, :6] - r).max()
        dp = e.debug("dpose")
        line += f" | Phi {ph:.1e} r_pred {rp:.1e} |dpose| max {np.abs(dp).max():.3e} step/state {np.abs(out - ref).max():.3e}"
    print(line)
    ref, lam = new, lam_o
