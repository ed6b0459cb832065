#!/usr/bin/env python3
"""One-off stress of LARGE ragged batches: 24 distinct random windows (tests/random_windows.py) dealt in random order over
300 .. 1400 window slots of ONE handle (partitioned solve up to 1023 windows, the four-windows-per-wave walk beyond; both
kernel sets where the latency-mode one is allowed), chained 6-call schedule, against one-window handles: trial counts and
dampings exact where the states agree to 1e-9, states to 1e-5; equal windows in different slots must agree bit for bit.
usage: tools/stress_big_batches.py first last"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from random_windows import SCHEDULE, make
from vinsat_amd.engine import BAEngine

a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    rng = np.random.default_rng(9000 + seed)
    W = int(rng.integers(300, 1401))
    wins = [make(100 * seed + k, long_gaps=bool(os.environ.get("STRESS_LONG"))) for k in range(24)]
    n_max = max(w[5].size for w in wins); m_max = max(w[3].size for w in wins)
    single = []
    for (win, xyz, uv, ii, conf, t, st0) in wins:
        n = t.size
        e = BAEngine(max(n, 2), max(ii.size, 1))
        e.upload_observations(xyz, uv, conf, ii, n); e.upload_window(win.intrinsics, win.cumrot_last, t)
        e.set_states(st0, 1e-4)
        for it, init in SCHEDULE:
            e.step(it, init)
        single.append(e.get_states()); e.close()
    deal = rng.integers(0, 24, size=W)
    for name, mode, solver in [("auto", -1, None), ("bw-walk", 0, 0), ("bw-part", 0, -1)]:
        try:
            e = BAEngine(n_max, m_max, windows=W, mode=mode)
            if solver is not None:
                e.set_solver(solver)
            for k in range(W):
                win, xyz, uv, ii, conf, t, st0 = wins[deal[k]]
                e.upload_observations(xyz, uv, conf, ii, t.size, window=k); e.upload_window(win.intrinsics, win.cumrot_last, t, window=k)
                e.set_states(st0, 1e-4, window=k)
            e.run_schedule([s[0] for s in SCHEDULE], [s[1] for s in SCHEDULE])
            st, lam, _, ntr, fl = e.get_states_all()
            first = {}
            for k in range(W):
                d = int(deal[k]); r = single[d]; n = r[0].shape[0]
                err = np.abs(st[k, :n] - r[0]).max() / np.abs(r[0]).max()
                assert err < 1e-5, (name, k, d, err)
                if err < 1e-9:
                    assert lam[k] == r[1] and ntr[k] == r[3], (name, k, d, lam[k], r[1], ntr[k], r[3])
                if d in first:
                    assert np.array_equal(st[k, :n], st[first[d], :n]) and lam[k] == lam[first[d]], (name, "slots differ", k, first[d], d)
                else:
                    first[d] = k
            print(f"seed {seed} W {W} {name}: mode {e.mode()} ok", flush=True)
            e.close()
        except Exception as ex:
            bad.append((seed, name))
            print(f"seed {seed} W {W} {name}: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
