#!/usr/bin/env python3
"""Handles of very many windows: every window of a W-window handle (all windows alike) must end the 20-call schedule where a
4-window handle ends it.  python tools/big_batch_check.py CONFIG W [W ...]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
from bench import run_steps, load_windows

cfg = sys.argv[1]
win = od_pipe.prepare_window(*synth.make_sequence(cfg))
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
ref = None
for W in [4] + [int(x) for x in sys.argv[2:]]:
    e = BAEngine(n, m, windows=W, mode=0)
    e.set_solver(0)
    e.set_accumulate_lanes(8)
    load_windows(e, win, n, W)
    try:
        if os.environ.get("STEP") == "1":
            from bench import schedule
            e.set_states(st0, 1e-4, window=-1)
            for j in range(20):
                e.step(*schedule(j))
        else:
            run_steps(e, st0, 20, windows=W)
    except Exception as exc:
        print("failed:", repr(exc)[:150], flush=True)
    st, lam, _, nt, fl = e.get_states_all()
    st = np.asarray(st).reshape(W, n, 10)
    if ref is None:
        ref = st[0].copy()
    d = np.abs(st - ref[None]).reshape(W, -1).max(axis=1)
    bad = np.nonzero(d > 0)[0]
    print(json.dumps({"config": cfg, "W": W, "differing_windows": int(bad.size), "first_bad": [int(x) for x in bad[:6]], "max_diff": float(d.max()),
                      "lam": [float(np.min(lam)), float(np.max(lam))], "n_trials": [int(nt.min()), int(nt.max())], "flags": sorted(set(int(x) for x in fl))[:6], "misses": e.warm_select_misses()}), flush=True)
    e.close()
