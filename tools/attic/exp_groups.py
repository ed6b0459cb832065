#!/usr/bin/env python3
"""W C3 windows as a group of G handles (vba_run_schedule_group: the walks of the members run beside the others' streaming
kernels) against one handle of W windows.  python tools/exp_groups.py W G [G ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine, BAEngineGroup
from bench import run_steps, load_windows

W = int(sys.argv[1])
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
ref = None
for G in [int(x) for x in sys.argv[2:]]:
    if G == 1:
        e = BAEngine(n, m, windows=W, mode=0)
        e.set_solver(0)
    else:
        e = BAEngineGroup(n, m, W, groups=G)
    load_windows(e, win, n, W)
    run_steps(e, st0, 20, windows=W)
    t0 = time.perf_counter()
    run_steps(e, st0, 40, windows=W)
    dt = time.perf_counter() - t0
    st = e.get_states(W - 1)[0]
    if ref is None:
        ref = st
    print(f"W {W} G {G}: {1e3 * dt / 40:.3f} ms per step, {40 * W / dt:.0f} it/s, same bits as the first: {np.array_equal(st, ref)}", flush=True)
    e.close()
