#!/usr/bin/env python3
"""Per-call wall time of vba_iterate_resident (C ABI through ctypes) with and without the speculative pipeline, C3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
for pipe in (True, False):
    e = BAEngine(n, m)
    e.set_pipeline(pipe)
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    def window():
        st, lam, *_ = e.iterate(0, True, 1e-4, st0)
        for it in range(1, 20):
            st, lam, *_ = e.iterate_resident(it, it < 10)
        return st
    window(); window()
    t0 = time.perf_counter()
    for r in range(10):
        window()
    dt = (time.perf_counter() - t0) / 200
    # phases
    ti = []
    for r in range(5):
        e.iterate(0, True, 1e-4, st0)
        for it in range(1, 20):
            t1 = time.perf_counter(); e.iterate_resident(it, it < 10); ti.append((it, time.perf_counter() - t1))
    init = [t for it, t in ti if it < 10]; full = [t for it, t in ti if it > 10]
    print(f"pipeline {pipe}: {1e3 * dt:.4f} ms per call; landmark-only {1e6 * sum(init) / len(init):.1f} us, full {1e6 * sum(full) / len(full):.1f} us; stats {e.pipeline_stats()}", flush=True)
    e.close()
