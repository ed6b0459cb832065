#!/usr/bin/env python3
"""Diagnostic (-DVBA_RESIDENT_STAMPS build, VBA_LIB): wall-clock stamps of thread 0 of observation block 100 along k_trial of
a landmark-only and of a full C3 call: entry, prologue done, row leaders known, trial states of the block's poses formed,
rows reprojected + keys binned, bin reservations requested, first block sum, reserved bases there, keys in their buckets;
us since entry."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ctypes import byref, c_int64
from vinsat_amd import od_pipe, synth, _lib
from vinsat_amd.engine import BAEngine, _p
det, orb = synth.make_sequence(os.environ.get("VBA_CONFIG", "C3"))
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m)
e.set_pipeline(0)
e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
e.set_states(st0, 1e-4)
names = ["entry", "prologue", "leaders", "states formed", "rows + keys", "reservations requested", "block sum", "bases there", "keys stored"]
for it in range(13):
    e.step(it, it < 10)
    if it not in (5, 6, 11, 12):
        continue
    out = np.empty(64)
    cnt = c_int64()
    _lib.check(e.lib.vba_debug_fetch(e.h, 0, 102, _p(out), out.size, byref(cnt)), e.lib)
    t = out.view(np.uint64).astype(np.int64)
    print(f"call {it}:", "  ".join(f"{nm} {(t[i] - t[0]) * 0.01:.2f}" for i, nm in enumerate(names)), flush=True)
    an = ["entry", "loads requested", "median there", "rows done", "sums rotated", "reduced + stored", "end"]
    print(f"   accumulation, block 60:", "  ".join(f"{nm} {(t[16 + i] - t[16]) * 0.01:.2f}" for i, nm in enumerate(an)), flush=True)
e.close()
