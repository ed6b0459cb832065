#!/usr/bin/env python3
"""Diagnostic (-DVBA_RESIDENT_STAMPS build, VBA_LIB): wall-clock stamps inside k_solve_reduced_cr<., 2> of one C3 full call:
entry, fill done, then after every eliminate / fold of the level loop, every level of the back substitution, the level-1
recovery and the end; us since entry.  argv[1]: fusion mask (bit 7 set = without the LDS prefetch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ctypes import byref, c_int64
from vinsat_amd import od_pipe, synth, _lib
from vinsat_amd.engine import BAEngine, _p
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 15
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m)
e.set_fusion(mask)
e.set_pipeline(0)
e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
e.set_states(st0, 1e-4)
for it in range(13):
    e.step(it, it < 10)
    if it < 10:
        continue
    out = np.empty(128)
    cnt = c_int64()
    _lib.check(e.lib.vba_debug_fetch(e.h, 0, 101, _p(out), out.size, byref(cnt)), e.lib)
    t = out.view(np.uint64).astype(np.int64)
    k = int(np.argmax(t[:32])) + 1
    print(f"mask {mask} call {it} tail :", " ".join(f"{(x - t[0]) * 0.01:.2f}" for x in t[:k]), flush=True)
    print(f"   (fold done by wave 0, before the barrier):", " ".join(f"{(t[64 + i] - t[0]) * 0.01:.2f}" for i in range(k) if t[64 + i] > 0), flush=True)
    print("   (last fold of wave 0: entry, operands issued, operands there, results there, stored):",
          " ".join(f"{(t[i] - t[0]) * 0.01:.2f}" for i in (79, 80, 82, 84, 85)), flush=True)
    print("   (chunk 30, forming: start, columns formed, stored):", " ".join(f"{(t[i] - t[32]) * 0.01:.2f}" for i in (48, 50, 51)), flush=True)
    names = ["entry", "staged", "formed", "step0", "step1", "step2", "step3", "step4", "sweeps joined", "middle", "outward", "end"]
    idx = [32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 47]
    print(f"mask {mask} call {it} chunk 30:", " ".join(f"{nm} {(t[i] - t[32]) * 0.01:.2f}" for nm, i in zip(names, idx) if t[i] >= t[32]), flush=True)
e.close()
