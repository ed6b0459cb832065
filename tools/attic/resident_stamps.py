#!/usr/bin/env python3
"""Diagnostic (library built with -DVBA_RESIDENT_STAMPS, VBA_LIB pointing at it): where the time of k_solve_resident goes.
One C3 window, one full call through the resident solve; prints, per role (chunk / cyclic-reduction group / tail), the 100 MHz
wall-clock stamps of wave 0 relative to the first block's entry: entry, wait over, body over, flag published."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ctypes import byref, c_int64
from vinsat_amd import od_pipe, synth, _lib
from vinsat_amd.engine import BAEngine, _p
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 79
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
for it in range(12):
    e.step(it, it < 10)
    if it < 10:
        continue
    out = np.empty((n + 8) * 4)
    cnt = c_int64()
    _lib.check(e.lib.vba_debug_fetch(e.h, 0, 100, _p(out), out.size, byref(cnt)), e.lib)
    t = out.view(np.uint64).reshape(-1, 4).astype(np.int64)
    P = (n + 7) // 8
    G = (P - 1 + 3) // 4
    t0 = t[:P + G + 1, 0][t[:P + G + 1, 0] > 0].min()
    rel = (t - t0) * 0.01          # us
    def show(name, rows):
        r = rel[rows]
        print(f"call {it} {name:6s} entry {r[:,0].min():6.2f}..{r[:,0].max():6.2f}  wait over {r[:,1].min():6.2f}..{r[:,1].max():6.2f}  "
              f"body over {r[:,2].min():6.2f}..{r[:,2].max():6.2f}  published {r[:,3].min():6.2f}..{r[:,3].max():6.2f}", flush=True)
    show("chunks", slice(0, P))
    show("groups", slice(P, P + G))
    if mask & 64:
        show("tail", slice(P + G, P + G + 1))
    hop = [rel[P + g, 1] - rel[4 * g:min(4 * g + 8, P), 3].max() for g in range(G)]
    print("   last producer's flag stored -> group's wait over, us:", " ".join(f"{x:.2f}" for x in hop))
    print("   chunk: body over (wave 0) -> flag stored, us: min %.2f median %.2f max %.2f" % tuple(np.percentile(rel[:P, 3] - rel[:P, 2], [0, 50, 100])))
    if mask & 64:
        print("   last group's flag stored -> tail's wait over, us: %.2f" % (rel[P + G, 1] - rel[P:P + G, 3].max()))
e.close()
