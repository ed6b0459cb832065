#!/usr/bin/env python3
"""Time of the 20-call schedule of one window for a range of chunk sizes of the partitioned solve (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
det, orb = synth.make_sequence(cfg)
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m)
e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
iters, inits = list(range(20)), [k < 10 for k in range(20)]
if os.environ.get("VBA_LANES"):
    e.set_accumulate_lanes(int(os.environ["VBA_LANES"]))
if os.environ.get("VBA_FUSION"):
    e.set_fusion(int(os.environ["VBA_FUSION"]))
for chunk in [int(x) for x in sys.argv[2:]] or [-1, 4, 5, 6, 7, 8, 10, 12]:
    if chunk > 0:
        e.set_solver(chunk, -1)
    else:
        e.set_solver(-1)
    for rep in range(3):
        e.set_states(st0, 1e-4)
        e.run_schedule(iters, inits)
    t0 = time.perf_counter()
    for rep in range(10):
        e.set_states(st0, 1e-4)
        e.run_schedule(iters, inits)
    dt = (time.perf_counter() - t0) / 200
    print(f"{cfg} chunk {chunk:3d}: {1e6 * dt:7.1f} us per call, {1 / dt:8.0f} it/s", flush=True)
e.close()
