#!/usr/bin/env python3
"""Where a vinsat_amd.ba.BA() call of the reference's loop spends its time on the host: inside the library's resident call
(vba_iterate_resident: enqueue of the speculated next call + wait for this call's decision) against the Python around it.
C3, the 20-call loop, per phase (landmark-only calls 0..9, full calls 10..19)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vinsat_amd import od_pipe, synth, ba as ba_mod
from vinsat_amd.engine import BAEngine

win = od_pipe.prepare_window(*synth.make_sequence(os.environ.get("VBA_CONFIG", "C3")))
st0 = od_pipe.initial_guess(win)
n = win.time_idx.size
imu = torch.zeros((1, n, 1, 10), dtype=torch.float64)
imu[0, :, 0, 6:10] = torch.from_numpy(win.cumrot_last)
uv_t, xyz_t = torch.from_numpy(win.landmarks_uv)[None], torch.from_numpy(win.landmarks_xyz)[None]
intr_t, conf_t = torch.from_numpy(win.intrinsics)[None], torch.from_numpy(win.confidences)
s0_t = torch.from_numpy(st0)[None]
inside = [0.0, 0.0]
orig = BAEngine.iterate_resident
phase = [0]
def timed(self, it, init):
    t0 = time.perf_counter()
    r = orig(self, it, init)
    inside[phase[0]] += time.perf_counter() - t0
    return r
BAEngine.iterate_resident = timed
parts = {}
def wrap_part(name):
    f = getattr(ba_mod, name)
    parts[name] = [0.0, 0.0]
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        parts[name][phase[0]] += time.perf_counter() - t0
        return r
    setattr(ba_mod, name, g)
for nm in ("_engine_for", "_call", "_wrap", "_take_resident"):
    wrap_part(nm)
def loop(reps, acc):
    for _ in range(reps):
        s, lam = s0_t, 1e-4
        for it in range(20):
            phase[0] = 0 if it < 10 else 1
            t0 = time.perf_counter()
            s, _, lam, _ = ba_mod.BA(it, s, None, imu, uv_t, xyz_t, win.ii, win.time_idx, intr_t, conf_t, 1e-3, 1e-3, lam, None, initialize=it < 10)
            acc[phase[0]] += time.perf_counter() - t0
loop(2, [0.0, 0.0])
inside[:] = [0.0, 0.0]
for v in parts.values():
    v[:] = [0.0, 0.0]
tot = [0.0, 0.0]
reps = 10
loop(reps, tot)
for p, name in enumerate(("landmark-only", "full")):
    c = 10 * reps
    print(f"{name}: {1e6 * tot[p] / c:.1f} us per BA() call, {1e6 * inside[p] / c:.1f} inside vba_iterate_resident, {1e6 * (tot[p] - inside[p]) / c:.1f} Python around it", flush=True)
for nm, v in parts.items():
    print(f"   {nm}: {1e6 * v[0] / (10 * reps):.1f} / {1e6 * v[1] / (10 * reps):.1f} us (landmark-only / full; _call includes the library call)")
print("pipeline hits / discards:", ba_mod._cache["eng"].pipeline_stats())
