import os, sys
sys.path.insert(0, os.getcwd())
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m)
e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
e.set_states(st0, 1e-4)
for k in range(3): e.step(10 + k, False)
