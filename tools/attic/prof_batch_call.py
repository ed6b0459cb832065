"""cProfile of the batched BA() per-call path (22 windows as lists of torch tensors): where the ~0.33 ms of a batched call go on the host -- vba_step ~150 us (device ~110), _batch_result ~150, _batch_hit ~100 before the identity fast path."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from vinsat_amd import od_pipe, synth, ba as ba_mod
win = od_pipe.prepare_window(*synth.make_sequence("C3"))
st0 = od_pipe.initial_guess(win)
n = win.time_idx.size
B = 22
imu1 = torch.zeros((1, n, 1, 10), dtype=torch.float64)
imu1[0, :, 0, 6:10] = torch.from_numpy(win.cumrot_last)
one = dict(states=torch.from_numpy(st0)[None], imu=imu1, uv=torch.from_numpy(win.landmarks_uv)[None], xyz=torch.from_numpy(win.landmarks_xyz)[None],
           intr=torch.from_numpy(win.intrinsics)[None], conf=torch.from_numpy(win.confidences))
rep = lambda key: [one[key]] * B
def loop():
    st_l, lam_l = rep("states"), [1e-4] * B
    for it in range(20):
        st_l, _, lam_l, _ = ba_mod.BA(it, st_l, None, rep("imu"), rep("uv"), rep("xyz"), [win.ii] * B, [win.time_idx] * B, rep("intr"), rep("conf"), 1e-3, 1e-3, lam_l, None, initialize=it < 10)
loop()
t0 = time.perf_counter(); loop(); print("us per call", 1e6 * (time.perf_counter() - t0) / 20)
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
