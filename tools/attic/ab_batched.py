import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
if "torch" in mode:
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
det, orb = synth.make_sequence("C3")
win = od_pipe.prepare_window(det, orb)
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
iters, inits = list(range(20)), [k < 10 for k in range(20)]
if "single" in mode:
    e1 = BAEngine(n, m)
    e1.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    e1.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    for r in range(10):
        e1.set_states(st0, 1e-4); e1.run_schedule(iters, inits)
    if "prof" in mode:
        for k in range(40):
            if k % 20 == 0: e1.set_states(st0, 1e-4)
            e1.step_profiled(k % 20, k % 20 < 10)
if "iterate" in mode:
    stt, lam = st0, 1e-4
    for k in range(100):
        if k % 20 == 0: stt, lam = st0, 1e-4
        stt, lam, _, _, _ = e1.iterate(k % 20, k % 20 < 10, lam, stt)
if "pyba" in mode:
    import torch
    from vinsat_amd.ba import BA
    imu = torch.zeros((1, n, 1, 10), dtype=torch.float64)
    imu[0, :, 0, 6:10] = torch.from_numpy(win.cumrot_last)
    uv_t, xyz_t = torch.from_numpy(win.landmarks_uv)[None], torch.from_numpy(win.landmarks_xyz)[None]
    intr_t, conf_t = torch.from_numpy(win.intrinsics)[None], torch.from_numpy(win.confidences)
    gt_t, vel_t = torch.from_numpy(win.poses_gt), torch.from_numpy(win.velocities)[None]
    s0_t = torch.from_numpy(st0)[None]
    for _ in range(3):
        states_t, lam_ = s0_t, 1e-4
        for it in range(20):
            states_t, _, lam_, _ = BA(it, states_t, vel_t, imu, uv_t, xyz_t, win.ii, win.time_idx, intr_t, conf_t, 1e-3, 1e-3, lam_, gt_t, initialize=it < 10)
W = 4096
e = BAEngine(n, m, windows=W)
for w in range(W):
    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
e.set_states(st0, 1e-4, window=-1); e.run_schedule(iters, inits)
t0 = time.perf_counter()
for r in range(2):
    e.set_states(st0, 1e-4, window=-1); e.run_schedule(iters, inits)
dt = (time.perf_counter() - t0) / 40
print(mode, f"{1e3*dt:.3f} ms per step", flush=True)
if "profafter" in mode:
    for k in range(20):
        if k == 0: e.set_states(st0, 1e-4, window=-1)
        e.step_profiled(k, k < 10)
    t0 = time.perf_counter()
    for r in range(2):
        e.set_states(st0, 1e-4, window=-1); e.run_schedule(iters, inits)
    dt = (time.perf_counter() - t0) / 40
    print(mode, "after profiled steps", f"{1e3*dt:.3f} ms per step", flush=True)
