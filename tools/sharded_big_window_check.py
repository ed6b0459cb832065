#!/usr/bin/env python3
"""One-off check: a big single window (default CONFIG C5: 2004 poses / 500 000 rows -- the size at which a one-window handle
takes fusion mask 14) through the observation-sharded paths at ONE rank -- the caller-dispatched stages, the library-issued
call-by-call protocol and the library-issued chained schedule -- against the unsharded engine: trial counts and dampings
equal, states to 1e-9 after every call of the 20-call schedule."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 300))
import numpy as np
import torch
import torch.distributed as dist
from vinsat_amd import od_pipe, synth
from vinsat_amd.dist import ShardedBA
from vinsat_amd.engine import BAEngine
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
win = od_pipe.prepare_window(*synth.make_sequence(os.environ.get("CONFIG", "C5")))
st0 = od_pipe.initial_guess(win)
n, m = win.time_idx.size, win.ii.size
e = BAEngine(n, m)
e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
e.set_states(st0, 1e-4)
ref = []
for k in range(20):
    e.step(k, k < 10)
    ref.append(e.get_states())
e.close()
for name, native, chained in (("stages", False, False), ("library, call by call", True, False), ("library, chained", True, True)):
    sba = ShardedBA.from_window(win, device=0, native=native)
    sba.set_states(st0, 1e-4)
    worst = 0.0
    if chained:
        sba.run_schedule(list(range(20)), [k < 10 for k in range(20)])
        got = sba.get_states()
        worst = float(np.abs(got[0] - ref[19][0]).max() / np.abs(ref[19][0]).max())
        assert got[1] == ref[19][1], (name, got[1], ref[19][1])
    else:
        for k in range(20):
            sba.step(k, k < 10)
            got = sba.get_states()
            err = float(np.abs(got[0] - ref[k][0]).max() / np.abs(ref[k][0]).max())
            worst = max(worst, err)
            assert got[1] == ref[k][1] and got[3] == ref[k][3], (name, k, got[1], ref[k][1], got[3], ref[k][3])
    assert worst < 1e-9, (name, worst)
    print(f"{name}: ok, worst relative state difference {worst:.2e}", flush=True)
    sba.close()
dist.destroy_process_group()
