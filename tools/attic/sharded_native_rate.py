#!/usr/bin/env python3
"""us per BA() call of the observation-sharded window at ONE rank with the exchanges issued by the library (vba_sh_call),
C3, the 20-call schedule (chained: vba_sh_run_schedule; VBA_SH_STEPPED=1: call by call; VBA_SH_PROTOCOL=0: the round-3 protocol); for A/B runs
of two builds (VBA_LIB) and kernel traces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 300))
import torch
import torch.distributed as dist
from vinsat_amd import od_pipe, synth
from vinsat_amd.dist import ShardedBA
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
win = od_pipe.prepare_window(*synth.make_sequence("C3"))
st0 = od_pipe.initial_guess(win)
sba = ShardedBA.from_window(win, device=0, native=True)
chained = os.environ.get("VBA_SH_STEPPED") != "1"      # default: the 20 calls as one chained schedule (vba_sh_run_schedule)
if os.environ.get("VBA_SH_PROTOCOL"):
    sba.engine.set_protocol(int(os.environ["VBA_SH_PROTOCOL"]))
def run(reps):
    for _ in range(reps):
        sba.set_states(st0, 1e-4)
        if chained:
            sba.run_schedule(list(range(20)), [k < 10 for k in range(20)])
        else:
            for k in range(20):
                sba.step(k, k < 10)
run(3)
t0 = time.perf_counter()
run(10)
print(f"{os.environ.get('VBA_LIB', 'default')}: {1e6 * (time.perf_counter() - t0) / 200:.2f} us per sharded call", flush=True)
sba.close()
dist.destroy_process_group()
