#!/usr/bin/env python3
"""One-off stress of the carried-keys sharded protocol at one RCCL rank: random windows (tests/random_windows.py), the schedule
chained (vba_sh_run_schedule, twice: the second starts from carried keys) and call by call, against the unsharded engine:
dampings and trial counts exact, states to 1e-7.  usage: tools/stress_sharded.py first last"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
import numpy as np, torch, torch.distributed as dist
from random_windows import SCHEDULE, make
from vinsat_amd.dist import HipStageEngine, ShardedBA
from vinsat_amd.engine import BAEngine
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
iters, inits = [s[0] for s in SCHEDULE], [s[1] for s in SCHEDULE]
for seed in range(a, b):
    win, xyz, uv, ii, conf, t, st0 = make(seed, long_gaps=bool(os.environ.get("STRESS_LONG")))
    n, m = t.size, ii.size
    try:
        def engine():
            e = BAEngine(max(n, 2), max(m, 1))
            e.upload_observations(xyz, uv, conf, ii, n); e.upload_window(win.intrinsics, win.cumrot_last, t)
            return e
        ref = engine(); ref.set_states(st0, 1e-4)
        outs = []
        for rep in range(2):
            for it, init in SCHEDULE:
                ref.step(it, init); outs.append(ref.get_states())
        ref.close()
        res = {}
        for name, chained in (("chained", True), ("stepped", False)):
            stg = HipStageEngine(engine(), torch_stream=False); stg.attach_rccl(dist)
            sb = ShardedBA(stg, n, m, m); sb.set_states(st0, 1e-4)
            mids = []
            for rep in range(2):
                if chained:
                    sb.run_schedule(iters, inits); mids.append(sb.get_states())
                else:
                    for it, init in SCHEDULE:
                        sb.step(it, init)
                    mids.append(sb.get_states())
            res[name] = mids; stats = stg.stats(); sb.close()
        for rep in range(2):
            r = outs[6 * rep + 5]
            for name in res:
                s = res[name][rep]
                err = np.abs(s[0] - r[0]).max() / np.abs(r[0]).max()
                assert err < 1e-5 and (err > 1e-9 or (s[1] == r[1] and s[3] == r[3])), (name, rep, err, s[1], r[1], s[3], r[3])
            assert np.array_equal(res["chained"][rep][0], res["stepped"][rep][0]), ("chained != stepped", rep)
        print(f"seed {seed}: n {n} m {m} ok (misses {stats[1]}, lm {stats[2]})", flush=True)
    except Exception as ex:
        bad.append(seed)
        print(f"seed {seed} n {n} m {m}: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
print("failures:", bad)
dist.destroy_process_group()
sys.exit(1 if bad else 0)
