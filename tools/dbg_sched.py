import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import golden_inputs, load_golden
from vinsat_amd.engine import BAEngine
c2 = load_golden("c2"); g, inp = c2, golden_inputs(c2)
n, m = inp["K"].shape[0], inp["xyz"].shape[0]
confs = [inp["conf"], np.full_like(inp["conf"], 3.0), np.where(inp["ii"] % 3 == 0, -0.5, inp["conf"]), inp["conf"] * 0.9]
iters = list(range(20)); inits = [k < 10 for k in range(20)]
which = [int(x) for x in sys.argv[1:]] or [0,1,2,3]
e = BAEngine(n, m, windows=len(which)); e.set_accumulate_lanes(8)
for w, ci in enumerate(which):
    e.upload_observations(inp["xyz"], inp["uv"], confs[ci], inp["ii"], n, window=w)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=w)
    e.set_states(g["states0"][0], 1e-4, window=w)
try:
    print("trials", e.run_schedule(iters, inits), "misses", e.warm_select_misses())
except Exception as ex:
    print("ERR", ex)
