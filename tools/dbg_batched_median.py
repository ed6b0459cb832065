import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_golden, golden_inputs
from vinsat_amd.engine import BAEngine
g = load_golden("c2"); inp = golden_inputs(g)
n, m = inp["K"].shape[0], inp["xyz"].shape[0]
W = 16
res = {}
for mode in ("exact", "warm49", "warm44"):
    e = BAEngine(n, m, windows=W)
    if mode == "exact": e.set_warm_select(0)
    else: e.set_warm_shift(int(mode[4:]))
    for w in range(W):
        e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n, window=w)
        e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=w)
        e.set_states(g["states0"][0], 1e-4, window=w)
    out = []
    for k in range(20):
        e.step(k, k < 10)
        sc = e.debug("scalars", window=3)
        a = np.abs(inp["uv"] - e.debug("est", window=3)).reshape(-1)
        med = np.sort(a)[(a.size - 1) // 2]
        out.append((sc[0], med, sc[1], sc[7]))
        if sc[0] != med:
            print(mode, "call", k, "median", sc[0], "host", med, "rank of device value", int((np.sort(a) < sc[0]).sum()), "wanted", (a.size - 1) // 2)
    res[mode] = (out, e.get_states(window=3))
    print(mode, "misses", e.warm_select_misses())
    e.close()
for mode in res:
    print(mode, "states equal to exact:", np.array_equal(res[mode][1][0], res["exact"][1][0]), [o[0] for o in res[mode][0]][:3])
