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
for solver in (0, -2):
    for mask in (1, 5, 9, 13):
        e = BAEngine(n, m, windows=W)
        e.set_fusion(mask); e.set_solver(solver)
        for w in range(W):
            e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n, window=w)
            e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=w)
            e.set_states(g["states_out_9"][0], float(g["lamda_in"][10]), window=w)
        e.step(10, False)
        res[(solver, mask)] = (e.debug("dpose", window=5), e.get_states(window=5)[0], e.get_states(window=5)[2])
        e.close()
ref = res[(-2, 1)]
for k, v in res.items():
    d = np.abs(v[0] - ref[0]); 
    print(k, "dpose equal:", np.array_equal(v[0], ref[0]), "max abs diff", d.max(), "rel", d.max() / np.abs(ref[0]).max(), "first rows differing", np.nonzero(d.max(axis=1))[0][:5], "hess equal", np.array_equal(v[2], ref[2]))
