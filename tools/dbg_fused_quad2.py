import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_golden, golden_inputs
from vinsat_amd.engine import BAEngine
g = load_golden("c2"); inp = golden_inputs(g)
n, m = inp["K"].shape[0], inp["xyz"].shape[0]
W = 16
e = BAEngine(n, m, windows=W)
e.set_fusion(13); e.set_solver(0)
for w in range(W):
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n, window=w)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=w)
    e.set_states(g["states_out_9"][0], float(g["lamda_in"][10]), window=w)
e.step(10, False)
os.environ["VBA_DBG_RAW_BANDS"] = "1"
formed = e.debug("bands", window=5).copy()
formed_rhs = e.debug("rhs", window=5).copy()
del os.environ["VBA_DBG_RAW_BANDS"]
asm = e.debug("bands", window=5)
asm_rhs = e.debug("rhs", window=5)
print("rhs equal", np.array_equal(formed_rhs, asm_rhs), np.abs(formed_rhs - asm_rhs).max(), np.argwhere(formed_rhs != asm_rhs)[:10].tolist())
d = np.abs(formed - asm)
print("bands equal", np.array_equal(formed, asm), "max abs", d.max())
idx = np.argwhere(d > 0)
print("differing entries", len(idx), idx[:20].tolist())
for (i, b, r, c) in idx[:8]:
    print(i, b, r, c, formed[i, b, r, c], asm[i, b, r, c])
