#!/usr/bin/env python3
"""Step the C2 fixture call by call for a few confidence settings and print what every call reports (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_inputs, load_golden
from vinsat_amd.engine import BAEngine
g = load_golden("c2"); inp = golden_inputs(g)
n, m = inp["K"].shape[0], inp["xyz"].shape[0]
confs = [inp["conf"], np.full_like(inp["conf"], 3.0), np.where(inp["ii"] % 3 == 0, -0.5, inp["conf"]), inp["conf"] * 0.9]
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for ci, conf in enumerate(confs):
    if only >= 0 and ci != only: continue
    e = BAEngine(n, m); e.set_accumulate_lanes(lanes); e.set_warm_select(warm)
    e.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    e.set_states(g["states0"][0], 1e-4)
    for k in range(20):
        try:
            e.step(k, k < 10)
        except Exception as ex:
            print(f"conf {ci} call {k}: {ex}", flush=True)
            print("   scalars", e.debug("scalars")[:8], "misses", e.warm_select_misses(), "fallbacks", e.solver_fallbacks(), flush=True)
            break
        s, lam, hess, ntr, flags = e.get_states()
        print(f"conf {ci} call {k}: ntr {ntr} flags {flags} lam {lam:g} misses {e.warm_select_misses()} median {e.debug('scalars')[0]:.6g}", flush=True)
    e.close()
