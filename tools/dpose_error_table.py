#!/usr/bin/env python3
"""Per solver variant: error of the solution of ONE solve (dpose) against the reference's dense LU (torch.linalg.solve,
BA_filtering.py:55) on every system captured in tests/golden (GPU box).  Relative to max |dpose_ref| of that system.
Also listed: the NumPy oracle's banded LU (LAPACK gbsv) on the same systems -- two LAPACK-grade factorizations of the
same ill-conditioned system (cond 1e10..1e14) already differ by that much.

    python tools/dpose_error_table.py > profiles/r02_dpose_error_by_variant.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_inputs, load_golden  # noqa: E402
from oracle import ba_oracle as O  # noqa: E402
from vinsat_amd.engine import BAEngine  # noqa: E402

VARIANTS = {"default (two-sided chunks + cyclic reduction)": (-1, None, False, 2), "chunks walked by one wave": (-1, None, False, 1),
            "sequential (one wave)": (0, None, False, 2), "default, pivoted": (-1, None, True, 2),
            "one wave per chunk, pivoted": (-1, None, True, 1), "sequential, pivoted": (0, None, True, 2),
            "chunk 7, one level": (7, 0, False, 2), "two-level 5/4": (5, 4, False, 2), "cyclic 3, pivoted": (3, -1, True, 2)}


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def main():
    table = {}
    for fx in ("c1", "c2", "rej"):
        g = load_golden(fx)
        inp = golden_inputs(g)
        n, m = inp["K"].shape[0], inp["xyz"].shape[0]
        calls = [k for k in range(20) if f"dpose_{k}" in g]      # the device keeps the LAST trial's solution: compare with the reference's last
        for name, (c1, c2, piv, waves) in VARIANTS.items():
            eng = BAEngine(n, m)
            if c2 is None:
                eng.set_solver(c1)
            else:
                eng.set_solver(c1, c2)
            eng.set_pivoting(piv)
            eng.set_chunk_waves(waves)
            eng.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
            eng.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
            errs = {}
            for k in calls:
                st_in = g[f"states_in_{k}"][0] if f"states_in_{k}" in g else (g["states0"][0] if k == 0 else g[f"states_out_{k-1}"][0])
                eng.iterate(int(g["iters"][k]), bool(g["initialize"][k]), float(g["lamda_in"][k]), st_in)
                errs[k] = rel(eng.debug("dpose"), g[f"dpose_{k}"][-1].reshape(n, 9))
            eng.close()
            table.setdefault(fx, {})[name] = {"max": max(errs.values()), "worst_call": max(errs, key=errs.get),
                                              "median": float(np.median(list(errs.values())))}
        errs = {}
        for k in calls:
            A, b = g[f"A_bands_{k}"][-1], g[f"JTr_{k}"][0].reshape(-1, 9)
            errs[k] = rel(O.solve_tridiag(A, b, "banded"), g[f"dpose_{k}"][-1].reshape(n, 9))
        table[fx]["LAPACK banded LU on the captured system (NumPy oracle)"] = {"max": max(errs.values()), "worst_call": max(errs, key=errs.get),
                                                                                "median": float(np.median(list(errs.values())))}
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
