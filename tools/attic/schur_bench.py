#!/usr/bin/env python3
"""Free-landmark Schur add-on (parity unpinned) at the headline window size: 500 poses, ~20 000 tracked landmarks.
Prints one JSON line: ms per phase of an LM trial, and the blocked Cholesky's flop rate against the fp64 matrix peak."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import synth
from vinsat_amd.schur import SchurBA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
d = synth.make_tracked_landmarks(n_poses=n, n_landmarks=L, seed=0)
rng = np.random.default_rng(1)
st = d["states_gt"].copy()
st[:, :3] += rng.normal(0, 2.0, (n, 3))
w = np.full(d["uv"].shape[0], 0.95)
t0 = time.perf_counter()
e = SchurBA(st, d["X0"], d["uv"], w, d["pose_of_row"], d["landmark_of_row"], d["intrinsics"], sigma_prior=d["sigma"])
t_setup = time.perf_counter() - t0
ms, hist = [], []
lam = 1e-4
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    t1 = time.perf_counter()
    c0, c1, ok = e.iterate(lam)
    wall = time.perf_counter() - t1
    m_ = e.last_ms(); m_["wall"] = 1e3 * wall
    ms.append(m_); hist.append((c0, c1, ok))
    lam = max(lam * 0.1, 1e-9) if ok else lam * 10
N = 6 * n
Np = (N + 255) // 256 * 256
fact = [x["factor"] for x in ms[1:]]
flops = Np ** 3 / 3.0
out = {"poses": n, "landmarks": int(d["X_true"].shape[0]), "rows": int(d["uv"].shape[0]), "reduced_system": N,
       "pairs": int(e.structure["pair_k"].size), "blocks": int(e.structure["blk_i"].size), "setup_s": t_setup,
       "ms": {k: float(np.mean([x[k] for x in ms[1:]])) for k in ms[0]},
       "cholesky_TFLOPs": flops / (np.mean(fact) * 1e-3) / 1e12, "fp64_matrix_peak_TFLOPs": 78.6,
       "cost": [h[0] for h in hist] + [hist[-1][1]], "accepted": [h[2] for h in hist], "parity": "unpinned (no counterpart in the reference)"}
print(json.dumps(out))
e.close()
