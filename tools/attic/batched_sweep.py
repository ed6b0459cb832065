#!/usr/bin/env python3
"""Per-kernel time of one BA call for W batched C3 windows (diagnostic; prints one JSON line per W)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
from bench import ALG_BYTES, schedule

def main():
    Ws = [int(x) for x in sys.argv[1:]] or [64, 256, 1024]
    det, orb = synth.make_sequence("C3")
    win = od_pipe.prepare_window(det, orb)
    st0 = od_pipe.initial_guess(win)
    n, m = win.time_idx.size, win.ii.size
    for W in Ws:
        e = BAEngine(n, m, windows=W)
        if os.environ.get("VBA_SWEEP_SOLVER"):
            e.set_solver(int(os.environ["VBA_SWEEP_SOLVER"]))
        if os.environ.get("VBA_SWEEP_LANES"):
            e.set_accumulate_lanes(int(os.environ["VBA_SWEEP_LANES"]))
        for w in range(W):
            e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
            e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
        acc = {k: [] for k in e.KERNELS}
        by_phase = {True: {k: [] for k in e.KERNELS}, False: {k: [] for k in e.KERNELS}}
        for rep in range(2):
            for k in range(20):
                it, init = schedule(k)
                if it == 0:
                    e.set_states(st0, 1e-4, window=-1)
                ms = e.step_profiled(it, init)
                if rep:
                    for name, v in ms.items():
                        if v > 0:
                            acc[name].append(v)
                            by_phase[init][name].append(v)
        t0 = time.perf_counter()
        for k in range(20):
            it, init = schedule(k)
            if it == 0:
                e.set_states(st0, 1e-4, window=-1)
            e.step(it, init)
        dt = time.perf_counter() - t0
        out = {"W": W, "it_per_s": 20 * W / dt, "ms_per_step": 1e3 * dt / 20,
               "kernels": {k: {"ms": round(float(np.mean(v)), 4), "GBps": round(ALG_BYTES[k](n, m) * W / (np.mean(v) * 1e-3) / 1e9, 1)} for k, v in acc.items() if v}}
        out["landmark_only_ms"] = {k: round(float(np.mean(v)), 4) for k, v in by_phase[True].items() if v}
        out["full_ms"] = {k: round(float(np.mean(v)), 4) for k, v in by_phase[False].items() if v}
        print(json.dumps(out), flush=True)
        e.close()

if __name__ == "__main__":
    main()
