#!/usr/bin/env python3
"""W sweep of the batched regime with the kernel set and the solver FORCED: where do the mode switches belong?

    python tools/mode_sweep.py [W ...]            one JSON line per (W, mode, solver)

mode: lat = latency-mode kernels (vba_create_mode 1), bw = bandwidth-mode kernels (0); solver: part = chunks of 8 poses +
cyclic reduction of the separators, seq = sequential walk (four windows per wavefront).  Timed: the chained 20-call
schedule (vba_run_schedule), C3 windows (500 poses / 50 000 rows), all windows alike.
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
from vinsat_amd.engine import BAEngine
from bench import run_steps


def main():
    Ws = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 15, 16, 22, 32, 64, 128, 256, 512, 1024, 4096]
    combos = os.environ.get("VBA_SWEEP_COMBOS", "lat:part,lat:seq,bw:part,bw:seq").split(",")
    cfg = os.environ.get("VBA_SWEEP_CONFIG", "C3")       # a BASELINE config, or "POSESxROWS_PER_POSE" (e.g. 500x200)
    if "x" in cfg:
        cfg = synth.WindowConfig(cfg, int(cfg.split("x")[0]), int(cfg.split("x")[1]), 5)
    det, orb = synth.make_sequence(cfg)
    win = od_pipe.prepare_window(det, orb)
    st0 = od_pipe.initial_guess(win)
    n, m = win.time_idx.size, win.ii.size
    for W in Ws:
        for combo in combos:
            mode, solver = combo.split(":")
            if mode == "lat" and W > 512:
                continue            # (33 MB of bin buckets per window)
            if mode == "bw" and W < 2:
                continue
            if mode == "auto":
                mode_arg = -1
            else:
                mode_arg = 1 if mode == "lat" else 0
            try:
                e = BAEngine(n, m, windows=W, mode=mode_arg)
                if solver == "part":
                    e.set_solver(int(os.environ.get("VBA_SWEEP_CHUNK", "8")), -1)
                elif solver == "seq":
                    e.set_solver(0)
                if os.environ.get("VBA_SWEEP_FUSION"):
                    e.set_fusion(int(os.environ["VBA_SWEEP_FUSION"]))
                if os.environ.get("VBA_SWEEP_TILES"):
                    e.set_trial_tiles(int(os.environ["VBA_SWEEP_TILES"]))
                if os.environ.get("VBA_SWEEP_LANES"):
                    e.set_accumulate_lanes(int(os.environ["VBA_SWEEP_LANES"]))
                if os.environ.get("VBA_SWEEP_CWAVES"):
                    e.set_chunk_waves(int(os.environ["VBA_SWEEP_CWAVES"]))
                for w in range(W):
                    e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
                    e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
                run_steps(e, st0, 20, windows=W)
                reps = 3 if W <= 256 else 1
                t0 = time.perf_counter()
                run_steps(e, st0, 20 * reps, windows=W)
                dt = time.perf_counter() - t0
                st = e.get_states(0)[0]
                print(json.dumps({"W": W, "mode": mode, "solver": solver, "chosen": e.mode(), "it_per_s": round(20 * reps * W / dt, 1),
                                  "ms_per_step": round(1e3 * dt / (20 * reps), 4), "chk": float(np.abs(st).sum())}), flush=True)
                e.close()
            except Exception as exc:
                print(json.dumps({"W": W, "mode": mode, "solver": solver, "error": repr(exc)[:200]}), flush=True)


if __name__ == "__main__":
    main()
