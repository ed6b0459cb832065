#!/usr/bin/env python3
"""Benchmark of the BA hot path on MI355X: BA iterations/sec on the 500-pose / 50 000-observation window.

    python bench.py --gpus N --steps K --warmup W

One "step" is one call of BA() (reference estimation/BA/BA_filtering.py:4-98) through the C ABI with every
input already resident in HBM.  The steps walk the reference driver's own schedule (od_pipe.py:1036-1040):
20 calls per window -- 10 landmark-only (initialize=True) then 10 full -- restarting from the perturbed
initial guess after each 20, so the mix of kernels is the one the reference's 20-iteration loop executes.
The 20 calls of a schedule are issued with one host call (vba_run_schedule) and chained on the device.

N = 1 : one window (BASELINE.json configs[2], the headline config) on one GPU -- a latency-bound chain.
N > 1 : one process per GPU (torchrun / torch.distributed, backend nccl = RCCL); every rank runs its own
        500/50k window (the reference's outer loop over sequences, od_pipe.py:1069-1077): weak scaling, no
        data-path collective, value = N windows' iterations / max-over-ranks time.  The observation-sharded
        mode (landmarks of ONE window split over the ranks, three all-gathers per call) is measured in the
        same run and reported under "sharded".

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel of the timed run, HIP-event
timings on the library's stream), "cpu_baseline" (the NumPy oracle on this box's host, bounded sample),
"batched" (W windows per launch: the HBM-bound regime), "kernels_ms" (per-kernel averages).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# Algorithmic bytes per unit, per kernel class (SURVEY.md section 8d itemisation):
#   per observation: inputs 56 B (xyz 24 + uv 16 + conf 8 + ii 8), |r| 16 B, weight 8 B
#   "residual" runs only on a call whose states were replaced by the host; otherwise the previous call's trial
#   kernel has left the keys behind (trial = 56 in + 8 weight + 16 keys out)
#   per pose: see DESIGN.md table (bands 1944, rhs 72, X/z 1440 written+read, ...)
ALG_BYTES = {
    "residual": lambda n, m: 72 * m,
    "select": lambda n, m: 16 * m,
    "accumulate": lambda n, m: 64 * m + 216 * n,
    "dynamics": lambda n, m: (80 + 32 + 8 + 288 + 48 + 48 + 8 + 24 + 216) * n,
    "assemble": lambda n, m: (216 + 2 * 288 + 96 + 24 + 216 + 1944 + 72) * n,
    "solve": lambda n, m: (1944 + 72 + 2 * 1440 + 144 + 160) * n,
    "trial": lambda n, m: 80 * m + (80 + 32 + 8) * n,
    "decide": lambda n, m: 160 * n,
    "begin": lambda n, m: 160 * n,
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--windows", type=int, default=4096, help="windows per launch of the batched series (0 = skip)")
    ap.add_argument("--batched-steps", type=int, default=40)
    ap.add_argument("--profile-tag", default="r01", help="profiles/<tag>_w1_traffic.json supplies roofline.traffic")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-sharded", action="store_true")
    return ap.parse_args()


def schedule(k):
    j = k % 20
    return j, j < 10


def run_steps(eng, st0, nsteps, windows=1):
    """nsteps BA() calls walking the 20-call schedule; each schedule is issued through vba_run_schedule, i.e. the
    driver's loop `for iter in range(20): BA(iter, ...)` (od_pipe.py:1036-1040) as one host call whose calls are
    chained on the device (bit-identical to call-by-call stepping, tests/test_gpu_parity.py)."""
    k = 0
    while k < nsteps:
        cnt = min(20, nsteps - k)
        eng.set_states(st0, 1e-4, window=-1 if windows > 1 else 0)
        calls = [schedule(j) for j in range(cnt)]
        eng.run_schedule([c[0] for c in calls], [c[1] for c in calls])
        k += cnt


def cpu_baseline(win, st0, budget_s):
    """The NumPy oracle (a structure-aware CPU port, NOT the reference's dense autograd path) on this host."""
    from oracle import ba_oracle as O
    t_end = time.perf_counter() + budget_s
    calls = 0
    t0 = time.perf_counter()
    while True:
        st, lam = st0, 1e-4
        for it in range(20):
            st, lam, _, _ = O.ba_iteration(it, st, win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii, win.time_idx,
                                           win.intrinsics, win.confidences, lam, initialize=it < 10)
            calls += 1
        if time.perf_counter() > t_end:
            break
    dt = time.perf_counter() - t0
    return {"value": calls / dt, "unit": "BA iterations/s", "cores": 1, "kind": "port",
            "sample": f"{calls} BA calls ({calls // 20} x the 20-call schedule) of the same 500/50k window, NumPy fp64 oracle, "
                      f"{dt:.1f} s; the reference's own dense-autograd path measured in the build container (8 vCPU): "
                      "0.25 it/s over the same schedule (tests/golden/c3.npz ref_wall_seconds)"}


def main():
    args = parse()
    # Libraries (RCCL prints a version banner at init) must not add lines to stdout: everything written to fd 1
    # during the run goes to stderr, the ONE JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # VBA_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, collectives, sharded window) even with a
    # single rank -- the only way to rehearse it on a one-GPU box
    force_dist = os.environ.get("VBA_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from vinsat_amd import od_pipe, synth
    from vinsat_amd.engine import BAEngine

    cfg = synth.CONFIGS[args.config]
    det, orb = synth.make_sequence(cfg, seed=rank)          # every replica gets its own sequence
    win = od_pipe.prepare_window(det, orb)
    st0 = od_pipe.initial_guess(win)
    n, m = win.time_idx.size, win.ii.size
    eng = BAEngine(n, m, windows=1, device=local)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(eng, st0, args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(eng, st0, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * args.steps / dt

    # ---- per-kernel timing with HIP events on the library's stream (same steps, run right after the timed region)
    kern = {k: [] for k in BAEngine.KERNELS}
    phase = {"landmark_only": [], "full": []}
    nprof = min(args.steps, 40)
    for k in range(nprof):
        it, init = schedule(k)
        if it == 0:
            eng.set_states(st0, 1e-4)
        ms = eng.step_profiled(it, init)
        for name, v in ms.items():
            if v > 0:
                kern[name].append(v)
        phase["landmark_only" if init else "full"].append(sum(ms.values()))
    kernels_ms = {k: (float(np.mean(v)) if v else 0.0) for k, v in kern.items()}
    share = {k: kernels_ms[k] * len(kern[k]) for k in kern}
    dom = max(share, key=share.get)
    alg = float(ALG_BYTES[dom](n, m))
    achieved = alg / (kernels_ms[dom] * 1e-3) / 1e9
    # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/profile_pmc.sh + tools/summarize_pmc.py:
    # 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md); a kernel class sums its kernels
    KERNELS_OF = {"solve": ["k_solve_chunks<false>", "k_cr_level0<false>", "k_solve_reduced_cr<false, true>", "k_solve_reduced_cr<false, false>", "k_solve_chunks2<false>", "k_solve_reduced<false>",
                            "k_solve_recover2", "k_solve_recover", "k_solve_blockdiag<false>", "k_solve<false>", "k_solve_packed<false>"],
                  "select": ["k_select_pass<1, false, 8>", "k_select_pass<2, true, 8>"]}
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"{args.profile_tag}_w1_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            names = KERNELS_OF.get(dom, [k for k in tj if k.startswith("k_" + dom) or k.startswith("k_obs_" + dom)])
            calls = max([tj[k].get("calls", 0) for k in names if k in tj] or [0])
            vals = [tj[k]["hbm_bytes_per_launch"] * tj[k].get("calls", calls) / max(calls, 1) for k in names if k in tj and "hbm_bytes_per_launch" in tj[k]]
            traffic = float(sum(vals)) if vals else None
        except Exception:
            traffic = None
    if dom == "solve":
        dom_name = "solve: k_solve_chunks + k_cr_level0 + k_solve_reduced_cr + k_solve_recover"
    else:
        dom_name = "k_" + dom
    roofline = {"kernel": dom_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg,
                "avg_launch_ms": kernels_ms[dom], "launches_timed": len(kern[dom]),
                "note": "single window = a latency-bound chain (chunks of ceil(n/65) poses eliminated in parallel, then block "
                        "cyclic reduction over the separators: ~7 + 6 dependent 9x9 block steps); see "
                        "'batched' for the bandwidth regime"}

    # ---- the same calls through vba_iterate: states cross PCIe both ways on every call (never the headline value)
    host_roundtrip = None
    if rank == 0:
        stt, lam = st0, 1e-4
        eng.iterate(0, True, lam, stt)
        th = time.perf_counter()
        nh = 100
        for k in range(nh):
            it, init = schedule(k)
            if it == 0:
                stt, lam = st0, 1e-4
            stt, lam, _, _, _ = eng.iterate(it, init, lam, stt)
        dth = time.perf_counter() - th
        host_roundtrip = {"value": nh / dth, "unit": "BA iterations/s", "ms_per_call": 1e3 * dth / nh,
                          "note": "vba_iterate: 40 kB of states host->device and back, one host synchronisation per call"}

    # ---- batched windows: W independent windows per launch
    batched = None
    if args.windows > 0 and rank == 0 and world == 1 and not force_dist:
        W = args.windows
        be = BAEngine(n, m, windows=W, device=local)
        for w in range(W):
            be.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=w)
            be.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
        run_steps(be, st0, 20, windows=W)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        run_steps(be, st0, args.batched_steps, windows=W)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - tb
        bk = {k: [] for k in BAEngine.KERNELS}
        for k in range(20):
            it, init = schedule(k)
            if it == 0:
                be.set_states(st0, 1e-4, window=-1)
            for name, v in be.step_profiled(it, init).items():
                if v > 0:
                    bk[name].append(v)
        bms = {k: (float(np.mean(v)) if v else 0.0) for k, v in bk.items()}
        per_kernel = {k: {"ms": bms[k], "GBps": (ALG_BYTES[k](n, m) * W / (bms[k] * 1e-3) / 1e9) if bms[k] > 0 else 0.0}
                      for k in bms}
        bdom = max(bms, key=lambda k: bms[k] * len(bk[k]))
        # bytes and time of the average call of the 20-call schedule (a class counts for the calls it ran in)
        step_bytes = sum(ALG_BYTES[k](n, m) * len(bk[k]) / 20.0 for k in ALG_BYTES) * W
        step_ms = sum(bms[k] * len(bk[k]) / 20.0 for k in bms)
        batched = {"windows": W, "value": W * args.batched_steps / dtb, "unit": "BA iterations/s", "steps": args.batched_steps,
                   "ms_per_step": 1e3 * dtb / args.batched_steps, "dominant_kernel": "k_" + bdom,
                   "roofline": {"kernel": "k_" + bdom, "bound": "hbm", "achieved": per_kernel[bdom]["GBps"], "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": per_kernel[bdom]["GBps"] / HBM_PEAK_GBS},
                   "whole_step_GBps": step_bytes / (1e-3 * step_ms) / 1e9,
                   "whole_step_bytes_per_window": step_bytes / W,
                   "kernels": per_kernel}
        be.close()

    # ---- accuracy: the 20-call schedule once more from the initial guess, against the reference's final states
    accuracy = None
    gpath = os.path.join(ROOT, "tests", "golden", f"{cfg.name.lower()}.npz")
    if rank == 0 and os.path.exists(gpath):
        g = np.load(gpath)
        eng.set_states(g["states0"][0], 1e-4)
        for k in range(20):
            eng.step(*schedule(k))
        s_fin = eng.get_states()[0]
        ref = g["states_out_19"][0]
        dpos = np.linalg.norm(s_fin[:, :3] - ref[:, :3], axis=1)
        ang = 2 * np.arccos(np.clip(np.abs((s_fin[:, 3:7] * ref[:, 3:7]).sum(-1)), 0, 1))
        gt = g["in_poses_gt_eci"]
        accuracy = {"pose_rmse_vs_ref_km": float(np.sqrt((dpos ** 2).mean())), "max_rel_pos_err_vs_ref": float(np.abs(s_fin[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max()),
                    "max_attitude_err_vs_ref_rad": float(ang.max()),
                    "pos_rmse_vs_truth_km": float(np.sqrt(((s_fin[:, :3] - gt[:, :3]) ** 2).sum(1).mean())),
                    "reference": "states after call 19 of the reference's own run on the same inputs (tests/golden)"}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu = cpu_baseline(win, st0, args.cpu_seconds)

    emitted = []

    def emit(sharded):
        """Rank 0 prints the one JSON line (once)."""
        if rank != 0 or emitted:
            return
        emitted.append(1)
        out = {
            "metric": "BA iterations/sec (500 poses, 50k landmarks) + final pose RMSE vs ref",
            "value": value, "unit": "BA iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cfg.name}: {n}-pose / {m}-observation window, 20-call schedule (10 landmark-only + 10 full)"
                                   + (f", {world} independent windows (one per GPU)" if world > 1 else ""),
                       "poses": n, "observations": m, "windows_per_gpu": 1, "parallelism": f"replicas{world}"},
            "phase_ms": {k: float(np.mean(v)) if v else None for k, v in phase.items()},
            "kernels_ms": kernels_ms,
            "roofline": roofline,
            "host_roundtrip": host_roundtrip,
            "accuracy": accuracy,
            "cpu_baseline": cpu,
            "batched": batched,
            "sharded": sharded,
        }
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    # The secondary (sharded) measurement must never cost the headline line: if a collective hangs, a watchdog prints
    # the line without it and ends the process.
    import threading

    def bail():
        emit({"error": "sharded measurement timed out"})
        os._exit(0)

    watchdog = threading.Timer(240.0, bail)
    watchdog.daemon = True
    if world > 1 or force_dist:
        watchdog.start()
    # ---- observation-sharded mode (N > 1): ONE window whose rows are split over the ranks
    sharded = None
    if (world > 1 or force_dist) and not args.no_sharded:
      try:
        from vinsat_amd.dist import ShardedBA
        cfg_s = synth.WindowConfig("sharded", cfg.n_poses, cfg.obs_per_pose * world, cfg.stride)
        det_s, orb_s = synth.make_sequence(cfg_s, seed=0)
        win_s = od_pipe.prepare_window(det_s, orb_s)
        sba = ShardedBA.from_window(win_s, device=local)
        st_s = od_pipe.initial_guess(win_s)
        for k in range(20):
            it, init = schedule(k)
            if it == 0:
                sba.set_states(st_s, 1e-4)
            sba.step(it, init)
        barrier()
        ts = time.perf_counter()
        ns = min(args.steps, 100)
        for k in range(ns):
            it, init = schedule(k)
            if it == 0:
                sba.set_states(st_s, 1e-4)
            sba.step(it, init)
        barrier()
        dts = time.perf_counter() - ts
        t = torch.tensor([dts], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sharded = {"value": ns / float(t.item()), "unit": "BA iterations/s", "poses": cfg.n_poses,
                   "observations_total": int(win_s.ii.size), "observations_per_rank": int(win_s.ii.size // world),
                   "collectives_per_call": "3 all-gathers (|r| keys, per-pose blocks, trial sums) over RCCL"}
        sba.close()
      except Exception as exc:      # never lose the headline line to the secondary measurement
        sharded = {"error": repr(exc)[:300]}

    emit(sharded)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    watchdog.cancel()


if __name__ == "__main__":
    main()
