"""Result files of the OD driver and their consumer (host side).

The reference's ``__main__`` (``estimation/od_pipe.py:1063-1086``) loops over sequence files, calls
``streaming_version`` on each and saves two object arrays, ``errors.npy`` and ``times.npy`` (one entry per
sequence: position error in km and the matching time stamps in s); ``estimation/errors_eval.py:19-50`` turns
them into the time-to-<5 km statistics.  This module provides both ends with the same file formats.
"""
from __future__ import annotations

import argparse
import glob
import os

import numpy as np


def save_results(folder, errors, times):
    """Write errors.npy / times.npy exactly as od_pipe.py:1085-1086 does (object arrays, one row per sequence)."""
    np.save(os.path.join(folder, "errors.npy"), np.array([np.asarray(e) for e in errors], dtype=object), allow_pickle=True)
    np.save(os.path.join(folder, "times.npy"), np.array([np.asarray(t) for t in times], dtype=object), allow_pickle=True)


def time_to_error(errors, times, threshold_km=5.0):
    """Per sequence: the first time stamp at which the position error is below the threshold (NaN if never).

    Same rule as ``time_to_error_hist`` (errors_eval.py:26-31): ``times[i][argmax(errors[i] < thr)]``.
    """
    out = []
    for e, t in zip(errors, times):
        e, t = np.asarray(e, dtype=np.float64), np.asarray(t)
        below = e < threshold_km
        out.append(float(t[np.argmax(below)]) if below.any() else float("nan"))
    return np.array(out)


def _pairs(folder):
    for det in sorted(glob.glob(os.path.join(folder, "tmp_dets", "*_all_detections.npy"))):
        sid = os.path.basename(det).split("_")[0]
        orb = os.path.join(folder, "tmp_pose", f"{sid}_orbit_eci_zyxvecs.npy")
        if os.path.exists(orb):
            yield det, orb


def _resolve(ba):
    """``ba`` as a callable: a callable itself, None, or ``"module:attribute"`` (the form that travels to worker processes)."""
    if ba is None or callable(ba):
        return ba
    import importlib
    mod, _, attr = str(ba).partition(":")
    return getattr(importlib.import_module(mod), attr)


def _run_share(pairs, ba, batched, timing=None):
    """The sequences of ``pairs`` through the driver on the current device; returns (errors, times) per sequence."""
    from .od_pipe import prepared_runs, streaming_batched, streaming_version
    if batched:
        if ba is not None:
            raise ValueError("batched=True drives vinsat_amd.ba.BA_window itself")
        results = streaming_batched(list(pairs), timing=timing) if pairs else []       # (files are read by the preparing threads)
    else:
        # the next sequences are read and prepared on host threads while this one's BA calls run (`prep` in the timing is then
        # what the consumer still waited for)
        import time
        results = []
        it = prepared_runs(list(pairs), device=0 if ba is None else None)      # (the HIP BA in use: per-row preparation on its device)
        while True:
            t0 = time.perf_counter()
            run = next(it, None)
            if timing is not None:
                timing["prep"] = timing.get("prep", 0.0) + time.perf_counter() - t0
            if run is None:
                break
            results.append(streaming_version(ba=ba, timing=timing, run=run))
    errors, times = [], []
    for e, _, t in results:
        errors.append(e.detach().cpu().numpy())
        times.append(np.concatenate([np.atleast_1d(np.asarray(x)) for x in t]))
    return errors, times


def split_longest_first(sizes, workers):
    """Sequence indices per worker: longest first, each to the worker with the least rows so far (ties: the lowest worker).  The
    cost of a sequence is dominated by its rows (data preparation and the observation kernels are linear in them)."""
    load = [0] * workers
    share = [[] for _ in range(workers)]
    for k in sorted(range(len(sizes)), key=lambda k: (-sizes[k], k)):
        w = min(range(workers), key=lambda w: (load[w], w))
        share[w].append(k)
        load[w] += sizes[k]
    return [sorted(s_) for s_ in share]


def _run_replicas(pairs, devices, ba, batched, configure, timeout):
    """One fresh worker process per entry of ``devices`` (the reference's outer loop over sequence files, od_pipe.py:1069-1077, is
    embarrassingly parallel: SURVEY.md 8(e) "the shape that does scale").  THIS process does not touch a GPU for it: every worker is
    started as a child (`python -m vinsat_amd.errors_eval --worker job.json`) with HIP_VISIBLE_DEVICES narrowed to its device
    before anything in it initialises the GPU, runs its share with `_run_share` and leaves its results in a file; a worker
    that fails or outlives ``timeout`` fails the run (its PID is ended, nothing is retried)."""
    import json
    import pickle
    import subprocess
    import sys
    import tempfile
    import time
    def rows(path):
        try:
            return int(np.load(path, mmap_mode="r", allow_pickle=False).shape[0])       # (the header only)
        except ValueError:
            return int(np.load(path, allow_pickle=True).shape[0])
    sizes = [rows(det) for det, _ in pairs]
    share = split_longest_first(sizes, len(devices))
    visible = os.environ.get("HIP_VISIBLE_DEVICES")
    visible = [v for v in visible.split(",") if v] if visible else None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory(prefix="vba_replicas_") as tmp:
        procs = []
        for w, (dev, idx) in enumerate(zip(devices, share)):
            if not idx:
                procs.append(None)
                continue
            job = dict(pairs=[pairs[k] for k in idx], ba=ba if (ba is None or isinstance(ba, str)) else None, batched=bool(batched),
                       configure=configure or {}, out=os.path.join(tmp, f"w{w}.pkl"))
            with open(os.path.join(tmp, f"w{w}.json"), "w") as f:
                json.dump(job, f)
            env = dict(os.environ, HIP_VISIBLE_DEVICES=str(visible[dev] if visible else dev),
                       PYTHONPATH=os.pathsep.join([root] + [q for q in os.environ.get("PYTHONPATH", "").split(os.pathsep) if q]))
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            # (stderr into a file: a pipe nobody reads while another worker is waited for would fill up and block its writer)
            procs.append(subprocess.Popen([sys.executable, "-m", "vinsat_amd.errors_eval", "--worker", os.path.join(tmp, f"w{w}.json")],
                                          env=env, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, f"w{w}.err"), "w")))
        deadline = time.monotonic() + timeout
        failed = []
        for w, p in enumerate(procs):
            if p is None:
                continue

            def tail():
                try:
                    return open(os.path.join(tmp, f"w{w}.err")).read()[-400:]
                except OSError:
                    return ""
            try:
                p.wait(timeout=max(1.0, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
                failed.append((w, "timed out", tail()))
                continue
            if p.returncode != 0:
                failed.append((w, f"exit code {p.returncode}", tail()))
        if failed:
            raise RuntimeError("replica worker(s) failed: " + "; ".join(f"worker {w} on device {devices[w]}: {why}: {(err or '')[-400:]}"
                                                                          for w, why, err in failed))
        errors, times, stats = [None] * len(pairs), [None] * len(pairs), []
        for w, idx in enumerate(share):
            if not idx:
                continue
            with open(os.path.join(tmp, f"w{w}.pkl"), "rb") as f:
                res = pickle.load(f)
            for k, e, t in zip(idx, res["errors"], res["times"]):
                errors[k], times[k] = e, t
            stats.append(dict(worker=w, device=devices[w], sequences=len(idx), rows=sum(sizes[k] for k in idx), **res["timing"]))
    return errors, times, stats


def _worker(job_path):
    """Body of a replica worker process (see `_run_replicas`)."""
    import json
    import pickle
    import time
    with open(job_path) as f:
        job = json.load(f)
    if job["configure"]:
        from . import ba as ba_mod
        ba_mod.configure(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in job["configure"].items()})
    timing = {}
    t0 = time.perf_counter()
    errors, times = _run_share([tuple(p) for p in job["pairs"]], _resolve(job["ba"]), job["batched"], timing=timing)
    timing["wall"] = time.perf_counter() - t0
    with open(job["out"], "wb") as f:
        pickle.dump(dict(errors=errors, times=times, timing=timing), f)


def run_folder(folder, ba=None, batched=False, gpus=None, workers_per_gpu=1, configure=None, timeout=3600.0, stats=None):
    """Process every ``*_all_detections.npy`` / ``*_orbit_eci_zyxvecs.npy`` pair under ``folder`` (the layout of
    od_pipe.py:1064-1075: ``tmp_dets/`` and ``tmp_pose/``) and save errors.npy / times.npy next to them.

    ``batched=True``: the sequences are the batch dimension of ``BA`` -- every sequence's current batch is a window of ONE
    ragged handle and each kernel launch covers all of them (:func:`vinsat_amd.od_pipe.streaming_batched`) -- instead of
    one sequence after the other as the reference's loop (od_pipe.py:1069-1077) runs them.

    ``gpus``: None = this process, its current device.  An int N (devices 0 .. N-1) or a list of device indices (a device may be
    named more than once) = REPLICAS: the sequences are dealt longest-first to ``len(gpus) * workers_per_gpu`` fresh worker
    processes, one device each, no communication between them -- the reference's loop over sequence files spread over the GPUs
    of a node; the results are merged into the same two files in the folder's order.  Data preparation is host work of the
    worker that owns the sequence, so ``workers_per_gpu`` > 1 also hides it behind another worker's kernels.  ``ba`` must then
    be None or a ``"module:attribute"`` string (it has to be importable in the workers).  ``configure``: keyword arguments of
    :func:`vinsat_amd.ba.configure` applied in every worker (pinned handle settings give a sequence the same bits whichever
    worker and batch it lands in).  ``stats`` (a list) receives one dict per worker: device, sequences, rows, wall and phase times."""
    pairs = list(_pairs(folder))
    if gpus is None:
        if configure:
            from . import ba as ba_mod
            ba_mod.configure(**configure)
        timing = {}
        errors, times = _run_share(pairs, _resolve(ba), batched, timing=timing)
        if stats is not None:
            stats.append(dict(worker=0, device=None, sequences=len(pairs), **timing))
    else:
        devices = list(range(gpus)) if isinstance(gpus, int) else [int(d) for d in gpus]
        if not devices or workers_per_gpu < 1:
            raise ValueError("gpus must name at least one device and workers_per_gpu be >= 1")
        if ba is not None and not isinstance(ba, str):
            raise ValueError("with gpus=..., ba must be None or a 'module:attribute' string (it is imported in the worker processes)")
        devices = [d for d in devices for _ in range(workers_per_gpu)]
        errors, times, st = _run_replicas(pairs, devices, ba, batched, configure, timeout)
        if stats is not None:
            stats.extend(st)
    save_results(folder, errors, times)
    return errors, times


def main():
    ap = argparse.ArgumentParser(description="run the OD driver over a folder of simulated sequences")
    ap.add_argument("folder", nargs="?")
    ap.add_argument("--threshold-km", type=float, default=5.0)
    ap.add_argument("--batched", action="store_true", help="all sequences (of a worker) as windows of one device handle")
    ap.add_argument("--gpus", type=int, default=0, help="replicas: deal the sequences to worker processes on devices 0 .. N-1")
    ap.add_argument("--workers-per-gpu", type=int, default=1)
    ap.add_argument("--worker", default=None, help=argparse.SUPPRESS)      # internal: body of a replica worker
    a = ap.parse_args()
    if a.worker:
        _worker(a.worker)
        return
    if not a.folder:
        ap.error("folder is required")
    stats = []
    errors, times = run_folder(a.folder, batched=a.batched, gpus=a.gpus or None, workers_per_gpu=a.workers_per_gpu, stats=stats)
    tt = time_to_error(errors, times, a.threshold_km)
    print(f"{len(errors)} sequences; time to <{a.threshold_km} km: median {np.nanmedian(tt):.0f} s, "
          f"{int(np.isnan(tt).sum())} never")
    for s_ in stats:
        print("  " + ", ".join(f"{k} {v:.3f}" if isinstance(v, float) else f"{k} {v}" for k, v in s_.items()))


if __name__ == "__main__":
    main()
