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


def run_folder(folder, ba=None, batched=False):
    """Process every ``*_all_detections.npy`` / ``*_orbit_eci_zyxvecs.npy`` pair under ``folder`` (the layout of
    od_pipe.py:1064-1075: ``tmp_dets/`` and ``tmp_pose/``) and save errors.npy / times.npy next to them.

    ``batched=True``: the sequences are the batch dimension of ``BA`` -- every sequence's current batch is a window of ONE
    ragged handle and each kernel launch covers all of them (:func:`vinsat_amd.od_pipe.streaming_batched`) -- instead of
    one sequence after the other as the reference's loop (od_pipe.py:1069-1077) runs them."""
    from .od_pipe import streaming_batched, streaming_version
    errors, times = [], []
    if batched:
        if ba is not None:
            raise ValueError("batched=True drives vinsat_amd.ba.BA_window itself")
        seqs = [(np.load(det, allow_pickle=True), np.load(orb, allow_pickle=True)) for det, orb in _pairs(folder)]
        results = streaming_batched(seqs)
    else:
        results = [streaming_version(detections_file_name=det, orbit_file_name=orb, ba=ba) for det, orb in _pairs(folder)]
    for e, _, t in results:
        errors.append(e.detach().cpu().numpy())
        times.append(np.concatenate([np.atleast_1d(np.asarray(x)) for x in t]))
    save_results(folder, errors, times)
    return errors, times


def main():
    ap = argparse.ArgumentParser(description="run the OD driver over a folder of simulated sequences")
    ap.add_argument("folder")
    ap.add_argument("--threshold-km", type=float, default=5.0)
    ap.add_argument("--batched", action="store_true", help="all sequences as windows of one device handle")
    a = ap.parse_args()
    errors, times = run_folder(a.folder, batched=a.batched)
    tt = time_to_error(errors, times, a.threshold_km)
    print(f"{len(errors)} sequences; time to <{a.threshold_km} km: median {np.nanmedian(tt):.0f} s, "
          f"{int(np.isnan(tt).sum())} never")


if __name__ == "__main__":
    main()
