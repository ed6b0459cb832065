"""Free-landmark bundle adjustment with a Schur-complement solve on the GPU -- ADD-ON, PARITY UNPINNED.

The reference (CMUAbstract/VINSat) keeps its landmarks fixed (``estimation/BA/BA_filtering.py:32-37``): there is nothing to
marginalise and no counterpart of this module in it.  This is the variant BASELINE.json's north_star describes on top of the
reference's reprojection model: the landmarks become unknowns (3 each, held by a catalogue prior ``N(X0, sigma^2 I)``), the
landmark blocks are eliminated (3x3 inversions), and the dense reduced camera system is factorised on the matrix cores
(``vinsat_amd/csrc/vba_schur.hip``).  It is validated against ``oracle/schur_oracle.py`` -- this repository's own CPU
restatement -- only, and it never runs inside :func:`vinsat_amd.ba.BA`.

Host side (this file): the static structure of a window -- rows sorted by landmark, the rows of each pose, and for every
6x6 block of the reduced system the list of row pairs that share a landmark -- is built once with NumPy and uploaded.
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_double, c_float, c_int, c_void_p

import numpy as np

from . import _lib
from ._lib import PD


def build_structure(pose_of_row, landmark_of_row, n, L):
    """Static index structure of a window (see ``include/vinsat_ba.h``, ``vba_schur_upload``).

    Returns ``(order, s)``: ``order`` sorts the caller's rows by (landmark, pose); ``s`` holds the int32 arrays ``lm_ptr,
    row_pose, row_lm, pose_ptr, pose_rows, blk_i, blk_j, blk_ptr, pair_k, pair_k2`` over the SORTED rows.
    """
    pose_of_row = np.asarray(pose_of_row, dtype=np.int64)
    landmark_of_row = np.asarray(landmark_of_row, dtype=np.int64)
    m = pose_of_row.size
    if landmark_of_row.size != m:
        raise ValueError("pose_of_row and landmark_of_row disagree on the number of rows")
    if m == 0 or pose_of_row.min() < 0 or pose_of_row.max() >= n or landmark_of_row.min() < 0 or landmark_of_row.max() >= L:
        raise ValueError("row indices out of range")
    order = np.lexsort((pose_of_row, landmark_of_row))
    rp, rl = pose_of_row[order], landmark_of_row[order]
    if np.any((rp[1:] == rp[:-1]) & (rl[1:] == rl[:-1])):
        raise ValueError("a landmark is observed twice from one pose")
    lm_ptr = np.zeros(L + 1, dtype=np.int64)
    np.add.at(lm_ptr, rl + 1, 1)
    lm_ptr = np.cumsum(lm_ptr)
    pose_rows = np.argsort(rp, kind="stable")
    pose_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(pose_ptr, rp + 1, 1)
    pose_ptr = np.cumsum(pose_ptr)
    # row pairs (k, k') of one landmark with pose(k) >= pose(k'): rows of a landmark are sorted by pose, so k >= k'
    cnt = np.diff(lm_ptr)
    npairs = int((cnt * (cnt + 1) // 2).sum())
    if npairs > 400_000_000:
        raise ValueError(f"{npairs} row pairs: tracks this long are not a bundle-adjustment window (check the visibility model)")
    ks, k2s = [], []
    for t in np.unique(cnt):
        if t == 0:
            continue
        starts = lm_ptr[:-1][cnt == t]
        a, b = np.tril_indices(int(t))
        ks.append((starts[:, None] + a[None, :]).ravel())
        k2s.append((starts[:, None] + b[None, :]).ravel())
    pk, pk2 = np.concatenate(ks), np.concatenate(k2s)
    bi, bj = rp[pk], rp[pk2]
    # every diagonal block exists (it holds B_i) even for a pose without rows
    key = bi * n + bj
    diag = np.arange(n, dtype=np.int64) * (n + 1)
    blocks = np.unique(np.concatenate([key, diag]))
    srt = np.argsort(key, kind="stable")            # pairs grouped by block, fixed order inside a block
    pk, pk2, key = pk[srt], pk2[srt], key[srt]
    blk_ptr = np.searchsorted(key, np.concatenate([blocks, [blocks[-1] + 1]]), side="left")
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    s = dict(lm_ptr=i32(lm_ptr), row_pose=i32(rp), row_lm=i32(rl), pose_ptr=i32(pose_ptr), pose_rows=i32(pose_rows),
             blk_i=i32(blocks // n), blk_j=i32(blocks % n), blk_ptr=i32(blk_ptr), pair_k=i32(pk), pair_k2=i32(pk2))
    return order, s


class SchurBA:
    """Device-resident free-landmark BA problem: ``n`` poses, ``L`` landmarks, ``m`` observation rows."""

    def __init__(self, states, landmarks0, uv, weights, pose_of_row, landmark_of_row, intrinsics, sigma_prior=0.05, device=0):
        self.lib = _lib.load()
        states = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 10)
        X0 = np.ascontiguousarray(landmarks0, dtype=np.float64).reshape(-1, 3)
        self.n, self.L = states.shape[0], X0.shape[0]
        self.order, s = build_structure(pose_of_row, landmark_of_row, self.n, self.L)
        self.m = self.order.size
        uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)[self.order]
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64).reshape(-1)[self.order])
        K = np.ascontiguousarray(intrinsics, dtype=np.float64).reshape(-1, 4)
        if K.shape[0] != self.n:
            raise ValueError("intrinsics need one row per pose")
        self.structure = s
        self.h = c_void_p()
        self._check(self.lib.vba_schur_create(device, self.n, self.m, self.L, int(s["blk_i"].size), int(s["pair_k"].size), byref(self.h)))
        u, v = np.ascontiguousarray(uv[:, 0]), np.ascontiguousarray(uv[:, 1])
        p = lambda a: a.ctypes.data_as(c_void_p)
        self._check(self.lib.vba_schur_upload(self.h, p(s["lm_ptr"]), p(s["row_pose"]), p(s["row_lm"]), p(u), p(v), p(w), p(s["pose_ptr"]),
                                              p(s["pose_rows"]), p(s["blk_i"]), p(s["blk_j"]), p(s["blk_ptr"]), p(s["pair_k"]), p(s["pair_k2"]),
                                              p(K), p(X0), float(sigma_prior)))
        self.set_state(states, X0)

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.vba_schur_last_error()
            raise _lib.VbaError(f"libvinsat_ba (schur) error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vba_schur_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, states, landmarks):
        s = np.ascontiguousarray(states, dtype=np.float64).reshape(self.n, 10)
        X = np.ascontiguousarray(landmarks, dtype=np.float64).reshape(self.L, 3)
        self._check(self.lib.vba_schur_set_state(self.h, s.ctypes.data_as(PD), X.ctypes.data_as(PD)))

    def get_state(self):
        s, X = np.empty((self.n, 10)), np.empty((self.L, 3))
        self._check(self.lib.vba_schur_get_state(self.h, s.ctypes.data_as(PD), X.ctypes.data_as(PD)))
        return s, X

    def iterate(self, lamda):
        """One LM trial; returns ``(cost_before, cost_after, accepted)``."""
        c0, c1, acc = c_double(), c_double(), c_int()
        self._check(self.lib.vba_schur_iterate(self.h, float(lamda), byref(c0), byref(c1), byref(acc)))
        return c0.value, c1.value, bool(acc.value)

    def last_info(self):
        """0 if the last factorisation went through, else 1 + the row at which it met a non-positive pivot."""
        v = c_int()
        self._check(self.lib.vba_schur_last_info(self.h, byref(v)))
        return v.value

    def last_ms(self):
        a, b, c = c_float(), c_float(), c_float()
        self._check(self.lib.vba_schur_last_ms(self.h, byref(a), byref(b), byref(c)))
        return dict(build=a.value, factor=b.value, solve=c.value)

    def last_step(self):
        """``(dc [n,6], dl [L,3])`` of the last iterate."""
        out = np.empty(6 * self.n + 3 * self.L)
        self._check(self.lib.vba_schur_debug_fetch(self.h, 0, out.ctypes.data_as(PD), out.size))
        return out[: 6 * self.n].reshape(self.n, 6), out[6 * self.n:].reshape(self.L, 3)

    def cholesky_factor(self):
        out = np.empty((6 * self.n, 6 * self.n))
        self._check(self.lib.vba_schur_debug_fetch(self.h, 1, out.ctypes.data_as(PD), out.size))
        return out

    def solve(self, lamda0=1e-4, max_iters=20, tol=1e-10):
        """Levenberg-Marquardt driver: damping x10 on a rejected trial, /10 on an accepted one.  Returns the cost history."""
        lam, hist = float(lamda0), []
        for _ in range(max_iters):
            c0, c1, ok = self.iterate(lam)
            hist.append((c0, c1, ok, lam))
            if ok:
                lam = max(lam * 0.1, 1e-9)
                if c0 - c1 <= tol * max(c0, 1e-300):
                    break
            else:
                lam *= 10.0
                if lam > 1e8:
                    break
        return hist
