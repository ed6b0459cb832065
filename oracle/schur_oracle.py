"""CPU restatement of the free-landmark Schur-complement BA add-on -- TEST INFRASTRUCTURE ONLY, PARITY UNPINNED.

The reference (CMUAbstract/VINSat) keeps its landmarks fixed (``estimation/BA/BA_filtering.py:32-37``); it has no
free-landmark mode, hence no golden vector and no function to pin this file to.  What is restated here is this
repository's OWN formulation (``vinsat_amd/csrc/vba_schur.hip``): the reference's reprojection model and pose Jacobian
(``oracle/ba_oracle.py:landmark_project``, itself pinned to ``BA_utils.py:30-50``), landmarks as additional unknowns with a
catalogue prior, confidences as weights, dense NumPy linear algebra.  Only ``tests/`` import it.

Two routes to the same step are given so that the test-suite can check the algebra against itself: the FULL normal
equations over ``[poses | landmarks]`` solved densely, and the Schur-complement route the GPU takes.
"""
from __future__ import annotations

import numpy as np

from . import ba_oracle as O


def linearise(states, X, uv, w, pose_of_row, landmark_of_row, K):
    """Residuals and Jacobians of every row: r [m,2], Jc [m,2,6] (d uv / d [dp, dtheta]), Jl [m,2,3] (d uv / d X = -Jc[:, :, :3])."""
    est, J = O.landmark_project(states, X[landmark_of_row], K, pose_of_row, jacobian=True)
    return uv - est, J, -J[:, :, :3]


def cost(states, X, X0, uv, w, pose_of_row, landmark_of_row, K, sigma):
    est = O.landmark_project(states, X[landmark_of_row], K, pose_of_row, jacobian=False)
    r = uv - est
    return float((w[:, None] * r * r).sum() + ((X - X0) ** 2).sum() / sigma ** 2)


def normal_equations(states, X, X0, uv, w, pose_of_row, landmark_of_row, K, sigma, lamda):
    """Dense blocks of [[B, E], [E^T, C]] [dc; dl] = [v; wl] (descent convention: J = d est, r = uv - est)."""
    n, L = states.shape[0], X.shape[0]
    r, Jc, Jl = linearise(states, X, uv, w, pose_of_row, landmark_of_row, K)
    B = np.zeros((6 * n, 6 * n))
    C = np.zeros((3 * L, 3 * L))
    E = np.zeros((6 * n, 3 * L))
    v = np.zeros(6 * n)
    wl = np.zeros(3 * L)
    for k in range(uv.shape[0]):
        i, l = pose_of_row[k], landmark_of_row[k]
        ci, cl = slice(6 * i, 6 * i + 6), slice(3 * l, 3 * l + 3)
        B[ci, ci] += w[k] * Jc[k].T @ Jc[k]
        C[cl, cl] += w[k] * Jl[k].T @ Jl[k]
        E[ci, cl] += w[k] * Jc[k].T @ Jl[k]
        v[ci] += w[k] * Jc[k].T @ r[k]
        wl[cl] += w[k] * Jl[k].T @ r[k]
    B += lamda * np.eye(6 * n)
    C += (1.0 / sigma ** 2 + lamda) * np.eye(3 * L)
    wl -= (X - X0).reshape(-1) / sigma ** 2
    return B, C, E, v, wl


def step_full(B, C, E, v, wl):
    """Step from the full system (no elimination)."""
    n6 = B.shape[0]
    H = np.block([[B, E], [E.T, C]])
    d = np.linalg.solve(H, np.concatenate([v, wl]))
    return d[:n6], d[n6:]


def step_schur(B, C, E, v, wl):
    """Step by eliminating the landmarks: S = B - E C^-1 E^T, dense Cholesky of S; returns (dc, dl, S, chol(S))."""
    L3 = C.shape[0]
    Cinv = np.zeros_like(C)
    for l in range(L3 // 3):
        s = slice(3 * l, 3 * l + 3)
        Cinv[s, s] = np.linalg.inv(C[s, s])
    S = B - E @ Cinv @ E.T
    g = v - E @ Cinv @ wl
    Lc = np.linalg.cholesky(S)
    dc = np.linalg.solve(Lc.T, np.linalg.solve(Lc, g))
    dl = Cinv @ (wl - E.T @ dc)
    return dc, dl, S, Lc


def apply_step(states, X, dc, dl):
    n = states.shape[0]
    d9 = np.zeros((n, 9))
    d9[:, :6] = dc.reshape(n, 6)
    return O.retract(states, d9), X + dl.reshape(-1, 3)


def lm_trial(states, X, X0, uv, w, pose_of_row, landmark_of_row, K, sigma, lamda):
    """One LM trial as ``vba_schur_iterate`` runs it: (cost_before, cost_after, accepted, states', X', dc, dl)."""
    c0 = cost(states, X, X0, uv, w, pose_of_row, landmark_of_row, K, sigma)
    B, C, E, v, wl = normal_equations(states, X, X0, uv, w, pose_of_row, landmark_of_row, K, sigma, lamda)
    dc, dl, _, _ = step_schur(B, C, E, v, wl)
    s1, X1 = apply_step(states, X, dc, dl)
    c1 = cost(s1, X1, X0, uv, w, pose_of_row, landmark_of_row, K, sigma)
    ok = c1 < c0
    return c0, c1, ok, (s1 if ok else states), (X1 if ok else X), dc.reshape(-1, 6), dl.reshape(-1, 3)
