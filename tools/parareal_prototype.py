#!/usr/bin/env python3
"""NumPy prototype of the parallel-in-time propagation of a long gap (vinsat_amd/csrc/vba_long.hip): convergence table of the
iteration against the serial walk.  Fine propagator = the reference's chain of 1 s RK4 steps (BA_utils.py:73-87), coarse propagator
= one RK4 step per chunk; the correction sweep linearised around the current chain (c_{j+1} = (F_j(U_j) - U_{j+1}) + A_j c_j with
A_j the Jacobian of the coarse step), convergence by the chain's defect.  Same partition rule, tolerances and stopping rule as the
kernel; prints, per gap length, the defect after every fine pass and the error of the end state against the serial chain.

    python tools/parareal_prototype.py [gap seconds ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as O     # noqa: E402  (the oracle's RK4 and its sensitivity: tools are not the product)

TIGHT, LOOSE = 2.0 ** -48, 2.0 ** -45


def plan(s):
    """vba_math.h:long_plan."""
    L = 1
    while 45 * L * L < 34 * s:
        L += 1
    if 64 * L < s:
        L = (s + 63) // 64
    P = (s + L - 1) // L
    return L, P


def fine(x, n):
    for _ in range(n):
        x = O.rk4_step(x)
    return x


def rel_defect(F, N):
    return max(np.abs(F[:3] - N[:3]).max() / np.abs(N[:3]).max(), np.abs(F[3:] - N[3:]).max() / np.abs(N[3:]).max())


def parallel_in_time(x0, s, verbose=False):
    L, P = plan(s)
    lens = [L] * (P - 1) + [s - (P - 1) * L]
    U = [x0]
    for j in range(P):                                  # the coarse chain
        U.append(O.rk4_step(U[j], float(lens[j])))
    prev = 1.0
    for it in range(P + 1):
        F = [fine(U[j], lens[j]) for j in range(P)]     # in parallel on the device: a lane per chunk
        if it > 0:
            worst = max(rel_defect(F[j], U[j + 1]) for j in range(P))
            if verbose:
                print(f"   fine pass {it + 1}: defect {worst:.2e}")
            if worst <= TIGHT or (it >= 2 and worst <= LOOSE and worst > 0.25 * prev):
                break
            prev = worst
        A = [O.rk4_step_stm(U[j], np.eye(6), float(lens[j]))[1] for j in range(P)]     # in parallel: Jacobian of each coarse step
        c = np.zeros(6)
        for j in range(P):                              # the serial part: a matrix-vector product per chunk
            c = (F[j] - U[j + 1]) + A[j] @ c
            U[j + 1] = U[j + 1] + c
    return U[-1], it, P, L


def main():
    gaps = [int(a) for a in sys.argv[1:]] or [65, 100, 300, 510, 935, 1000, 3000, 6000]
    a = 6978.0
    v = np.sqrt(O.MU / a)
    x0 = np.concatenate([np.array([-a * 0.6, 0.0, a * 0.8]), np.array([0.8, 0.0, 0.6]) * v * 1.001])
    for s in gaps:
        ref = fine(x0, s)
        print(f"gap {s} s")
        xp, it, P, L = parallel_in_time(x0, s, verbose=True)
        ep = np.abs(xp[:3] - ref[:3]).max() / np.abs(ref[:3]).max()
        ev = np.abs(xp[3:] - ref[3:]).max() / np.abs(ref[3:]).max()
        print(f"   {P} chunks of {L} steps, {it} sweep(s); end state against the serial chain: position {ep:.1e}, velocity {ev:.1e} (relative)")


if __name__ == "__main__":
    main()
