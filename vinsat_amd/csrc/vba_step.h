// vba_step.h -- the step of ONE pose from the solved system and its retraction (A7 tail + A8 head), as device functions.
//
// Latency mode (few windows) has no kernel of its own for this: the trial kernel forms the step of every pose it needs
// itself -- each observation thread for its own pose (redundantly: the ~100 observations of a pose compute the same
// step), the pose-chain threads for the poses of their edges -- so the solve's recovery launch (full phase) and the whole
// solve launch (landmark-only phase, a 6x6 system per pose) disappear from the chain.  Both forms are pure functions of
// data that earlier kernels left in memory, so every thread that forms the step of a pose gets the same bits.
#pragma once

#include "vba_device.h"

namespace vba {

// 1/x to ~1 ulp from v_rcp_f64's seed: r (1 + e + e^2), e = 1 - x r (three dependent operations)
__device__ __forceinline__ double step_fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    const double t = fma(e, e, e);
    r = fma(r, t, r);
    return r;
}

// Landmark-only phase (BA_utils.py:463-466: no dynamics factor): the system is block diagonal and inside a pose the
// velocity rows carry only the damping, so the step of a pose solves (H_i / w_max + lam32 I) x = b_i / w_max (6x6,
// Gauss-Jordan without row exchanges, every pivot checked against the diagonal entry it started from as in the chain
// solver); d9[6..8] = 0.  H: packed upper triangle [21], b [6].  Returns false if a pivot check failed.
__device__ __forceinline__ bool step_blockdiag6(const double* H, const double* b, double inv_wmax, double lam32, double* d9) {
    double A[6][7], d0[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = 0; c < 6; ++c) A[a][c] = H[sym6(a, c)] * inv_wmax + (a == c ? lam32 : 0.0);
        A[a][6] = b[a] * inv_wmax;
        d0[a] = A[a][a];
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (!(A[k][k] > 1e-10 * d0[k])) ok = false;
        const double inv = step_fast_rcp(A[k][k]);
#pragma unroll
        for (int c = 0; c < 7; ++c) A[k][c] *= inv;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r != k) {
                const double f = A[r][k];
#pragma unroll
                for (int c = 0; c < 7; ++c) A[r][c] -= f * A[k][c];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) d9[r] = r < 6 ? A[r][6] : 0.0;
    return ok;
}

// Partitioned solve: x_i = yhat_i - Vhat_i x_left - What_i x_right for an interior block of chunk c = i / s, the reduced
// solution itself for a separator (csol: [n][19][9] chunk solutions, xsep: [separators][9]).
__device__ __forceinline__ void step_recover(int i, int n, int s, const double* csol, const double* xsep, double* d9) {
    const int c = i / s;
    const int P = (n + s - 1) / s;
    if ((c < P - 1) && (i == (c + 1) * s - 1)) {
#pragma unroll
        for (int r = 0; r < 9; ++r) d9[r] = xsep[(size_t)c * 9 + r];
        return;
    }
    const double* so = csol + (size_t)i * 171;
    double xl[9], xr[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        xl[r] = c > 0 ? xsep[(size_t)(c - 1) * 9 + r] : 0.0;
        xr[r] = c < P - 1 ? xsep[(size_t)c * 9 + r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        double v = so[r];
#pragma unroll
        for (int k = 0; k < 9; ++k) v -= so[(1 + k) * 9 + r] * xl[k] + so[(10 + k) * 9 + r] * xr[k];
        d9[r] = v;
    }
}

// The trial state of pose i of window w (BA_filtering.py:55-60): MODE 1 landmark-only 6x6 solve, MODE 2 recovery of the
// partitioned solve; o[10] = retracted state, d9 = the step.  bad: bit 0 pivot check failed, bit 1 non-finite step.
template <int MODE>
__device__ __forceinline__ void pose_trial_state(const DevView& V, int w, int i, double inv_wmax, double lam32, double* o,
                                                 double* d9, unsigned& bad) {
    const size_t sb = (size_t)w * V.n_max;
    if (MODE == 1) {
        if (!step_blockdiag6(V.Hraw + (sb + i) * 21, V.braw + (sb + i) * 6, inv_wmax, lam32, d9)) bad |= 1u;
    } else {
        step_recover(i, V.n[w], V.chunk, V.csol + sb * 171, V.rx + (size_t)w * V.p_max * 9, d9);
    }
#pragma unroll
    for (int r = 0; r < 9; ++r)
        if (!(fabs(d9[r]) <= 1.79e308)) bad |= 2u;
    retract(V.states + (sb + i) * 10, d9, o);
}

}  // namespace vba
