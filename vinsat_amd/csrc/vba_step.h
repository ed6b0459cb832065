// vba_step.h -- the step of ONE pose from the solved system and its retraction (A7 tail + A8 head), as device functions.
//
// Latency mode (few windows) can do without a kernel of its own for this (VBA_OPT_FUSION bit 0): the trial kernel forms
// the step of every pose it needs itself -- each observation block for the handful of poses its rows belong to, the
// pose-chain blocks for the poses of their edges (16 lanes per pose either way) -- so the solve's recovery launch (full
// phase) and the assembly + solve launch (landmark-only phase, a 6x6 system per pose) disappear from the chain.  Both
// forms are pure functions of data that earlier kernels left in memory, so every block that forms the step of a pose gets
// the same bits.
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

// The same two forms of the step with the 16 lanes of a group working on ONE pose (l16 = lane inside the group, gbase =
// wave lane of its lane 0): the serial versions above cost every wave that runs them their full instruction count, and
// in latency mode instruction issue of a single wave IS the time.  Same arithmetic per element, so the same bits.
//   landmark-only: lane c < 7 owns column c of [A | b] (Gauss-Jordan, pivot row broadcast from lane k);
//   recovery:      lane r < 9 forms row r of the step.
// d9 is returned to every lane of the group.  Returns false if a pivot check failed (landmark-only form).
template <int K>
__device__ __forceinline__ void step_blockdiag6_pivots(double (&a)[6], double d0, int l16, bool& ok) {
    if constexpr (K < 6) {
        const double piv = bcast_row16<K>(a[K]);
        if (l16 == K && !(piv > 1e-10 * d0)) ok = false;
        const double inv = step_fast_rcp(piv);
        double f[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) f[r] = (r != K) ? bcast_row16<K>(a[r]) : 0.0;
        a[K] *= inv;
#pragma unroll
        for (int r = 0; r < 6; ++r)
            if (r != K) a[r] -= f[r] * a[K];
        step_blockdiag6_pivots<K + 1>(a, d0, l16, ok);
    }
}

__device__ __forceinline__ bool step_blockdiag6_group(const double* H, const double* b, double inv_wmax, double lam32, int l16,
                                                      int gbase, double* d9) {
    double a[6];
    const int c = l16 < 7 ? l16 : 6;
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r] = c < 6 ? H[sym6(r, c)] * inv_wmax + (r == c ? lam32 : 0.0) : b[r] * inv_wmax;
    double d0 = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) d0 = (r == c) ? a[r] : d0;
    // (the 16 lanes of a group are a DPP row: bcast_row16<K> is one move per half where __shfl is an LDS crossbar round trip)
    bool ok = true;
    step_blockdiag6_pivots<0>(a, d0, l16, ok);
#pragma unroll
    for (int r = 0; r < 9; ++r) d9[r] = r < 6 ? bcast_row16<6>(a[r]) : 0.0;
    (void)gbase;
    return ok;
}

__device__ __forceinline__ void step_recover_group(int i, int n, int s, const double* csol, const double* xsep, int l16, int gbase,
                                                   double* d9) {
    const int c = i / s;
    const int P = (n + s - 1) / s;
    const int r = l16 < 9 ? l16 : 0;
    double v;
    if ((c < P - 1) && (i == (c + 1) * s - 1)) {
        v = xsep[(size_t)c * 9 + r];                // a separator: copied from the reduced solution
    } else {
        const double* so = csol + (size_t)i * 171;
        v = so[r];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double xl = c > 0 ? xsep[(size_t)(c - 1) * 9 + k] : 0.0;
            const double xr = c < P - 1 ? xsep[(size_t)c * 9 + k] : 0.0;
            v -= so[(1 + k) * 9 + r] * xl + so[(10 + k) * 9 + r] * xr;
        }
    }
    d9[0] = bcast_row16<0>(v); d9[1] = bcast_row16<1>(v); d9[2] = bcast_row16<2>(v);
    d9[3] = bcast_row16<3>(v); d9[4] = bcast_row16<4>(v); d9[5] = bcast_row16<5>(v);
    d9[6] = bcast_row16<6>(v); d9[7] = bcast_row16<7>(v); d9[8] = bcast_row16<8>(v);
    (void)gbase;
}

// The trial state of pose i of window w (BA_filtering.py:55-60), formed by a 16-lane group: MODE 1 landmark-only 6x6 solve,
// MODE 2 recovery of the partitioned solve.  o[10] (retracted state) and d9 (the step) are complete in lane 0 of the group;
// every lane of the group must call (shuffles), `live` = the group has a pose.  bad: bit 0 pivot check failed (any lane
// may carry it), bit 1 non-finite step.
template <int MODE>
__device__ __forceinline__ void pose_trial_state_group(const DevView& V, int w, int i, bool live, int l16, int gbase, double inv_wmax,
                                                       double lam32, double* o, double* d9, unsigned& bad) {
    const size_t sb = (size_t)w * V.n_max;
    const int ii = live ? i : 0;
    if (MODE == 1) {
        if (!step_blockdiag6_group(V.Hraw + (sb + ii) * 21, V.braw + (sb + ii) * 6, inv_wmax, lam32, l16, gbase, d9) && live) bad |= 1u;
    } else {
        step_recover_group(ii, V.n[w], V.chunk, V.csol + sb * 171, V.rx + (size_t)w * V.p_max * 9, l16, gbase, d9);
    }
    if (live && l16 == 0) {
#pragma unroll
        for (int r = 0; r < 9; ++r)
            if (!(fabs(d9[r]) <= 1.79e308)) bad |= 2u;
        retract(V.states + (sb + ii) * 10, d9, o);
    }
}

}  // namespace vba
