// vba_math.h -- per-observation / per-pose arithmetic of the BA iteration, fp64.
//
// Pure functions shared by the HIP kernels.  They compile for the host as well (plain C++), which the
// test-suite uses to check the formulas on a machine without a GPU; the product never runs them there.
//
// Reference formulas: estimation/BA/BA_utils.py (cited per function).  The reference differentiates with
// autograd; the closed forms here were checked against its output.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VBA_HD __host__ __device__ __forceinline__
#else
#define VBA_HD inline
#endif

namespace vba {

constexpr double kMu = 398600.4418;    // BA_utils.py:883
constexpr double kJ2c = 1.75553e10;    // BA_utils.py:883
constexpr double kZMin = 0.1;          // BA_utils.py:13
constexpr double kQuatCoeff = 100.0;   // BA_filtering.py:11
constexpr double kVelCoeff = 100.0;    // BA_filtering.py:12

// A product / a sum rounded on its own, never contracted into an fma: the build uses -ffp-contract=fast-honor-pragmas
// (HIP's default mode), under which the pragma below is obeyed also after inlining into a kernel.
VBA_HD double vba_mul(double a, double b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return a * b;
}
VBA_HD double vba_add(double a, double b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return a + b;
}

// index of (a,b), a<=b, in the packed upper triangle of a symmetric 6x6
VBA_HD int sym6(int a, int b) {
    if (a > b) { int t = a; a = b; b = t; }
    return a * 6 - (a * (a - 1)) / 2 + (b - a);
}

// ------------------------------------------------------------------------------------------------ pose
struct PoseCam {
    double t[3];    // position
    double R[9];    // camera->ECI rotation of the normalised quaternion, row major
    double fx, fy, cx, cy;
};

// R(q/|q|); BA_utils.py:1058 normalises before rotating.
VBA_HD void pose_camera(const double* s /*[10]*/, const double* K /*[4]*/, PoseCam& pc) {
    pc.t[0] = s[0]; pc.t[1] = s[1]; pc.t[2] = s[2];
    double x = s[3], y = s[4], z = s[5], w = s[6];
    // one division, four products (the reference divides each component, BA_utils.py:1058: same value to an ulp; the
    // four fp64 divisions were a quarter of the instructions of the per-observation kernels)
    // Every product and sum below is rounded on its own (vba_mul / vba_add: never contracted into an fma).  With
    // -ffp-contract=fast the compiler would otherwise be free to contract `a*b + c*d` one way in one kernel and another way
    // in the next, and the |r| keys a trial kernel leaves behind must carry the bits the next call's own reprojection
    // produces.  Operation by operation this is also what the oracle's NumPy expressions do (oracle/ba_oracle.py).
    const double inv = 1.0 / sqrt(vba_add(vba_add(vba_add(vba_mul(x, x), vba_mul(y, y)), vba_mul(z, z)), vba_mul(w, w)));
    x = vba_mul(x, inv); y = vba_mul(y, inv); z = vba_mul(z, inv); w = vba_mul(w, inv);
    const double xx = vba_mul(x, x), yy = vba_mul(y, y), zz = vba_mul(z, z);
    const double xy = vba_mul(x, y), xz = vba_mul(x, z), yz = vba_mul(y, z), xw = vba_mul(x, w), yw = vba_mul(y, w), zw = vba_mul(z, w);
    pc.R[0] = vba_add(1.0, -vba_mul(2.0, vba_add(yy, zz))); pc.R[1] = vba_mul(2.0, vba_add(xy, -zw)); pc.R[2] = vba_mul(2.0, vba_add(xz, yw));
    pc.R[3] = vba_mul(2.0, vba_add(xy, zw)); pc.R[4] = vba_add(1.0, -vba_mul(2.0, vba_add(xx, zz))); pc.R[5] = vba_mul(2.0, vba_add(yz, -xw));
    pc.R[6] = vba_mul(2.0, vba_add(xz, -yw)); pc.R[7] = vba_mul(2.0, vba_add(yz, xw)); pc.R[8] = vba_add(1.0, -vba_mul(2.0, vba_add(xx, yy)));
    pc.fx = K[0]; pc.fy = K[1]; pc.cx = K[2]; pc.cy = K[3];
}

// Reprojection (BA_utils.py:30-43, 1052-1069, 7-17): p_c = R^T (X - t), u = fx x/z + cx, z clamped at 0.1.
VBA_HD void project(const PoseCam& pc, double X, double Y, double Z, double& u, double& v, double* cam /*[3]*/,
                    double& d) {
    const double dx = X - pc.t[0], dy = Y - pc.t[1], dz = Z - pc.t[2];
    cam[0] = vba_add(vba_add(vba_mul(pc.R[0], dx), vba_mul(pc.R[3], dy)), vba_mul(pc.R[6], dz));     // (see pose_camera)
    cam[1] = vba_add(vba_add(vba_mul(pc.R[1], dx), vba_mul(pc.R[4], dy)), vba_mul(pc.R[7], dz));
    cam[2] = vba_add(vba_add(vba_mul(pc.R[2], dx), vba_mul(pc.R[5], dy)), vba_mul(pc.R[8], dz));
    const double zc = cam[2] > kZMin ? cam[2] : kZMin;
#if defined(__HIP_DEVICE_COMPILE__)
    // 1 / zc from v_rcp_f64 refined to ~1 ulp (r (1 + e + e^2), e = 1 - zc r): the IEEE division sequence is ~30
    // instructions in every per-observation kernel, this is 4
    double rz = __builtin_amdgcn_rcp(zc);
    const double ez = fma(-zc, rz, 1.0);
    d = fma(rz, fma(ez, ez, ez), rz);
#else
    d = 1.0 / zc;
#endif
    u = vba_add(vba_mul(pc.fx, vba_mul(d, cam[0])), pc.cx);
    v = vba_add(vba_mul(pc.fy, vba_mul(d, cam[1])), pc.cy);
}

// 2x6 Jacobian of (u,v) w.r.t. [dp, dtheta] (BA_utils.py:44-48): [-Jpi R^T | 2 Jpi hat(p_c)], with the
// z-derivative switched off below the clamp.  J row major [2][6].
VBA_HD void project_jacobian(const PoseCam& pc, const double* cam, double d, double* J) {
    const double live = cam[2] > kZMin ? 1.0 : 0.0;
    const double a00 = pc.fx * d, a02 = -pc.fx * cam[0] * d * d * live;
    const double a11 = pc.fy * d, a12 = -pc.fy * cam[1] * d * d * live;
    for (int c = 0; c < 3; ++c) {   // -Jpi R^T : column c uses row c of R
        J[c] = -(a00 * pc.R[3 * c + 0] + a02 * pc.R[3 * c + 2]);
        J[6 + c] = -(a11 * pc.R[3 * c + 1] + a12 * pc.R[3 * c + 2]);
    }
    const double x = cam[0], y = cam[1], z = cam[2];
    J[3] = 2.0 * (-a02 * y);
    J[4] = 2.0 * (-a00 * z + a02 * x);
    J[5] = 2.0 * (a00 * y);
    J[9] = 2.0 * (a11 * z - a12 * y);
    J[10] = 2.0 * (a12 * x);
    J[11] = 2.0 * (-a11 * x);
}

// ------------------------------------------------------------------------------------------------ weights
struct RobustParams {
    double c;           // lower median of |r| (BA_filtering.py:23)
    double inv_c;       // 1/c
    double inv_c2;      // 1/c^2
    double inv_am2;     // 1/|alpha-2| (inf at alpha == 2, unused there)
    double am2;         // |alpha-2|
    double expo;        // alpha/2 - 1
    int alpha_is_2;     // alpha == 2: ((r/c)^2/0 + 1)^0 == 1 by IEEE inf**0 / nan**0 (BA_filtering.py:24)
    int expo_is_mhalf;  // alpha == 1 (every call from iter 3 on): x^(-1/2) = 1/sqrt(x), no pow() needed
};

// x^(-1/2) for x >= 1: on the device v_rsq_f64 refined by two Newton steps (~1 ulp); the IEEE sqrt + divide
// sequence is ~8x as many instructions and this sits in the per-observation loop of an issue-bound kernel
VBA_HD double inv_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
#else
    return 1.0 / sqrt(x);
#endif
}

// mean over the two pixel components of the Barron-style weight (BA_filtering.py:24), before /max and *conf
VBA_HD double robust_weight_raw(const RobustParams& rp, double ru, double rv) {
    if (rp.alpha_is_2) return rp.inv_c2;
    const double su = ru * rp.inv_c, sv = rv * rp.inv_c;
    const double xu = su * su * rp.inv_am2 + 1.0, xv = sv * sv * rp.inv_am2 + 1.0;
    double wu, wv;
    if (rp.expo_is_mhalf) {
        wu = inv_sqrt(xu) * rp.inv_c2;
        wv = inv_sqrt(xv) * rp.inv_c2;
    } else {
        wu = pow(xu, rp.expo) * rp.inv_c2;
        wv = pow(xv, rp.expo) * rp.inv_c2;
    }
    return (wu + wv) * 0.5;
}

// ------------------------------------------------------------------------------------------------ orbit
// J2 two-body acceleration (BA_utils.py:883-899) and its linearisation at p.
// On the device 1/r comes from v_rsq_f64 refined by two Newton steps (~1 ulp) and the powers of 1/r by
// multiplication: the IEEE sqrt + three divisions of the plain formula are ~100 instructions on the critical path
// of every stage of every RK4 step (a 1000 s gap is 4000 dependent evaluations).
// The derivative along a tangent tp is  da_i = A_i tp_i + p_i sum_j W_ij p_j tp_j  with A_i = k7 u_i - k3 (a_i = A_i p_i),
// W_ij = (3 k3 - 7 k7 u_i) / r^2 + k7 c_ij, c = d u_i / d(p_j^2) (rows 0 and 1 share u_0): the eight numbers A0, A2, W0[3], W2[3]
// depend on p alone, so every tangent carried through the same stage shares them -- 15 operations per tangent and stage where
// the directional-derivative formula written out per tangent took 36 (round 5; the acceleration itself is computed as before,
// bit for bit).
struct AccelLin { double A0, A2, W0[3], W2[3]; };
VBA_HD void accel_lin(const double* p, double* a, AccelLin* L) {
    const double px2 = p[0] * p[0], py2 = p[1] * p[1], pz2 = p[2] * p[2];
    const double r2 = px2 + py2 + pz2;
#if defined(__HIP_DEVICE_COMPILE__)
    double ir = __builtin_amdgcn_rsq(r2);
    ir = ir * fma(-0.5 * r2 * ir, ir, 1.5);
    ir = ir * fma(-0.5 * r2 * ir, ir, 1.5);
    const double ir2 = ir * ir;
    const double ir3 = ir2 * ir;
    const double k3 = kMu * ir3;
    const double k7 = kJ2c * (ir3 * ir3 * ir);
#else
    const double r = sqrt(r2);
    const double r3 = r * r * r;
    const double r7 = r3 * r3 * r;
    const double k3 = kMu / r3;
    const double k7 = kJ2c / r7;
    const double ir2 = 1.0 / r2;
#endif
    const double u0 = 6.0 * px2 - 1.5 * py2 - 1.5 * pz2;
    const double u2 = 3.0 * px2 - 4.5 * py2 - 4.5 * pz2;
    a[0] = -k3 * p[0] + k7 * u0 * p[0];
    a[1] = -k3 * p[1] + k7 * u0 * p[1];
    a[2] = -k3 * p[2] + k7 * u2 * p[2];
    if (!L) return;
    L->A0 = k7 * u0 - k3;
    L->A2 = k7 * u2 - k3;
    const double B0 = (3.0 * k3 - 7.0 * k7 * u0) * ir2, B2 = (3.0 * k3 - 7.0 * k7 * u2) * ir2;
    L->W0[0] = B0 + 12.0 * k7; L->W0[1] = B0 - 3.0 * k7; L->W0[2] = L->W0[1];
    L->W2[0] = B2 + 6.0 * k7;  L->W2[1] = B2 - 9.0 * k7; L->W2[2] = L->W2[1];
}
VBA_HD void accel_apply(const double* p, const AccelLin& L, const double* tp, double* da) {
    const double q0 = p[0] * tp[0], q1 = p[1] * tp[1], q2 = p[2] * tp[2];
    const double s0 = L.W0[0] * q0 + L.W0[1] * q1 + L.W0[2] * q2;
    const double s2 = L.W2[0] * q0 + L.W2[1] * q1 + L.W2[2] * q2;
    da[0] = L.A0 * tp[0] + p[0] * s0;
    da[1] = L.A0 * tp[1] + p[1] * s0;
    da[2] = L.A2 * tp[2] + p[2] * s2;
}
// (the acceleration alone, or with its derivative along one tangent)
VBA_HD void accel_jvp(const double* p, const double* tp, double* a, double* da, bool with_tangent) {
    if (!with_tangent) { accel_lin(p, a, nullptr); return; }
    AccelLin L;
    accel_lin(p, a, &L);
    accel_apply(p, L, tp, da);
}

// One RK4 step of length h of x = [p, v] and, optionally, of one tangent vector t (forward mode); this is
// RK4 at BA_utils.py:901-912 and, chained over the gap, propagate_orbit_dynamics :73-87 (h = 1 s).
template <bool TANGENT>
VBA_HD void rk4_step(double* x /*[6]*/, double* t /*[6]*/, double h = 1.0) {
    double k1[6], k2[6], k3[6], k4[6], d1[6], d2[6], d3[6], d4[6], xs[6], ts[6];
    for (int i = 0; i < 3; ++i) { k1[i] = x[3 + i]; if (TANGENT) d1[i] = t[3 + i]; }
    accel_jvp(x, t, k1 + 3, d1 + 3, TANGENT);
    for (int i = 0; i < 6; ++i) { xs[i] = x[i] + 0.5 * h * k1[i]; if (TANGENT) ts[i] = t[i] + 0.5 * h * d1[i]; }
    for (int i = 0; i < 3; ++i) { k2[i] = xs[3 + i]; if (TANGENT) d2[i] = ts[3 + i]; }
    accel_jvp(xs, ts, k2 + 3, d2 + 3, TANGENT);
    for (int i = 0; i < 6; ++i) { xs[i] = x[i] + 0.5 * h * k2[i]; if (TANGENT) ts[i] = t[i] + 0.5 * h * d2[i]; }
    for (int i = 0; i < 3; ++i) { k3[i] = xs[3 + i]; if (TANGENT) d3[i] = ts[3 + i]; }
    accel_jvp(xs, ts, k3 + 3, d3 + 3, TANGENT);
    for (int i = 0; i < 6; ++i) { xs[i] = x[i] + h * k3[i]; if (TANGENT) ts[i] = t[i] + h * d3[i]; }
    for (int i = 0; i < 3; ++i) { k4[i] = xs[3 + i]; if (TANGENT) d4[i] = ts[3 + i]; }
    accel_jvp(xs, ts, k4 + 3, d4 + 3, TANGENT);
    for (int i = 0; i < 6; ++i) {
        x[i] = x[i] + (h / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        if (TANGENT) t[i] = t[i] + (h / 6.0) * (d1[i] + 2 * d2[i] + 2 * d3[i] + d4[i]);
    }
}

// Propagation over a gap of `steps` seconds.  hop == 0: `steps` one-second RK4 steps (the reference's CPU branch
// `predict`, BA_utils.py:73-87, the parity default).  hop != 0: the schedule of `propagate_orbit_dynamics_skip`
// (BA_utils.py:52-71) that the reference's `predict_gpu` uses: floor(steps/100) steps of 100 s, then one step of
// steps % 100 s (a zero-length last step leaves the state untouched and is skipped).
template <bool TANGENT>
VBA_HD void propagate_gap(double* x, double* t, int steps, int hop) {
    if (!hop) {
        for (int q = 0; q < steps; ++q) rk4_step<TANGENT>(x, t, 1.0);
        return;
    }
    const int nfull = steps / 100, rem = steps % 100;
    for (int q = 0; q < nfull; ++q) rk4_step<TANGENT>(x, t, 100.0);
    if (rem) rk4_step<TANGENT>(x, t, (double)rem);
}

// The same step carrying NT tangents in one thread (batched windows: two lanes per pose with three tangents each instead
// of six lanes with one -- the base trajectory is integrated twice per pose instead of six times).  Same operations per
// element in the same order as rk4_step: the stage derivatives are added into the weighted sum as they appear instead
// of being kept (k1 + 2 k2 + 2 k3 + k4 is evaluated left to right either way), which keeps 3 x 18 instead of 3 x 42
// doubles of tangent state live.
template <int NT>
VBA_HD void rk4_step_multi(double* x /*[6]*/, double (*t)[6], double h) {
    double xs[6], ax[6], ts[NT][6], at[NT][6], k[6], d[NT][6];
    // stage 1 at (x, t)
    for (int i = 0; i < 3; ++i) k[i] = x[3 + i];
    AccelLin L;
    accel_lin(x, k + 3, &L);
    for (int j = 0; j < NT; ++j) {
        for (int i = 0; i < 3; ++i) d[j][i] = t[j][3 + i];
        accel_apply(x, L, t[j], d[j] + 3);
    }
    for (int i = 0; i < 6; ++i) { ax[i] = k[i]; xs[i] = x[i] + 0.5 * h * k[i]; }
    for (int j = 0; j < NT; ++j)
        for (int i = 0; i < 6; ++i) { at[j][i] = d[j][i]; ts[j][i] = t[j][i] + 0.5 * h * d[j][i]; }
    // stage 2
    for (int i = 0; i < 3; ++i) k[i] = xs[3 + i];
    accel_lin(xs, k + 3, &L);
    for (int j = 0; j < NT; ++j) {
        for (int i = 0; i < 3; ++i) d[j][i] = ts[j][3 + i];
        accel_apply(xs, L, ts[j], d[j] + 3);
    }
    for (int i = 0; i < 6; ++i) { ax[i] = ax[i] + 2 * k[i]; xs[i] = x[i] + 0.5 * h * k[i]; }
    for (int j = 0; j < NT; ++j)
        for (int i = 0; i < 6; ++i) { at[j][i] = at[j][i] + 2 * d[j][i]; ts[j][i] = t[j][i] + 0.5 * h * d[j][i]; }
    // stage 3
    for (int i = 0; i < 3; ++i) k[i] = xs[3 + i];
    accel_lin(xs, k + 3, &L);
    for (int j = 0; j < NT; ++j) {
        for (int i = 0; i < 3; ++i) d[j][i] = ts[j][3 + i];
        accel_apply(xs, L, ts[j], d[j] + 3);
    }
    for (int i = 0; i < 6; ++i) { ax[i] = ax[i] + 2 * k[i]; xs[i] = x[i] + h * k[i]; }
    for (int j = 0; j < NT; ++j)
        for (int i = 0; i < 6; ++i) { at[j][i] = at[j][i] + 2 * d[j][i]; ts[j][i] = t[j][i] + h * d[j][i]; }
    // stage 4
    for (int i = 0; i < 3; ++i) k[i] = xs[3 + i];
    accel_lin(xs, k + 3, &L);
    for (int j = 0; j < NT; ++j) {
        for (int i = 0; i < 3; ++i) d[j][i] = ts[j][3 + i];
        accel_apply(xs, L, ts[j], d[j] + 3);
    }
    for (int i = 0; i < 6; ++i) x[i] = x[i] + (h / 6.0) * (ax[i] + k[i]);
    for (int j = 0; j < NT; ++j)
        for (int i = 0; i < 6; ++i) t[j][i] = t[j][i] + (h / 6.0) * (at[j][i] + d[j][i]);
}

template <int NT>
VBA_HD void propagate_gap_multi(double* x, double (*t)[6], int steps, int hop) {
    if (!hop) {
        for (int q = 0; q < steps; ++q) rk4_step_multi<NT>(x, t, 1.0);
        return;
    }
    const int nfull = steps / 100, rem = steps % 100;
    for (int q = 0; q < nfull; ++q) rk4_step_multi<NT>(x, t, 100.0);
    if (rem) rk4_step_multi<NT>(x, t, (double)rem);
}

// How a LONG gap of s one-second steps is cut for the parallel-in-time propagation (vba_long.hip) -- a function of s alone, so
// that the host (which sizes the carried chunk states) and every kernel cut an edge alike.  P chunks of L steps (the last one
// shorter), a lane of one wavefront each: the serial part costs ~272 instructions per chunk (coarse chain + linearised sweep), the
// two fine passes 360 per step, L ~ sqrt(0.76 s) balances them; at most 64 chunks.  Each chunk is cut again into sub-chunks of
// `sub` steps whose start states the last fine pass leaves behind: the transition matrix of the edge is the ordered product of
// the G <= 128 sub-chunk matrices.
struct LongPlan { int L, P, sub, nsubL, G; };
VBA_HD LongPlan long_plan(int s) {
    LongPlan p;
    int L = 1;
    while (45 * L * L < 34 * s) ++L;
    if (64 * L < s) L = (s + 63) / 64;
    p.L = L;
    p.P = (s + L - 1) / L;             // <= 64
    const int per_chunk = 128 / p.P < 4 ? 128 / p.P : 4;      // sub-chunks a chunk may have (2 .. 4)
    const int even = (L + per_chunk - 1) / per_chunk;
    p.sub = even > 8 ? even : 8;
    p.nsubL = (L + p.sub - 1) / p.sub; // <= per_chunk
    const int last = s - (p.P - 1) * L;
    p.G = (p.P - 1) * p.nsubL + (last + p.sub - 1) / p.sub;
    return p;
}

// ------------------------------------------------------------------------------------------------ attitude
VBA_HD void quat_mul(const double* a, const double* b, double* o) {   // BA_utils.py:992-1000
    const double x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    const double x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    o[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    o[1] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    o[2] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
    o[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
}

// G(q) [4][3] (BA_utils.py:19-28)
VBA_HD void attitude_jac(const double* q, double* G) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    G[0] = w;  G[1] = -z; G[2] = y;
    G[3] = z;  G[4] = w;  G[5] = -x;
    G[6] = -y; G[7] = x;  G[8] = w;
    G[9] = -x; G[10] = -y; G[11] = -z;
}

// o = R_m(c) q = q (x) c ;  oT = R_m(c)^T q
VBA_HD void rm_apply(const double* c, const double* q, double* o) { quat_mul(q, c, o); }
VBA_HD void rmT_apply(const double* c, const double* q, double* o) {
    // R_m(c)^T = R_m(conj c) for unit-agnostic bilinear form: q (x) conj(c)
    const double cc[4] = {-c[0], -c[1], -c[2], c[3]};
    quat_mul(q, cc, o);
}

// Attitude dynamics term of pose i (BA_utils.py:481-487, 494-500, 519-523; closed form SURVEY appendix A.3).
// q_prev/c_prev may be null at i == 0, q_next null at i == n-1.  Outputs:
//   f      = 100 (1 - |<q_i (x) c_i, q_{i+1}>|)                       (valid when q_next)
//   qgrad  [3]   = G(q_i)^T grad_{q_i} sum f
//   Hd [9] = block (i,i); Hu [9] = block (i,i+1) (valid when q_next); Hl [9] = block (i,i-1) (valid when q_prev)
VBA_HD void attitude_term(const double* q_prev, const double* c_prev, const double* q, const double* c,
                          const double* q_next, double& f, double* qgrad, double* Hd, double* Hu, double* Hl) {
    double G[12];
    attitude_jac(q, G);
    double g[4] = {0, 0, 0, 0};
    f = 0.0;
    for (int k = 0; k < 9; ++k) { Hu[k] = 0.0; Hl[k] = 0.0; }
    if (q_next) {
        double qp[4];
        quat_mul(q, c, qp);
        const double d = qp[0] * q_next[0] + qp[1] * q_next[1] + qp[2] * q_next[2] + qp[3] * q_next[3];
        const double s = d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0);
        f = kQuatCoeff * (1.0 - fabs(d));
        double t4[4];
        rmT_apply(c, q_next, t4);               // R_m(c_i)^T q_{i+1}
        for (int k = 0; k < 4; ++k) g[k] += -kQuatCoeff * s * t4[k];
        // Hu = G(q_i)^T (-kappa s R_m(c_i)^T) G(q_{i+1})
        double Gn[12];
        attitude_jac(q_next, Gn);
        for (int b = 0; b < 3; ++b) {
            const double col[4] = {Gn[b], Gn[3 + b], Gn[6 + b], Gn[9 + b]};
            double rc[4];
            rmT_apply(c, col, rc);
            for (int a = 0; a < 3; ++a)
                Hu[3 * a + b] = -kQuatCoeff * s * (G[a] * rc[0] + G[3 + a] * rc[1] + G[6 + a] * rc[2] + G[9 + a] * rc[3]);
        }
    }
    if (q_prev) {
        double qp[4];
        quat_mul(q_prev, c_prev, qp);           // R_m(c_{i-1}) q_{i-1}
        const double d = qp[0] * q[0] + qp[1] * q[1] + qp[2] * q[2] + qp[3] * q[3];
        const double s = d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0);
        for (int k = 0; k < 4; ++k) g[k] += -kQuatCoeff * s * qp[k];
        double Gp[12];
        attitude_jac(q_prev, Gp);
        for (int b = 0; b < 3; ++b) {
            const double col[4] = {Gp[b], Gp[3 + b], Gp[6 + b], Gp[9 + b]};
            double rc[4];
            rm_apply(c_prev, col, rc);
            for (int a = 0; a < 3; ++a)
                Hl[3 * a + b] = -kQuatCoeff * s * (G[a] * rc[0] + G[3 + a] * rc[1] + G[6 + a] * rc[2] + G[9 + a] * rc[3]);
        }
    }
    for (int a = 0; a < 3; ++a) qgrad[a] = G[a] * g[0] + G[3 + a] * g[1] + G[6 + a] * g[2] + G[9 + a] * g[3];
    // Hd = B(g) G, B[a][c] = sum_k dG[k][a]/dq[c] g[k]
    const double B[12] = {-g[3], -g[2], g[1], g[0],
                          g[2], -g[3], -g[0], g[1],
                          -g[1], g[0], -g[3], g[2]};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            Hd[3 * a + b] = B[4 * a + 0] * G[b] + B[4 * a + 1] * G[3 + b] + B[4 * a + 2] * G[6 + b] + B[4 * a + 3] * G[9 + b];
}

// residual-only variant
VBA_HD double attitude_residual(const double* q, const double* c, const double* q_next) {
    double qp[4];
    quat_mul(q, c, qp);
    const double d = qp[0] * q_next[0] + qp[1] * q_next[1] + qp[2] * q_next[2] + qp[3] * q_next[3];
    return kQuatCoeff * (1.0 - fabs(d));
}

// ------------------------------------------------------------------------------------------------ retraction
// BA_filtering.py:56-60 with quaternion_exp (BA_utils.py:970-985).
VBA_HD void retract(const double* s, const double* dp /*[9]*/, double* o /*[10]*/) {
    o[0] = s[0] + dp[0]; o[1] = s[1] + dp[1]; o[2] = s[2] + dp[2];
    const double th = sqrt(dp[3] * dp[3] + dp[4] * dp[4] + dp[5] * dp[5]);
    double e[4];
    if (th < 1e-16) {
        e[0] = e[1] = e[2] = 0.0; e[3] = 1.0;
    } else {
        double sn, cs;
        sincos(th / 2, &sn, &cs);       // one argument reduction for both
        const double sc = sn / (th + 1e-16);
        e[0] = dp[3] * sc; e[1] = dp[4] * sc; e[2] = dp[5] * sc; e[3] = cs;
    }
    double r[4];
    quat_mul(s + 3, e, r);
    const double inv = 1.0 / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    o[3] = r[0] * inv; o[4] = r[1] * inv; o[5] = r[2] * inv; o[6] = r[3] * inv;
    o[7] = s[7] + dp[6]; o[8] = s[8] + dp[7]; o[9] = s[9] + dp[8];
}

// ------------------------------------------------------------------------------------------------ assembly
// Inputs of one pose row of the block-tridiagonal system (BA_filtering.py:40-48, 54).
struct AsmRow {
    const double* Hraw;       // [21] packed sum (w_raw conf) J^T J of pose i
    const double* braw;       // [6]
    double inv_wmax;          // 1 / max raw weight
    double sigma;             // 0 in the landmark-only phase
    const double* Phi_i;      // [36] row major, pose i      (null if i == n-1 or sigma == 0)
    const double* Phi_im1;    // [36] pose i-1               (null if i == 0 or sigma == 0)
    const double* rorb_i;     // [6]  D (x_hat_i - x_{i+1})  (null if i == n-1)
    const double* rorb_im1;   // [6]                         (null if i == 0)
    const double* qgrad;      // [3]
    const double* Hd;         // [9]
    const double* Hu;         // [9]
    const double* Hl;         // [9]
    // BA_reg: propagated-covariance prior on (position, velocity) of this pose (null without prior)
    const double* prior_H;    // [36] hessian_state_t, row major
    const double* prior_r;    // [6]  H [p_prior - p ; v_prior - v]
};

// Prior of BA_reg (BA_filtering.py:146, 157-159, 166; prior_gpu BA_utils.py:617-627): r = H d, Jp = -H on the
// position / velocity columns (vel_coeff = 1 as called), so J^T J gains H^T H and the right-hand side H^T r there.
// The rotation part of prior_gpu is analytically constant (G(q)^T q = 0): no gradient, no Hessian.
// (Every sum of products of the assembly is written with explicit fma / vba_mul / vba_add: three code paths form these
// entries -- per entry here, in uniform passes in vba_asm_fast.h, column-wise in the four-windows-per-wave walk -- and under
// -ffp-contract=fast the optimiser would contract `v - sigma * x` in one of them and not in the other, depending on how it
// happened to if-convert the surrounding selects.  The systems must agree to the bit.)
VBA_HD double prior_hth(const double* H, int a, int b) {
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s = fma(H[k * 6 + a], H[k * 6 + b], s);
    return s;
}
VBA_HD double prior_htr(const double* H, const double* r, int a) {
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s = fma(H[k * 6 + a], r[k], s);
    return s;
}
// r = H d for one pose: st = its state [10], xp = prior position / velocity [6]
VBA_HD void prior_residual(const double* H, const double* xp, const double* st, double* r /*[6]*/) {
    const double d[6] = {xp[0] - st[0], xp[1] - st[1], xp[2] - st[2], xp[3] - st[7], xp[4] - st[8], xp[5] - st[9]};
    for (int a = 0; a < 6; ++a) {
        double s = 0.0;
        for (int b = 0; b < 6; ++b) s += H[a * 6 + b] * d[b];
        r[a] = s;
    }
}

// column c (0..8) of E_i = D Phi_i laid out over [dp, dtheta, dv]: rows r = 0..5; rotation columns are 0.
VBA_HD double E_entry(const double* Phi, int r, int c) {
    if (c >= 3 && c < 6) return 0.0;
    const int pc = c < 3 ? c : c - 3;
    const double D = r < 3 ? 1.0 : kVelCoeff;
    return vba_mul(D, Phi[6 * r + pc]);
}
// F = -D selects position / velocity: column c has a single non-zero at row frow(c)
VBA_HD int F_row(int c) { return c < 3 ? c : (c >= 6 ? c - 3 : -1); }
VBA_HD double F_val(int c) { return c < 3 ? -1.0 : -kVelCoeff; }

// entry (a,b) of band `which` (0 sub (i,i-1), 1 diag, 2 super (i,i+1)) of pose row i, no damping.
VBA_HD double band_entry(const AsmRow& R, int which, int a, int b) {
    double v = 0.0;
    const bool rot = (a >= 3 && a < 6 && b >= 3 && b < 6);
    if (which == 1) {
        if (a < 6 && b < 6) v = vba_mul(R.Hraw[sym6(a, b)], R.inv_wmax);
        if (R.sigma != 0.0) {
            if (R.Phi_i) {
                double s = 0.0;
                for (int r = 0; r < 6; ++r) s = fma(vba_mul(E_entry(R.Phi_i, r, a), R.sigma), E_entry(R.Phi_i, r, b), s);
                v = vba_add(v, s);
            }
            if (R.Phi_im1 && a == b && F_row(a) >= 0) v = fma(vba_mul(F_val(a), R.sigma), F_val(a), v);
            if (rot) v = fma(R.sigma, R.Hd[3 * (a - 3) + (b - 3)], v);
        }
        if (R.prior_H && F_row(a) >= 0 && F_row(b) >= 0) v = vba_add(v, prior_hth(R.prior_H, F_row(a), F_row(b)));
    } else if (R.sigma != 0.0) {
        if (which == 2 && R.Phi_i) {
            const int r = F_row(b);
            if (r >= 0) v = vba_mul(vba_mul(E_entry(R.Phi_i, r, a), R.sigma), F_val(b));
            if (rot) v = vba_mul(R.sigma, R.Hu[3 * (a - 3) + (b - 3)]);         // (exclusive: a rotation column has no F row)
        }
        if (which == 0 && R.Phi_im1) {
            const int r = F_row(a);
            if (r >= 0) v = vba_mul(vba_mul(F_val(a), R.sigma), E_entry(R.Phi_im1, r, b));
            if (rot) v = vba_mul(R.sigma, R.Hl[3 * (a - 3) + (b - 3)]);
        }
    }
    return v;
}

VBA_HD double rhs_entry(const AsmRow& R, int a) {
    double v = 0.0;
    if (a < 6) v = vba_mul(R.braw[a], R.inv_wmax);
    if (R.sigma != 0.0) {
        if (R.Phi_i) {
            double s = 0.0;
            for (int r = 0; r < 6; ++r) s = fma(vba_mul(E_entry(R.Phi_i, r, a), R.sigma), R.rorb_i[r], s);
            v = vba_add(v, -s);
        }
        if (R.Phi_im1) {
            const int r = F_row(a);
            if (r >= 0) v = fma(vba_mul(F_val(a), R.sigma), -R.rorb_im1[r], v);
        }
        if (a >= 3 && a < 6) v = fma(R.sigma, -R.qgrad[a - 3], v);
    }
    if (R.prior_H && F_row(a) >= 0) v = vba_add(v, prior_htr(R.prior_H, R.prior_r, F_row(a)));
    return v;
}

}  // namespace vba
