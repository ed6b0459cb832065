"""CPU oracle for the VINSat bundle-adjustment iteration -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy fp64 restatement of the reference algorithm
(``estimation/BA/BA_filtering.py:4-98`` and the parts of ``estimation/BA/BA_utils.py``
it calls).  It exists so that the HIP path can be checked on machines where the
reference itself is not present (the GPU box).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it;
the product package ``vinsat_amd`` never does.

Parity pinning: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned against outputs of the reference
itself, captured in the build container by ``tools/gen_golden.py`` and committed under
``tests/golden/`` (see ``tests/test_oracle_golden.py``).

The reference obtains its Jacobians by reverse-mode autograd through dense (9n x 9n)
objects; here they are the closed forms (each checked against the captured autograd
output), and the normal equations are kept in their true block-tridiagonal shape.

State layout per pose: ``[px py pz | qx qy qz qw | vx vy vz]`` (km, scalar-last unit
quaternion, km/s); tangent layout per pose: ``[dp(3) | dtheta(3) | dv(3)]``.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

MU = 398600.4418            # BA_utils.py:883
J2C = 1.75553e10            # BA_utils.py:883
J2_MAT = np.array([[6.0, -1.5, -1.5], [6.0, -1.5, -1.5], [3.0, -4.5, -4.5]])  # BA_utils.py:888-892
QUAT_COEFF = 100.0          # BA_filtering.py:11
VEL_COEFF = 100.0           # BA_filtering.py:12
Z_MIN = 0.1                 # BA_utils.py:13


# ----------------------------------------------------------------------------- quaternions
def qmul(q1, q2):
    """Hamilton product, scalar last (BA_utils.py:992-1000)."""
    x1, y1, z1, w1 = np.moveaxis(q1, -1, 0)
    x2, y2, z2, w2 = np.moveaxis(q2, -1, 0)
    return np.stack([
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
    ], axis=-1)


def qexp(d):
    """BA_utils.py:970-985 (half-angle exponential, identity for |d| < 1e-16)."""
    th = np.linalg.norm(d, axis=-1)[..., None]
    mask = (th < 1e-16).astype(np.float64)
    ident = np.concatenate([np.zeros_like(d), np.ones_like(th)], -1)
    q = np.concatenate([d * np.sin(th / 2) / (th + 1e-16), np.cos(th / 2)], -1)
    return ident * mask + q * (1 - mask)


def attitude_jacobian(q):
    """G(q) [...,4,3] = d(q (x) [delta;1])/d delta (BA_utils.py:19-28)."""
    x, y, z, w = np.moveaxis(q, -1, 0)
    return np.stack([
        np.stack([w, -z, y], -1),
        np.stack([z, w, -x], -1),
        np.stack([-y, x, w], -1),
        np.stack([-x, -y, -z], -1),
    ], -2)


def right_mult_matrix(c):
    """R_m(c) with q (x) c = R_m(c) q (cf. BA_utils.py:1002-1010)."""
    x, y, z, w = np.moveaxis(c, -1, 0)
    return np.stack([
        np.stack([w, z, -y, x], -1),
        np.stack([-z, w, x, y], -1),
        np.stack([y, -x, w, z], -1),
        np.stack([-x, -y, -z, w], -1),
    ], -2)


def rotation_matrix(q):
    x, y, z, w = np.moveaxis(q, -1, 0)
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1),
    ], -2)


# ----------------------------------------------------------------------------- A1 / A2
def landmark_project(states, xyz, intr, ii, jacobian=True):
    """Reprojection of fixed landmarks and its 2x6 Jacobian per observation.

    Reference: ``landmark_project`` BA_utils.py:30-50, ``apply_inverse_pose_transformation``
    :1052-1069, ``proj`` :7-17, ``attitude_jacobian`` :19-28.
    states [n,10], xyz [m,3], intr [n,4], ii [m] -> est [m,2] (, Jg [m,2,6] over [dp, dtheta]).
    """
    p = states[ii, :3]
    q = states[ii, 3:7]
    qn = q / np.linalg.norm(q, axis=-1, keepdims=True)
    R = rotation_matrix(qn)                     # camera -> ECI
    pc = np.einsum("kji,kj->ki", R, xyz - p)    # R^T (X - t)
    K = intr[ii]
    zc = np.maximum(pc[:, 2], Z_MIN)
    d = 1.0 / zc
    est = np.stack([K[:, 0] * (d * pc[:, 0]) + K[:, 2], K[:, 1] * (d * pc[:, 1]) + K[:, 3]], -1)
    if not jacobian:
        return est
    live = (pc[:, 2] > Z_MIN).astype(np.float64)   # clamp => zero z-derivative below Z_MIN
    m = xyz.shape[0]
    Jpi = np.zeros((m, 2, 3))
    Jpi[:, 0, 0] = K[:, 0] * d
    Jpi[:, 1, 1] = K[:, 1] * d
    Jpi[:, 0, 2] = -K[:, 0] * pc[:, 0] * d * d * live
    Jpi[:, 1, 2] = -K[:, 1] * pc[:, 1] * d * d * live
    hat = np.zeros((m, 3, 3))
    hat[:, 0, 1], hat[:, 0, 2] = -pc[:, 2], pc[:, 1]
    hat[:, 1, 0], hat[:, 1, 2] = pc[:, 2], -pc[:, 0]
    hat[:, 2, 0], hat[:, 2, 1] = -pc[:, 1], pc[:, 0]
    Jt = -np.einsum("kab,kcb->kac", Jpi, R)     # -Jpi R^T
    Jr = 2.0 * np.einsum("kab,kbc->kac", Jpi, hat)
    return est, np.concatenate([Jt, Jr], -1)


# ----------------------------------------------------------------------------- A3a
def lm_schedule(it):
    """alpha and Sigma for a given outer iteration (BA_filtering.py:22, 26)."""
    alpha = min(max(1 - (2 * (it / 5) - 1), 1), 2)
    sigma = min(10000 * (it + 1) ** 2, 1000000)
    return alpha, sigma


def robust_weights(r_obs, it, conf):
    """BA_filtering.py:21-25.  r_obs [m,2], conf [m] -> (w [m], c_obs, wmax)."""
    alpha, _ = lm_schedule(it)
    a = np.abs(r_obs).reshape(-1)
    c = np.sort(a)[(a.size - 1) // 2]           # torch.median == lower median
    with np.errstate(divide="ignore", invalid="ignore"):
        per = (((r_obs / c) ** 2) / abs(alpha - 2) + 1) ** (alpha / 2 - 1) / (c ** 2)
    w_raw = per.mean(-1)
    wmax = w_raw.max()
    return (w_raw / wmax) * conf, c, wmax


# ----------------------------------------------------------------------------- A3b
def accumulate(Jg, w, r_obs, ii, n):
    """Per-pose normal equations H_i [n,6,6], b_i [n,6] (BA_filtering.py:30-37, 44)."""
    JtJ = np.einsum("kca,kcb->kab", Jg * w[:, None, None], Jg)
    Jtr = np.einsum("kca,kc->ka", Jg * w[:, None, None], r_obs)
    H = np.zeros((n, 6, 6))
    b = np.zeros((n, 6))
    np.add.at(H, ii, JtJ)
    np.add.at(b, ii, Jtr)
    return H, b


# ----------------------------------------------------------------------------- A4
def _accel(p):
    r2 = (p * p).sum(-1, keepdims=True)
    r = np.sqrt(r2)
    u = (p * p) @ J2_MAT.T
    return -(MU / r ** 3) * p + (J2C / r ** 7) * u * p


def _accel_jac(p):
    """d a / d p, [...,3,3]."""
    r2 = (p * p).sum(-1)
    r = np.sqrt(r2)
    u = (p * p) @ J2_MAT.T
    I = np.eye(3)
    pp = p[..., :, None] * p[..., None, :]
    G = -MU * (I / (r ** 3)[..., None, None] - 3 * pp / (r ** 5)[..., None, None])
    up = u * p
    G = G + J2C * (-7 * up[..., :, None] * p[..., None, :] / (r ** 9)[..., None, None]
                   + (2 * J2_MAT * pp + u[..., :, None] * I) / (r ** 7)[..., None, None])
    return G


def _deriv(x):
    return np.concatenate([x[..., 3:], _accel(x[..., :3])], -1)


def _deriv_jvp(x, T):
    """F(x) @ T with F = [[0, I], [G, 0]]; T [...,6,k]."""
    G = _accel_jac(x[..., :3])
    return np.concatenate([T[..., 3:, :], G @ T[..., :3, :]], -2)


def rk4_step(x, h=1.0):
    """One RK4 step of the J2 orbit model (BA_utils.py:901-912, 883-899)."""
    f1 = _deriv(x)
    f2 = _deriv(x + 0.5 * h * f1)
    f3 = _deriv(x + 0.5 * h * f2)
    f4 = _deriv(x + h * f3)
    return x + (h / 6.0) * (f1 + 2 * f2 + 2 * f3 + f4)


def rk4_step_stm(x, Phi, h=1.0, hm=None):
    """RK4 step of the state together with its 6x6 sensitivity (forward mode); hm = h shaped for the matrices."""
    if hm is None:
        hm = h
    f1 = _deriv(x)
    d1 = _deriv_jvp(x, Phi)
    x2 = x + 0.5 * h * f1
    f2 = _deriv(x2)
    d2 = _deriv_jvp(x2, Phi + 0.5 * hm * d1)
    x3 = x + 0.5 * h * f2
    f3 = _deriv(x3)
    d3 = _deriv_jvp(x3, Phi + 0.5 * hm * d2)
    x4 = x + h * f3
    f4 = _deriv(x4)
    d4 = _deriv_jvp(x4, Phi + hm * d3)
    return x + (h / 6.0) * (f1 + 2 * f2 + 2 * f3 + f4), Phi + (hm / 6.0) * (d1 + 2 * d2 + 2 * d3 + d4)


def propagate_orbit(x, steps, stm=True, hop=False):
    """Advance pose i over a gap of steps[i] seconds.

    hop=False: steps[i] one-second RK4 steps (``propagate_orbit_dynamics`` BA_utils.py:73-87, the CPU branch).
    hop=True : ``propagate_orbit_dynamics_skip`` (BA_utils.py:52-71, used by ``predict_gpu``): floor(d/100) steps of
    100 s followed by one step of d % 100 s.
    x [n,6] -> x_hat [n,6] (, Phi [n,6,6] = d x_hat / d x).
    """
    x = x.copy()
    n = x.shape[0]
    Phi = np.broadcast_to(np.eye(6), (n, 6, 6)).copy()
    if hop:
        hops = steps // 100
        for s in range(int(hops.max()) + 1 if n else 0):
            h = np.where(hops == s, steps % 100, np.where(hops > s, 100, 0)).astype(np.float64)
            act = h > 0
            if not act.any():
                continue
            hh = h[act][:, None]
            if stm:
                x[act], Phi[act] = rk4_step_stm(x[act], Phi[act], hh, hh[:, :, None])
            else:
                x[act] = rk4_step(x[act], hh)
        return (x, Phi) if stm else x
    for s in range(int(steps.max()) if n else 0):
        act = steps > s
        if stm:
            x[act], Phi[act] = rk4_step_stm(x[act], Phi[act])
        else:
            x[act] = rk4_step(x[act])
    return (x, Phi) if stm else x


def step_counts(time_idx):
    """Seconds between consecutive poses, with the reference's trailing 1 (BA_utils.py:74-75)."""
    d = np.diff(np.asarray(time_idx, dtype=np.int64))
    return np.concatenate([d, np.ones(1, dtype=np.int64)])


def orbit_factor(states, time_idx, jacobian=True, hop=False):
    """Position/velocity dynamics residual and its two 6x9 Jacobian blocks per edge.

    Reference: ``predict`` BA_utils.py:467-476, 488-490, 501-509.
    Returns r [n-1,6], (E [n-1,6,9] at pose i, F [6,9] at pose i+1).
    """
    x = np.concatenate([states[:, :3], states[:, 7:10]], -1)
    steps = step_counts(time_idx)
    D = np.array([1.0, 1.0, 1.0, VEL_COEFF, VEL_COEFF, VEL_COEFF])
    if jacobian:
        xh, Phi = propagate_orbit(x, steps, stm=True, hop=hop)
    else:
        xh = propagate_orbit(x, steps, stm=False, hop=hop)
    r = (xh[:-1] - x[1:]) * D
    if not jacobian:
        return r
    n = states.shape[0]
    E = np.zeros((n - 1, 6, 9))
    DPhi = D[None, :, None] * Phi[:-1]
    E[:, :, 0:3] = DPhi[:, :, 0:3]
    E[:, :, 6:9] = DPhi[:, :, 3:6]
    F = np.zeros((6, 9))
    F[0:3, 0:3] = -np.eye(3)
    F[3:6, 6:9] = -VEL_COEFF * np.eye(3)
    return r, E, F


# ----------------------------------------------------------------------------- A5
def attitude_factor(states, cumrot, jacobian=True):
    """Attitude dynamics residual, its tangent gradient and (non-symmetric) Newton blocks.

    Reference: ``predict`` BA_utils.py:481-487, 494-500, 519-523 and
    ``propagate_rotation_dynamics_precomp`` :290-304.
    cumrot [n,4] = rotation accumulated over the gap following pose i.
    Returns f [n-1]; qgrad [n,3]; Hd [n,3,3], Hu [n-1,3,3] (i,i+1), Hl [n-1,3,3] (i+1,i).
    """
    q = states[:, 3:7]
    n = q.shape[0]
    qp = qmul(q, cumrot)
    d = (qp[:-1] * q[1:]).sum(-1)
    f = QUAT_COEFF * (1 - np.abs(d))
    if not jacobian:
        return f
    s = np.sign(d)
    Rm = right_mult_matrix(cumrot)              # [n,4,4]
    G = attitude_jacobian(q)                    # [n,4,3]
    grad = np.zeros((n, 4))
    grad[:-1] += -QUAT_COEFF * s[:, None] * np.einsum("iba,ib->ia", Rm[:-1], q[1:])   # R_m(c_i)^T q_{i+1}
    grad[1:] += -QUAT_COEFF * s[:, None] * np.einsum("iab,ib->ia", Rm[:-1], q[:-1])   # R_m(c_{i-1}) q_{i-1}
    qgrad = np.einsum("ika,ik->ia", G, grad)
    g0, g1, g2, g3 = grad[:, 0], grad[:, 1], grad[:, 2], grad[:, 3]
    B = np.stack([
        np.stack([-g3, -g2, g1, g0], -1),
        np.stack([g2, -g3, -g0, g1], -1),
        np.stack([-g1, g0, -g3, g2], -1),
    ], -2)                                      # [n,3,4] = sum_k dG[k,a]/dq[c] grad[k]
    Hd = B @ G
    Hu = -QUAT_COEFF * s[:, None, None] * np.einsum("ika,ilk,ilb->iab", G[:-1], Rm[:-1], G[1:])
    Hl = -QUAT_COEFF * s[:, None, None] * np.einsum("ika,ikl,ilb->iab", G[1:], Rm[:-1], G[:-1])
    return f, qgrad, Hd, Hu, Hl


# ----------------------------------------------------------------------------- A6
def assemble(H, b, wmax_scale, sigma, E, F, r_orb, qgrad, Hd, Hu, Hl, initialize):
    """Block-tridiagonal normal equations WITHOUT damping (BA_filtering.py:40-48).

    Returns bands [n,3,9,9] (sub, diag, super) and rhs [n,9].
    """
    n = H.shape[0]
    bands = np.zeros((n, 3, 9, 9))
    rhs = np.zeros((n, 9))
    bands[:, 1, :6, :6] = H * wmax_scale
    rhs[:, :6] = b * wmax_scale
    if not initialize:
        EtE = np.einsum("irc,ird->icd", E * sigma, E)
        FtF = (F * sigma).T @ F
        EtF = np.einsum("irc,rd->icd", E * sigma, F)
        FtE = np.einsum("rc,ird->icd", F * sigma, E)
        bands[:-1, 1] += EtE
        bands[1:, 1] += FtF
        bands[:-1, 2] += EtF
        bands[1:, 0] += FtE
        rhs[:-1] -= np.einsum("irc,ir->ic", E * sigma, r_orb)
        rhs[1:] -= np.einsum("rc,ir->ic", F * sigma, r_orb)
        bands[:, 1, 3:6, 3:6] += sigma * Hd
        bands[:-1, 2, 3:6, 3:6] += sigma * Hu
        bands[1:, 0, 3:6, 3:6] += sigma * Hl
        rhs[:, 3:6] -= sigma * qgrad
    return bands, rhs


# ----------------------------------------------------------------------------- A7
def bands_to_dense(bands):
    n = bands.shape[0]
    A = np.zeros((n, 9, n, 9))
    for i in range(n):
        A[i, :, i, :] = bands[i, 1]
        if i > 0:
            A[i, :, i - 1, :] = bands[i, 0]
        if i < n - 1:
            A[i, :, i + 1, :] = bands[i, 2]
    return A.reshape(9 * n, 9 * n)


def solve_tridiag(bands, rhs, method="banded"):
    """dpose = A^{-1} rhs (BA_filtering.py:55: dense LU with partial pivoting).

    ``dense`` forms the full matrix like the reference; ``banded`` is LAPACK's banded
    LU with partial pivoting (same pivoting rule restricted to the band, measured to
    agree with the dense factorisation to <=5e-9 relative on the captured systems).
    """
    n = bands.shape[0]
    if method == "dense":
        return np.linalg.solve(bands_to_dense(bands), rhs.reshape(-1)).reshape(n, 9)
    N = 9 * n
    kl = ku = 17
    ab = np.zeros((kl + ku + 1, N))
    for i in range(n):
        for d, j in enumerate((i - 1, i, i + 1)):
            if 0 <= j < n:
                blk = bands[i, d]
                for a in range(9):
                    row = 9 * i + a
                    cols = 9 * j + np.arange(9)
                    ab[ku + row - cols, cols] = blk[a]
    return scipy.linalg.solve_banded((kl, ku), ab, rhs.reshape(-1)).reshape(n, 9)


# ----------------------------------------------------------------------------- A8
def retract(states, dpose):
    """BA_filtering.py:56-60."""
    pos = states[:, :3] + dpose[:, :3]
    vel = states[:, 7:] + dpose[:, 6:]
    rot = qmul(states[:, 3:7], qexp(dpose[:, 3:6]))
    rot = rot / np.linalg.norm(rot, axis=-1, keepdims=True)
    return np.concatenate([pos, rot, vel], -1)


def dynamics_residual(states, cumrot, time_idx, initialize, hop=False):
    """r_pred [n-1, 6 or 7] exactly as ``predict`` returns it (BA_utils.py:463-466, 476)."""
    n = states.shape[0]
    if initialize:
        return np.zeros((n - 1, 6))
    r = orbit_factor(states, time_idx, jacobian=False, hop=hop)
    f = attitude_factor(states, cumrot, jacobian=False)
    return np.concatenate([r, f[:, None]], -1)


def prior_factor(states, states_prior, hessian_state, vel_coeff=1.0):
    """State part of the propagated-covariance prior of ``BA_reg``: ``res_reg_state`` and its Jacobian
    (``prior_gpu`` BA_utils.py:617-627, 651-655; closed form of the autograd Jacobian).

    r_i = H_i [p_prior - p ; (v_prior - v) vel_coeff]  (6 per pose);  Jp_i [6,9] = d r_i / d(dp, dtheta, dv)
        = [-H_i[:, :3] | 0 | -vel_coeff H_i[:, 3:]].
    The rotation part ``res_reg_rot`` = quat_coeff (1 - |q_prior^T G(q_prior) H_rot G(q)^T q|) is analytically the
    CONSTANT quat_coeff: G(q)^T q = 0 for every q (the columns of the attitude Jacobian are orthogonal to q), so the
    argument of |.| is rounding noise (~1e-31), its gradient and Hessian (qgradp, Hqp) are rounding noise times
    H_rot (~1e-14) and are taken as exactly 0 here.  hessian_rot_t therefore does not enter.
    """
    d = np.concatenate([states_prior[:, :3] - states[:, :3], (states_prior[:, 7:] - states[:, 7:]) * vel_coeff], -1)
    r = np.einsum("iab,ib->ia", hessian_state, d)
    n = states.shape[0]
    Jp = np.zeros((n, 6, 9))
    Jp[:, :, 0:3] = -hessian_state[:, :, :3]
    Jp[:, :, 6:9] = -vel_coeff * hessian_state[:, :, 3:]
    return r, Jp


def ba_iteration(it, states, cumrot, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences,
                 lamda_init, initialize=False, solver="banded", debug=None, hop=False, prior=None):
    """One call of the reference's ``BA`` (BA_filtering.py:4-98) for batch size 1.

    hop=True swaps the orbit integrator for the coarse one of the reference's ``predict_gpu`` (BA_utils.py:52-71, 544).
    Arrays carry no batch dimension: states [n,10], cumrot [n,4] (= imu_meas[0,:,-1,6:10]),
    landmarks [m,2], landmarks_xyz [m,3], ii [m], time_idx [n], intrinsics [n,4],
    confidences [m].  Returns (states_new [n,10], lamda_out, last_hessian [9,9], n_trials).

    prior = (states_prior [n,10], hessian_state_t [n,6,6]) turns the call into the reference's ``BA_reg``
    (BA_filtering.py:100-210) AS WRITTEN: the prior enters the normal equations through Jp^T Jp and -Jp^T r_prior
    (:146, 157-159, 166); the residual means of the accept test carry 7 prior entries per pose whose last one is the
    constant quat_coeff of ``prior_gpu`` -- 1 in the initial residual (:121, 163), 100 in every trial because the
    trial call passes (quat_coeff_prior, vel_coeff) = (1, 100) into (vel_coeff, quat_coeff) (:175) -- and the trial's
    dynamics residual is evaluated with quat_coeff_prior = 1 instead of quat_coeff = 100 (:172, 174).  In the
    landmark-only phase the prior is switched off but still contributes 6 zeros per pose to both means (:609-612).
    """
    states = np.asarray(states, dtype=np.float64)
    n = states.shape[0]
    ii = np.asarray(ii, dtype=np.int64)
    alpha, sigma = lm_schedule(it)
    est, Jg = landmark_project(states, landmarks_xyz, intrinsics, ii, jacobian=True)
    r_obs = landmarks - est
    w, c_obs, wmax = robust_weights(r_obs, it, confidences)
    H, b = accumulate(Jg, w, r_obs, ii, n)
    if initialize:
        r_pred = np.zeros((n - 1, 6))
        E = F = r_orb = qgrad = Hd = Hu = Hl = None
    else:
        r_orb, E, F = orbit_factor(states, time_idx, jacobian=True, hop=hop)
        f, qgrad, Hd, Hu, Hl = attitude_factor(states, cumrot, jacobian=True)
        r_pred = np.concatenate([r_orb, f[:, None]], -1)
    bands, rhs = assemble(H, b, 1.0, float(sigma), E, F, r_orb, qgrad, Hd, Hu, Hl, initialize)
    sq = np.sqrt(sigma)
    r_prior = np.zeros(0)
    if prior is not None:
        states_prior, hessian_state = (np.asarray(a, dtype=np.float64) for a in prior)
        if initialize:
            r_prior = np.zeros((n, 6))
        else:
            rp, Jp = prior_factor(states, states_prior, hessian_state, vel_coeff=1.0)
            bands[:, 1] += np.einsum("irc,ird->icd", Jp, Jp)
            rhs -= np.einsum("irc,ir->ic", Jp, rp)
            r_prior = np.concatenate([rp, np.full((n, 1), 1.0)], -1)      # quat_coeff_prior = 1
    init_residual = np.abs(np.concatenate([r_obs.reshape(-1), r_pred.reshape(-1) * sq, r_prior.reshape(-1)])).mean()
    if debug is not None:
        debug.update(est=est, Jg=Jg, r_obs=r_obs, w=w, c_obs=c_obs, wmax=wmax, H=H, b=b, r_pred=r_pred,
                     E=E, F=F, qgrad=qgrad, Hd=Hd, Hu=Hu, Hl=Hl, bands=bands, rhs=rhs,
                     init_residual=init_residual, trials=[])
    lam = lamda_init
    n_trials = 0
    while True:
        lam32 = float(np.float32(lam))          # torch.eye(...) is float32 (BA_filtering.py:54)
        A = bands.copy()
        A[:, 1] += lam32 * np.eye(9)
        dpose = solve_tridiag(A, rhs, method=solver)
        states_new = retract(states, dpose)
        est1 = landmark_project(states_new, landmarks_xyz, intrinsics, ii, jacobian=False)
        r_obs1 = (landmarks - est1) * w[:, None]
        r_pred1 = dynamics_residual(states_new, cumrot, time_idx, initialize, hop=hop) * sq
        r_prior1 = np.zeros(0)
        if prior is not None:
            if initialize:
                r_prior1 = np.zeros((n, 6))
            else:
                r_pred1[:, 6] *= 1.0 / QUAT_COEFF          # predict(..., quat_coeff_prior = 1, ...) in the trial
                rp1, _ = prior_factor(states_new, states_prior, hessian_state, vel_coeff=1.0)
                r_prior1 = np.concatenate([rp1, np.full((n, 1), 100.0)], -1)
        residual = np.abs(np.concatenate([r_obs1.reshape(-1), r_pred1.reshape(-1), r_prior1.reshape(-1)])).mean()
        n_trials += 1
        if debug is not None:
            debug["trials"].append(dict(lam=lam, lam32=lam32, A=A, dpose=dpose, est=est1, residual=residual))
        lam = lam * 10
        if residual < init_residual:
            break
        if lam > 1e4:
            break
    lamda_out = max(min(1e-1, lam * 0.01), 1e-4)
    return states_new, lamda_out, A[-1, 1].copy(), n_trials
