"""Propagated-covariance prior for ``BA_reg``: host-side counterpart of the reference's
``propagate_dynamics_cov_init`` (``estimation/BA/BA_utils.py:227-248``) and its helpers
``propagate_orbit_dynamics_cov_init`` (``:138-157``), ``propagate_rotation_dynamics_cov_init`` (``:207-225``),
``compute_orbit_jacobian`` (``:133-136``), ``compute_rot_jacobian`` / ``qtoQ`` / ``L`` (``:171-205``).

The last 9x9 Hessian block a ``BA`` call returns is split into its (position, velocity) 6x6 corners and its 3x3
rotation block, both are inverted to covariances, the covariances are pushed through the 1 Hz dynamics
(``Sigma <- J Sigma J^T`` with the per-step Jacobian) first over the ``tdiff`` seconds up to the next window and
then over its ``duration`` seconds, and inverted back: one prior state and one information matrix per second of the
new window -- the ``states_prior`` / ``hessian_state_t`` / ``hessian_rot_t`` arguments of ``BA_reg``.

Host code (NumPy) like the reference's; the per-step Jacobian is the closed form of what the reference obtains by
autograd through one RK4 step.  One sequence at a time (the reference's Jacobian reshape is only meaningful for a
batch of one).
"""
from __future__ import annotations

import numpy as np

from . import quat
from .synth import J2C, J2_MAT, MU


def _accel(r):
    rn = np.linalg.norm(r)
    return -(MU / rn ** 3) * r + (J2C / rn ** 7) * (J2_MAT @ (r ** 2)) * r


def _accel_jac(r):
    """d a / d r of the J2 acceleration of ``orbit_dynamics`` (BA_utils.py:883-899)."""
    rn = np.linalg.norm(r)
    s = J2_MAT @ (r ** 2)
    A = -(MU / rn ** 3) * np.eye(3) + 3.0 * MU / rn ** 5 * np.outer(r, r)
    A += (J2C / rn ** 7) * (np.diag(s) + 2.0 * (r[:, None] * J2_MAT) * r[None, :]) - 7.0 * J2C / rn ** 9 * np.outer(s * r, r)
    return A


def _deriv(x):
    return np.concatenate([x[3:], _accel(x[:3])])


def _deriv_jac(x):
    J = np.zeros((6, 6))
    J[:3, 3:] = np.eye(3)
    J[3:, :3] = _accel_jac(x[:3])
    return J


def rk4_step_with_jacobian(x, h=1.0):
    """One RK4 step of the orbit state [r, v] and d x_next / d x (``RK4`` BA_utils.py:901-912, ``compute_orbit_jacobian``)."""
    f1 = _deriv(x)
    D1 = _deriv_jac(x)
    x2 = x + 0.5 * h * f1
    f2 = _deriv(x2)
    D2 = _deriv_jac(x2) @ (np.eye(6) + 0.5 * h * D1)
    x3 = x + 0.5 * h * f2
    f3 = _deriv(x3)
    D3 = _deriv_jac(x3) @ (np.eye(6) + 0.5 * h * D2)
    x4 = x + h * f3
    f4 = _deriv(x4)
    D4 = _deriv_jac(x4) @ (np.eye(6) + h * D3)
    return x + (h / 6.0) * (f1 + 2 * f2 + 2 * f3 + f4), np.eye(6) + (h / 6.0) * (D1 + 2 * D2 + 2 * D3 + D4)


def _hat(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rot_jacobian(omega, dt):
    """``compute_rot_jacobian`` (BA_utils.py:202-205) as written: ``qtoQ`` reads its argument scalar-FIRST
    (``L``: s = q[0], v = q[1:], BA_utils.py:171-183) although ``quaternion_exp`` returns [x, y, z, w]."""
    dq = quat.qexp(-dt * np.asarray(omega, dtype=np.float64))
    s, v = dq[0], dq[1:]
    Lq = np.zeros((4, 4))
    Lq[0, 0] = s
    Lq[0, 1:] = -v
    Lq[1:, 0] = v
    Lq[1:, 1:] = s * np.eye(3) + _hat(v)
    T = np.diag([1.0, -1.0, -1.0, -1.0])
    Hm = np.concatenate([np.zeros((1, 3)), np.eye(3)], 0)
    return Hm.T @ ((T @ Lq) @ (T @ Lq)) @ Hm


def propagate_orbit_cov(position, velocity, duration, dt, sigma, only_end=False):
    """``propagate_orbit_dynamics_cov_init`` (BA_utils.py:138-157), Q = 0."""
    x = np.concatenate([position, velocity]).astype(np.float64)
    xs, sig = [x], [np.asarray(sigma, dtype=np.float64)]
    for _ in range(duration):
        x, J = rk4_step_with_jacobian(x, float(dt))
        xs.append(x)
        sig.append(J @ sig[-1] @ J.T)
    xs, sig = np.stack(xs), np.stack(sig)
    if only_end:
        return xs[-1, :3], xs[-1, 3:], sig[-1]
    return xs[:, :3], xs[:, 3:], sig


def propagate_rotation_cov(quaternion, omegas, duration, dt, sigma_rot, only_end=False):
    """``propagate_rotation_dynamics_cov_init`` (BA_utils.py:207-225), Q_rot = 0."""
    q = np.asarray(quaternion, dtype=np.float64)
    qs, sig = [q], [np.asarray(sigma_rot, dtype=np.float64)]
    for i in range(duration):
        J = rot_jacobian(omegas[i], dt)
        q = quat.qmul(q, quat.qexp(dt * omegas[i]))
        qs.append(q)
        sig.append(J @ sig[-1] @ J.T)
    qs, sig = np.stack(qs), np.stack(sig)
    if only_end:
        return qs[-1], sig[-1]
    return qs, sig


def propagate_dynamics_cov_init(state, velocity, hessian, omega, tdiff, duration, dt=1):
    """Counterpart of the reference's ``propagate_dynamics_cov_init`` for one sequence.

    state [10] = last pose estimate [p, q, v] (position and attitude are read), velocity [3] = the velocity the
    driver carries beside the states, hessian [9,9] = ``last_hessian`` of the previous ``BA`` call, omega
    [tdiff + duration, 3] body rates.  Returns ``(states_t [duration+1, 10], velocities_t [duration+1, 3],
    hessian_state_t [duration+1, 6, 6], hessian_rot_t [duration+1, 3, 3])``.
    """
    state = np.asarray(state, dtype=np.float64)
    H = np.asarray(hessian, dtype=np.float64).reshape(9, 9)
    H_state = np.block([[H[:3, :3], H[:3, 6:]], [H[6:, :3], H[6:, 6:]]])
    cov_rot = np.linalg.inv(H[3:6, 3:6])
    cov_state = np.linalg.inv(H_state)
    omega = np.asarray(omega, dtype=np.float64)
    p_beg, v_beg, cov_state_beg = propagate_orbit_cov(state[:3], np.asarray(velocity, dtype=np.float64), tdiff, dt, cov_state, only_end=True)
    q_beg, cov_rot_beg = propagate_rotation_cov(state[3:7], omega[:tdiff], tdiff, dt, cov_rot, only_end=True)
    p_t, v_t, cov_state_t = propagate_orbit_cov(p_beg, v_beg, duration, dt, cov_state_beg)
    q_t, cov_rot_t = propagate_rotation_cov(q_beg, omega[tdiff:], duration, dt, cov_rot_beg)
    states_t = np.concatenate([p_t, q_t, v_t], -1)
    return states_t, v_t, np.linalg.inv(cov_state_t), np.linalg.inv(cov_rot_t)
