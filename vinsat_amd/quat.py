"""Scalar-last quaternion algebra on NumPy arrays (host side).

Conventions follow the reference (``estimation/BA/BA_utils.py:949-1000``):
``q = [x, y, z, w]``, Hamilton product, ``exp`` maps a rotation vector of angle
``|d|`` to the half-angle quaternion, ``log`` is its inverse (``2*acos(w)``).
"""
from __future__ import annotations

import numpy as np


def qmul(q1, q2):
    """Hamilton product, broadcasting over leading dims. ``BA_utils.py:992-1000``."""
    x1, y1, z1, w1 = np.moveaxis(q1, -1, 0)
    x2, y2, z2, w2 = np.moveaxis(q2, -1, 0)
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2
    z = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
    return np.stack([x, y, z, w], axis=-1)


def qconj(q):
    return np.concatenate([-q[..., :-1], q[..., -1:]], axis=-1)


def qexp(d_theta):
    """Rotation vector -> quaternion; identity below 1e-16 rad. ``BA_utils.py:970-985``."""
    theta = np.linalg.norm(d_theta, axis=-1)[..., None]
    mask = (theta < 1e-16).astype(np.float64)
    ident = np.concatenate([np.zeros_like(d_theta), np.ones_like(theta)], axis=-1)
    q = np.concatenate([d_theta * np.sin(theta / 2) / (theta + 1e-16), np.cos(theta / 2)], axis=-1)
    return ident * mask + q * (1 - mask)


def qlog(q):
    """Quaternion -> rotation vector (NaN at the identity, as the reference). ``BA_utils.py:949-967``."""
    q = np.clip(q / np.linalg.norm(q, axis=-1)[..., None], -1, 1)
    theta = 2 * np.arccos(q[..., -1])
    with np.errstate(invalid="ignore", divide="ignore"):
        n = q[..., :-1] / np.sin(theta / 2)[..., None]
        return n * theta[..., None]


def omega_from_quats(quat, dt):
    """Body rate between consecutive attitudes, zero-padded. ``BA_utils.py:1361-1367``."""
    dq = qmul(qconj(quat[:-1]), quat[1:])
    dq = dq / np.linalg.norm(dq, axis=-1)[..., None]
    omega = qlog(dq) / dt
    return np.concatenate([omega, np.zeros((1, 3))], axis=0)


def cumulative_rotations(omegas, dt):
    """Running product of exp(dt*omega) along axis -2. ``BA_utils.py:278-288``.

    ``omegas`` is [..., N, 3]; returns [..., N, 4].
    """
    rot = qexp(dt * omegas)
    out = np.empty_like(rot)
    out[..., 0, :] = rot[..., 0, :]
    for i in range(1, rot.shape[-2]):
        out[..., i, :] = qmul(out[..., i - 1, :], rot[..., i, :])
    return out
