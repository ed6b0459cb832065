"""Coordinate-frame helpers used by the OD driver (host side, NumPy fp64).

These restate the reference's conventions so that the arrays handed to ``BA`` are
the ones the reference's own data preparation would produce:

* GMST model ``theta = 280.16 deg + t * 360/86164.100352 deg/s``
  (reference ``estimation/BA/BA_utils.py:1172-1173``);
* ECEF<->ECI as a rotation about z by that angle (``BA_utils.py:1185-1218``);
* WGS84-like ellipsoid ``a=6378.137 km, b=6356.752 km`` for geodetic -> ECEF
  (``BA_utils.py:1178-1180, 1221-1236``);
* nadir-pointing camera attitude from position (``BA_utils.py:1276-1292``).

Units: km, seconds, degrees for lat/lon.  Quaternions are ``[x, y, z, w]``.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import transform

THETA_G0_DEG = 280.16
OMEGA_EARTH_DEG_PER_SEC = 360 / 86164.100352
A_EARTH_KM = 6378.137
B_EARTH_KM = 6356.752
ECC_EARTH = np.sqrt(1 - (B_EARTH_KM ** 2 / A_EARTH_KM ** 2))


def gmst_deg(times):
    return THETA_G0_DEG + OMEGA_EARTH_DEG_PER_SEC * np.asarray(times)


def ecef_to_eci(x_ecef, y_ecef, z_ecef, times):
    """Rotate ECEF coordinates into ECI at ``times`` (s). ``BA_utils.py:1185-1195``."""
    theta = np.deg2rad(gmst_deg(times))
    x_eci = x_ecef * np.cos(theta) - y_ecef * np.sin(theta)
    y_eci = x_ecef * np.sin(theta) + y_ecef * np.cos(theta)
    return x_eci, y_eci, z_ecef


def earth_rotation(times):
    """Rz(theta_G(t)) stacked as [T,3,3]; ECI->ECEF. ``BA_utils.py:1197-1208``."""
    th = np.deg2rad(gmst_deg(times))
    zero = np.zeros_like(th)
    one = np.ones_like(th)
    return np.stack([
        np.stack([np.cos(th), np.sin(th), zero], axis=-1),
        np.stack([-np.sin(th), np.cos(th), zero], axis=-1),
        np.stack([zero, zero, one], axis=-1),
    ], axis=-2)


def eci_to_ecef(r_eci, times):
    """``BA_utils.py:1210-1218``."""
    Rz = earth_rotation(times)
    return (Rz * r_eci[:, None, :]).sum(axis=-1)


def geodetic_to_ecef(latitude, longitude, altitude):
    """Geodetic (deg, deg, km) -> ECEF km. ``BA_utils.py:1221-1236``."""
    phi = np.deg2rad(latitude)
    lam = np.deg2rad(longitude)
    N = A_EARTH_KM / np.sqrt(1 - (ECC_EARTH ** 2 * np.sin(phi) ** 2))
    x = (N + altitude) * np.cos(phi) * np.cos(lam)
    y = (N + altitude) * np.cos(phi) * np.sin(lam)
    z = ((B_EARTH_KM ** 2 / A_EARTH_KM ** 2) * N + altitude) * np.sin(phi)
    return x, y, z


def latlon_to_eci(lat, lon, times, altitude=None):
    """Ground landmark (deg) observed at ``times`` -> ECI km. ``BA_utils.py:1238-1251``."""
    if altitude is None:
        altitude = np.zeros(lat.shape[0])
    x, y, z = geodetic_to_ecef(lat, lon, altitude)
    xe, ye, ze = ecef_to_eci(x, y, z, times)
    return np.stack([xe, ye, ze], axis=-1)


def nadir_quaternion(pos_eci):
    """Camera attitude of a nadir-pointing satellite from its ECI position.

    Camera z points at the Earth's centre, x is minus the (north x z) direction and
    y completes the frame; returned as scipy's ``[x,y,z,w]`` quaternion of the
    camera->ECI rotation matrix ``[xc yc zc]``.  ``BA_utils.py:1276-1292``.
    """
    zc = -pos_eci / (np.linalg.norm(pos_eci, axis=-1)[..., None])
    north = np.array([0, 0, 1])[None]
    rc = np.cross(north, zc)
    rc = rc / np.linalg.norm(rc, axis=-1)[..., None]
    xc = -rc
    yc = np.cross(rc, zc)
    R = np.stack([xc, yc, zc], axis=-1)
    return transform.Rotation.from_matrix(R).as_quat()


def finite_difference(x, dt):
    """Forward difference padded with a zero row. ``BA_utils.py:1370-1373``."""
    d = (x[1:] - x[:-1]) / dt
    return np.concatenate([d, np.zeros((1, x.shape[1]))], axis=0)
