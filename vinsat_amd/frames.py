"""Coordinate frames of the OD driver (host side, NumPy fp64).

The arrays handed to ``BA`` must be bit-for-bit the ones the reference's data preparation produces
(``tests/test_od_pipe_host.py`` compares them with fixtures captured from the reference), so the models and the
order of the floating-point operations are the reference's; the code is organised around two primitives of its own:

* :func:`spin_z` -- rotation of the equatorial components about the polar axis by a signed angle.  Earth-fixed
  -> inertial is ``spin_z(+theta_G)``, inertial -> Earth-fixed ``spin_z(-theta_G)`` with the Greenwich angle
  ``theta_G(t) = 280.16 deg + t * 360/86164.100352 deg/s`` (``estimation/BA/BA_utils.py:1172-1218``);
* :func:`camera_triad` -- the nadir camera frame from a position (``BA_utils.py:1276-1292``).

Ellipsoid: ``a = 6378.137 km, b = 6356.752 km`` (``BA_utils.py:1178-1180, 1221-1236``).
Units: km, seconds, degrees for lat/lon.  Quaternions are ``[x, y, z, w]``.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

THETA_G0_DEG = 280.16
OMEGA_EARTH_DEG_PER_SEC = 360 / 86164.100352
A_EARTH_KM = 6378.137
B_EARTH_KM = 6356.752
AXIS_RATIO_SQ = B_EARTH_KM ** 2 / A_EARTH_KM ** 2
ECC_EARTH = np.sqrt(1 - AXIS_RATIO_SQ)


def greenwich_angle(times):
    """theta_G in radians at ``times`` seconds."""
    return np.deg2rad(THETA_G0_DEG + OMEGA_EARTH_DEG_PER_SEC * np.asarray(times))


def spin_z(x, y, angle):
    """Components (x, y) turned by ``angle`` about +z; z is untouched by such a rotation and not passed."""
    c, s = np.cos(angle), np.sin(angle)
    return x * c - y * s, x * s + y * c


def ecef_to_eci(x_ecef, y_ecef, z_ecef, times):
    """Earth-fixed -> inertial at ``times`` (component arrays in, component arrays out)."""
    x, y = spin_z(x_ecef, y_ecef, greenwich_angle(times))
    return x, y, z_ecef


def eci_to_ecef(r_eci, times):
    """Inertial -> Earth-fixed for an ``[T, 3]`` array of positions at ``times``."""
    x, y = spin_z(r_eci[:, 0], r_eci[:, 1], -greenwich_angle(times))
    return np.column_stack([x, y, r_eci[:, 2]])


def geodetic_to_ecef(latitude, longitude, altitude):
    """Geodetic (deg, deg, km) -> Earth-fixed km on the ellipsoid above."""
    lat, lon = np.deg2rad(latitude), np.deg2rad(longitude)
    # prime-vertical radius of curvature; the eccentricity enters squared AFTER its square root was taken, as in the
    # reference (not the same bits as 1 - b^2/a^2)
    n_phi = A_EARTH_KM / np.sqrt(1 - (ECC_EARTH ** 2 * np.sin(lat) ** 2))
    ring = (n_phi + altitude) * np.cos(lat)             # distance from the polar axis
    return ring * np.cos(lon), ring * np.sin(lon), (AXIS_RATIO_SQ * n_phi + altitude) * np.sin(lat)


def latlon_to_eci(lat, lon, times, altitude=None):
    """Ground landmark (deg) observed at ``times`` -> inertial km, ``[M, 3]`` (``BA_utils.py:1238-1251``)."""
    if altitude is None:
        altitude = np.zeros(np.shape(lat)[0])
    return np.stack(ecef_to_eci(*geodetic_to_ecef(lat, lon, altitude), times), axis=-1)


def camera_triad(pos_eci):
    """Axes of a nadir-pointing camera at ``pos_eci`` ``[..., 3]``: boresight ``down`` towards the Earth's centre,
    image x along ``-(pole x down)`` (normalised), image y completing the right-handed frame.  With the pole at
    ``(0, 0, 1)`` the cross products collapse to the two-component expressions below."""
    down = -pos_eci / np.linalg.norm(pos_eci, axis=-1)[..., None]
    dx, dy, dz = down[..., 0], down[..., 1], down[..., 2]
    zero = np.zeros_like(dx)
    side = np.stack([-dy, dx, zero], axis=-1)                       # pole x down
    side = side / np.linalg.norm(side, axis=-1)[..., None]
    sx, sy = side[..., 0], side[..., 1]
    img_y = np.stack([sy * dz, -(sx * dz), sx * dy - sy * dx], axis=-1)   # side x down (side_z = 0)
    return -side, img_y, down


def nadir_quaternion(pos_eci):
    """Camera->inertial attitude of a nadir-pointing satellite as a ``[x, y, z, w]`` quaternion."""
    return Rotation.from_matrix(np.stack(camera_triad(pos_eci), axis=-1)).as_quat()


def finite_difference(x, dt):
    """Forward difference along axis 0, last row zero (``BA_utils.py:1370-1373``)."""
    out = np.zeros_like(x, dtype=np.float64)
    out[:-1] = np.diff(x, axis=0) / dt
    return out
