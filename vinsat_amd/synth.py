"""Deterministic synthetic orbit + detection generator (host side, NumPy).

Produces the two arrays the reference's simulator writes and its OD driver reads
(formats: reference ``sim/nadir_sim.py:140-149, 236, 256``):

* ``orbit_np  [N, 12]`` -- per-second ECEF position in **metres** in columns 0:3
  (columns 3:12 are attitude vectors the nadir branch never reads,
  ``estimation/od_pipe.py:101-103``);
* ``detections [M, 6]`` -- rows ``[frame, lon_deg, lat_deg, u_px, v_px, conf]``
  sorted by frame.

The orbit is a polar LEO integrated at 1 Hz with the same J2 two-body model the
estimator assumes (mu = 398600.4418 km^3/s^2, J2 term 1.75553e10, the constants of
``estimation/trajgen_pipe.py:130-152`` / ``BA/BA_utils.py:883-899``), so the
dynamics factor is consistent with the data.  Pixel measurements are the pinhole
projection at the nadir ground-truth pose plus Gaussian noise.

Named configurations (poses n, observations per pose, stride s) follow
BASELINE.json's configs C1..C5.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import frames

MU = 398600.4418
J2C = 1.75553e10
J2_MAT = np.array([[6.0, -1.5, -1.5], [6.0, -1.5, -1.5], [3.0, -4.5, -4.5]])

# fx, fy, cx, cy of the flight camera (row 0 of estimation/landmarks/intrinsics.csv)
INTRINSICS = np.array([3547.85, 3547.85, 2304.0, 1296.0])


@dataclass(frozen=True)
class WindowConfig:
    name: str
    n_poses: int
    obs_per_pose: int
    stride: int
    t0: int = 10

    @property
    def n_obs(self) -> int:
        return self.n_poses * self.obs_per_pose


CONFIGS = {
    "C1": WindowConfig("C1", 10, 20, 5),
    "C2": WindowConfig("C2", 100, 50, 5),
    "C3": WindowConfig("C3", 500, 100, 5),
    "C4": WindowConfig("C4", 500, 400, 5),
    "C5": WindowConfig("C5", 2000, 250, 3),
}


def _accel(x):
    r = x[:3]
    rn = np.linalg.norm(r)
    return -(MU / rn ** 3) * r + (J2C / rn ** 7) * (J2_MAT @ (r ** 2)) * r


def _deriv(x):
    return np.concatenate([x[3:], _accel(x)])


def rk4_step(x, h=1.0):
    f1 = _deriv(x)
    f2 = _deriv(x + 0.5 * h * f1)
    f3 = _deriv(x + 0.5 * h * f2)
    f4 = _deriv(x + h * f3)
    return x + (h / 6.0) * (f1 + 2 * f2 + 2 * f3 + f4)


def elements_to_eci(a, e, inc, raan, argp, nu):
    """Classical orbital elements -> ECI state [r(3) km, v(3) km/s]."""
    p = a * (1 - e * e)
    r = p / (1 + e * np.cos(nu))
    r_pf = np.array([r * np.cos(nu), r * np.sin(nu), 0.0])
    v_pf = np.sqrt(MU / p) * np.array([-np.sin(nu), e + np.cos(nu), 0.0])

    def rz(t):
        return np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]])

    def rx(t):
        return np.array([[1, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]])

    Q = rz(raan) @ rx(inc) @ rz(argp)
    return np.concatenate([Q @ r_pf, Q @ v_pf])


def integrate_orbit(n_seconds, x0=None):
    """1 Hz J2 RK4 trajectory, [n_seconds, 6] ECI km / km/s."""
    if x0 is None:
        x0 = elements_to_eci(6978.0, 0.005, np.pi / 2, np.pi, np.pi, np.pi)
    out = np.empty((n_seconds, 6))
    x = np.array(x0, dtype=np.float64)
    for k in range(n_seconds):
        out[k] = x
        x = rk4_step(x)
    return out


def quat_to_matrix(q):
    """Rotation matrices of unit quaternions [..., 4] (scalar last)."""
    x, y, z, w = np.moveaxis(q, -1, 0)
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1),
    ], -2)


def project(pos, quat, xyz, intr):
    """Pinhole projection of ECI points into cameras (one pose per point)."""
    R = quat_to_matrix(quat / np.linalg.norm(quat, axis=-1, keepdims=True))
    pc = np.einsum("kji,kj->ki", R, xyz - pos)
    z = np.maximum(pc[:, 2], 0.1)
    return np.stack([intr[0] * pc[:, 0] / z + intr[2], intr[1] * pc[:, 1] / z + intr[3]], -1)


def make_sequence(cfg: WindowConfig | str, seed: int = 0, pixel_noise: float = 1.0, conf: float = 0.95):
    """Return ``(detections [M,6], orbit_np [N,12])`` for a named configuration."""
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    n_sec = cfg.t0 + cfg.n_poses * cfg.stride + 5
    traj = integrate_orbit(n_sec)
    times = np.arange(n_sec)
    ecef_km = frames.eci_to_ecef(traj[:, :3], times)
    orbit_np = np.zeros((n_sec, 12))
    orbit_np[:, :3] = ecef_km * 1000.0

    # what the driver will reconstruct as ground truth (ECEF m -> ECI km)
    xe, ye, ze = frames.ecef_to_eci(orbit_np[:, 0] / 1000, orbit_np[:, 1] / 1000, orbit_np[:, 2] / 1000, times)
    pos_eci = np.stack([xe, ye, ze], -1)

    rng = np.random.default_rng(seed)
    frames_t = cfg.t0 + cfg.stride * np.arange(cfg.n_poses)
    k = cfg.obs_per_pose
    frame_col = np.repeat(frames_t, k)
    sub = ecef_km[frames_t]
    sub_lat = np.rad2deg(np.arcsin(sub[:, 2] / np.linalg.norm(sub, axis=-1)))
    sub_lon = np.rad2deg(np.arctan2(sub[:, 1], sub[:, 0]))
    lat = np.repeat(sub_lat, k) + rng.uniform(-1.2, 1.2, size=cfg.n_obs)
    lon = np.repeat(sub_lon, k) + rng.uniform(-2.0, 2.0, size=cfg.n_obs)
    xyz = frames.latlon_to_eci(lat, lon, frame_col)
    p = np.repeat(pos_eci[frames_t], k, axis=0)
    q = np.repeat(frames.nadir_quaternion(pos_eci[frames_t]), k, axis=0)
    uv = project(p, q, xyz, INTRINSICS) + rng.normal(0.0, pixel_noise, size=(cfg.n_obs, 2))
    det = np.stack([frame_col.astype(np.float64), lon, lat, uv[:, 0], uv[:, 1], np.full(cfg.n_obs, conf)], -1)
    return det, orbit_np


def make_subwindow(cfg: WindowConfig | str, n_poses: int, **kw):
    """The first ``n_poses`` frames of a named configuration's sequence: the same orbit, the same detections, cut after
    frame ``n_poses`` (orbit rows kept up to 5 s past the last frame, as :func:`make_sequence` does)."""
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    det, orbit = make_sequence(cfg, **kw)
    t_end = cfg.t0 + cfg.stride * n_poses
    return det[det[:, 0] < t_end].copy(), orbit[: t_end + 5].copy()


def make_tracked_landmarks(n_poses=40, n_landmarks=600, stride=5, seed=0, pixel_noise=1.0, catalogue_sigma_km=0.05, t0=10):
    """Synthetic free-landmark problem (the add-on of ``vinsat_amd/schur.py``; the reference has no such data): ground
    landmarks scattered along the ground track, each seen from every frame whose footprint (+-1.2 deg latitude, +-2 deg
    longitude around the sub-satellite point, as :func:`make_sequence`) contains it -- tracks of several frames.

    Returns a dict: ``states_gt [n,10]`` (nadir attitude, finite-difference velocity), ``X_true [L,3]`` ECI km (a landmark is
    taken at the Earth rotation angle of the first frame that sees it: the add-on treats landmarks as inertial points, so
    frames are kept within ~1 minute where the ground moves less than the catalogue uncertainty would allow -- synthetic
    data for the solver, not a model of the real scene), ``X0`` = truth + catalogue error, ``uv``, ``pose_of_row``,
    ``landmark_of_row``, ``intrinsics [n,4]``.
    """
    rng = np.random.default_rng(seed)
    n_sec = t0 + n_poses * stride + 5
    traj = integrate_orbit(n_sec)
    times = t0 + stride * np.arange(n_poses)
    pos = traj[times, :3]
    vel = traj[times, 3:]
    quat = frames.nadir_quaternion(pos)
    states = np.concatenate([pos, quat, vel], axis=1)
    # landmarks: points on a sphere of Earth radius under the track, scattered across the footprint
    sub = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    along = rng.uniform(0, n_poses - 1, n_landmarks)
    i0 = np.floor(along).astype(int)
    f = (along - i0)[:, None]
    c = sub[i0] * (1 - f) + sub[np.minimum(i0 + 1, n_poses - 1)] * f
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    e1 = np.cross(c, np.array([0.0, 0.0, 1.0]))
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(c, e1)
    off1, off2 = np.deg2rad(rng.uniform(-1.8, 1.8, n_landmarks)), np.deg2rad(rng.uniform(-1.0, 1.0, n_landmarks))
    X = c + e1 * off1[:, None] + e2 * off2[:, None]
    X = frames.A_EARTH_KM * X / np.linalg.norm(X, axis=1, keepdims=True)
    rows_p, rows_l, uvs = [], [], []
    for i in range(n_poses):
        uv = project(np.repeat(pos[i:i + 1], n_landmarks, 0), np.repeat(quat[i:i + 1], n_landmarks, 0), X, INTRINSICS)
        R = quat_to_matrix(quat[i])
        depth = (X - pos[i]) @ R[:, 2]
        horizon = np.sqrt(max(pos[i] @ pos[i] - frames.A_EARTH_KM ** 2, 1.0))      # beyond it the Earth is in the way
        vis = (uv[:, 0] > 50) & (uv[:, 0] < 4558) & (uv[:, 1] > 50) & (uv[:, 1] < 2542) & (depth > 1.0) & (depth < horizon)
        idx = np.nonzero(vis)[0]
        rows_p.append(np.full(idx.size, i))
        rows_l.append(idx)
        uvs.append(uv[idx] + rng.normal(0.0, pixel_noise, (idx.size, 2)))
    pose_of_row, landmark_of_row, uv = np.concatenate(rows_p), np.concatenate(rows_l), np.concatenate(uvs)
    # drop landmarks nobody sees, renumber
    seen = np.unique(landmark_of_row)
    remap = -np.ones(n_landmarks, dtype=np.int64)
    remap[seen] = np.arange(seen.size)
    X = X[seen]
    landmark_of_row = remap[landmark_of_row]
    X0 = X + rng.normal(0.0, catalogue_sigma_km, X.shape)
    return dict(states_gt=states, X_true=X, X0=X0, uv=uv, pose_of_row=pose_of_row, landmark_of_row=landmark_of_row,
                intrinsics=np.repeat(INTRINSICS[None], n_poses, 0), sigma=catalogue_sigma_km)


def make_two_pass_sequence(n_poses=12, obs_per_pose=6, stride=5, gap=1500, seed=4, pixel_noise=1.0, conf=0.95, tail=140):
    """Two ground-station-like passes separated by ``gap`` seconds without detections.

    Exercises what a single dense pass does not: the batch cut of the driver (``od_pipe.py:898-905``), knot poses
    every 1000 s that carry no observation, dynamics factors spanning ~1000 RK4 steps and dead-reckoning between
    batches.  Returns ``(detections [M,6], orbit_np [N,12])``.
    """
    t0 = 10
    n_sec = t0 + n_poses * stride + gap + n_poses * stride + tail
    traj = integrate_orbit(n_sec)
    times = np.arange(n_sec)
    ecef_km = frames.eci_to_ecef(traj[:, :3], times)
    orbit_np = np.zeros((n_sec, 12))
    orbit_np[:, :3] = ecef_km * 1000.0
    xe, ye, ze = frames.ecef_to_eci(orbit_np[:, 0] / 1000, orbit_np[:, 1] / 1000, orbit_np[:, 2] / 1000, times)
    pos_eci = np.stack([xe, ye, ze], -1)
    rng = np.random.default_rng(seed)
    first = t0 + stride * np.arange(n_poses)
    frames_t = np.concatenate([first, first + gap])
    k = obs_per_pose
    frame_col = np.repeat(frames_t, k)
    sub = ecef_km[frames_t]
    sub_lat = np.rad2deg(np.arcsin(sub[:, 2] / np.linalg.norm(sub, axis=-1)))
    sub_lon = np.rad2deg(np.arctan2(sub[:, 1], sub[:, 0]))
    m = frame_col.size
    lat = np.repeat(sub_lat, k) + rng.uniform(-1.2, 1.2, size=m)
    lon = np.repeat(sub_lon, k) + rng.uniform(-2.0, 2.0, size=m)
    xyz = frames.latlon_to_eci(lat, lon, frame_col)
    p = np.repeat(pos_eci[frames_t], k, axis=0)
    q = np.repeat(frames.nadir_quaternion(pos_eci[frames_t]), k, axis=0)
    uv = project(p, q, xyz, INTRINSICS) + rng.normal(0.0, pixel_noise, size=(m, 2))
    det = np.stack([frame_col.astype(np.float64), lon, lat, uv[:, 0], uv[:, 1], np.full(m, conf)], -1)
    return det, orbit_np


def make_multi_pass_sequence(passes=6, n_poses=20, obs_per_pose=50, stride=5, gap=1500, seed=5, pixel_noise=1.0, conf=0.95, tail=140):
    """``passes`` passes of ``n_poses`` frames, ``gap`` seconds apart -- what a sequence looks like after several passes: the window
    of its last batch holds every earlier pass, a knot pose every 1000 s and therefore one or two gaps of hundreds of seconds per
    pass (``od_pipe.py:213-221``).  Same construction as :func:`make_two_pass_sequence` (which stays as it is: fixtures were made
    from it).  Returns ``(detections [M,6], orbit_np [N,12])``."""
    t0 = 10
    n_sec = t0 + (passes - 1) * gap + n_poses * stride + tail
    traj = integrate_orbit(n_sec)
    times = np.arange(n_sec)
    ecef_km = frames.eci_to_ecef(traj[:, :3], times)
    orbit_np = np.zeros((n_sec, 12))
    orbit_np[:, :3] = ecef_km * 1000.0
    xe, ye, ze = frames.ecef_to_eci(orbit_np[:, 0] / 1000, orbit_np[:, 1] / 1000, orbit_np[:, 2] / 1000, times)
    pos_eci = np.stack([xe, ye, ze], -1)
    rng = np.random.default_rng(seed)
    first = t0 + stride * np.arange(n_poses)
    frames_t = np.concatenate([first + p * gap for p in range(passes)])
    k = obs_per_pose
    frame_col = np.repeat(frames_t, k)
    sub = ecef_km[frames_t]
    sub_lat = np.rad2deg(np.arcsin(sub[:, 2] / np.linalg.norm(sub, axis=-1)))
    sub_lon = np.rad2deg(np.arctan2(sub[:, 1], sub[:, 0]))
    m = frame_col.size
    lat = np.repeat(sub_lat, k) + rng.uniform(-1.2, 1.2, size=m)
    lon = np.repeat(sub_lon, k) + rng.uniform(-2.0, 2.0, size=m)
    xyz = frames.latlon_to_eci(lat, lon, frame_col)
    p = np.repeat(pos_eci[frames_t], k, axis=0)
    q = np.repeat(frames.nadir_quaternion(pos_eci[frames_t]), k, axis=0)
    uv = project(p, q, xyz, INTRINSICS) + rng.normal(0.0, pixel_noise, size=(m, 2))
    det = np.stack([frame_col.astype(np.float64), lon, lat, uv[:, 0], uv[:, 1], np.full(m, conf)], -1)
    return det, orbit_np
