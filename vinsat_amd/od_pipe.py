"""Orbit-determination driver: the host-side counterpart of the reference's
``estimation/od_pipe.py:streaming_version`` (lines 911-1062) and its helpers.

The driver owns data preparation (frame bookkeeping, ECEF->ECI, ground-truth nadir
attitude, outlier mask, pose renumbering, IMU pre-integration, initial guess), splits
the detection stream into batches and calls ``BA`` 20 times per batch.  All the
heavy arithmetic is inside ``BA`` (HIP); everything here is NumPy fp64 on the host
and reproduces the reference's integer outputs (``ii``, masks, batch cuts) exactly.

Input formats (reference ``sim/nadir_sim.py:140-149, 236, 256``):
``detections [M,6] = [frame, lon, lat, u, v, conf]`` and ``orbit [N,12]`` with ECEF
metres in columns 0:3.  Outputs: ``(errors, first_detection, times)`` as consumed by
``estimation/errors_eval.py:19-50``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import frames, quat
from .synth import INTRINSICS, quat_to_matrix

NUM_ITERS = 20          # od_pipe.py:918
KNOT_PERIOD = 1000      # od_pipe.py:216-225


@dataclass
class Window:
    """Everything ``BA`` needs for one sequence, without batch dimension."""
    time_idx: np.ndarray          # [T] int64, seconds
    ii: np.ndarray                # [M] int64 pose index of every observation
    landmarks_uv: np.ndarray      # [M,2]
    landmarks_xyz: np.ndarray     # [M,3] ECI km
    confidences: np.ndarray       # [M]
    intrinsics: np.ndarray        # [T,4]
    poses_gt: np.ndarray          # [T,7]
    vel_gt_full: np.ndarray       # [N,3] per-second finite-difference velocity
    quat_gt_full: np.ndarray      # [N,4]
    omega_gt: np.ndarray          # [N,3]
    cumrot_last: np.ndarray       # [T,4] attitude increment over the gap after each pose
    max_gap: int
    mask: np.ndarray              # [M_raw] bool, observations kept
    extras: dict = field(default_factory=dict)

    @property
    def velocities(self):
        return self.vel_gt_full[self.time_idx]

    @property
    def states_gt(self):
        return np.concatenate([self.poses_gt, self.velocities], -1)


def read_detections(detections, orbit, intrinsics=None):
    """Frame bookkeeping + ECEF->ECI (reference ``read_detections`` od_pipe.py:185-251).

    ``orbit`` is converted in place like the reference does (pass a copy).  Returns
    ``orbit, fields, intrinsics, time_idx, ii`` where time_idx holds every frame with a
    detection plus one "knot" every 1000 s.
    """
    det = np.asarray(detections)
    fields = dict(frame=det[:, 0], uv=det[:, 3:5], lonlat=det[:, 1:3], confidence=det[:, 5])
    frame = det[:, 0]
    if frame.size > 1 and np.all(frame[1:] >= frame[:-1]):     # rows in frame order (as the simulator writes them): no sort needed
        first = np.flatnonzero(np.concatenate([[True], frame[1:] != frame[:-1]]))
        uniq, counts = frame[first], np.diff(np.concatenate([first, [frame.size]]))
    else:
        uniq, counts = np.unique(frame, return_counts=True)
    uniq = uniq.astype(np.int64)
    filler = uniq.min() // KNOT_PERIOD + 1
    offset = 0
    time_new = []
    pose_of_frame = np.empty(uniq.size, dtype=np.int64)
    for i, t in enumerate(uniq.tolist()):
        if t == filler * KNOT_PERIOD:
            filler += 1
        while t > filler * KNOT_PERIOD:
            time_new.append(filler * KNOT_PERIOD)
            filler += 1
            offset += 1
        time_new.append(t)
        pose_of_frame[i] = i + offset
    ii = [np.repeat(pose_of_frame, counts)]
    n_sec = orbit.shape[0]
    orbit[:, 0], orbit[:, 1], orbit[:, 2] = frames.ecef_to_eci(
        orbit[:, 0] / 1000, orbit[:, 1] / 1000, orbit[:, 2] / 1000, times=np.arange(n_sec))
    if uniq[-1] < n_sec:
        while filler * KNOT_PERIOD < (n_sec // KNOT_PERIOD) * KNOT_PERIOD + 1:
            time_new.append(filler * KNOT_PERIOD)
            filler += 1
    if intrinsics is None:
        intrinsics = INTRINSICS
    return orbit, fields, np.asarray(intrinsics, dtype=np.float64), np.array(time_new, dtype=np.int64), np.concatenate(ii)


def rows_on_device(detections, ii, pos_gt, rot_gt, intr, device):
    """The per-row part of the preparation on GPU ``device`` (``vba_prepare_rows``): landmark positions ``xyz [M,3]``, their
    reprojection at the ground-truth poses ``proj [M,2]`` and the outlier mask ``[M]`` (reference od_pipe.py:924-930)."""
    _lib, lib = _lib_host()
    det = np.ascontiguousarray(detections, dtype=np.float64)
    ii = np.ascontiguousarray(ii, dtype=np.int64)
    pos = np.ascontiguousarray(pos_gt, dtype=np.float64)
    rot = np.ascontiguousarray(rot_gt, dtype=np.float64).reshape(-1, 9)
    k4 = np.ascontiguousarray(intr, dtype=np.float64)
    M = det.shape[0]
    xyz, proj, mask = np.empty((M, 3)), np.empty((M, 2)), np.empty(M, dtype=np.uint8)
    _lib.check(lib.vba_prepare_rows(int(device), M, _lib.as_pd(det), _lib.as_pi64(ii), pos.shape[0], _lib.as_pd(pos), _lib.as_pd(rot),
                                    _lib.as_pd(k4), _lib.as_pd(xyz), _lib.as_pd(proj), mask.ctypes.data), lib)
    return xyz, proj, mask.view(np.bool_)


def prepare_window(detections, orbit_np, intrinsics=None, dt=1.0, device=None) -> Window:
    """Reference od_pipe.py:924-961 (read, ground truth, outlier mask, renumber, IMU).

    ``device``: None = everything on the host in NumPy, bit-identical to the arrays the reference's own preparation produces
    (``tests/test_od_pipe_host.py``); a GPU index = the per-row part (lat / lon -> ECI, reprojection at ground truth, outlier
    mask: most of the time of a 50 000-row sequence) by one kernel of the library, equal to rounding."""
    orbit, f, intr, time_idx, ii = read_detections(detections, np.array(orbit_np, dtype=np.float64), intrinsics)
    # process_ground_truths (od_pipe.py:94-123)
    pos_full = orbit[:, :3]
    pos_gt = pos_full[time_idx]
    vel_full = frames.finite_difference(pos_full, dt)
    quat_full = frames.nadir_quaternion(pos_full)
    quat_gt = quat_full[time_idx]           # (the attitude of a position is a function of that row alone: the rows of one evaluation)
    uv = np.asarray(f["uv"], dtype=np.float64)
    conf = np.asarray(f["confidence"], dtype=np.float64)
    intr_rows = np.repeat(intr[None], len(pos_gt), axis=0)
    # outlier mask from the reprojection at ground truth (od_pipe.py:928-930): the rotation of a pose is formed once per pose and
    # gathered per row (the same operations per element as forming it per row -- ~100 rows share a pose)
    R = quat_to_matrix(quat_gt / np.linalg.norm(quat_gt, axis=-1, keepdims=True))
    if device is not None:
        xyz, proj, mask = rows_on_device(detections, ii, pos_gt, R, intr, device)
    else:
        xyz = frames.latlon_to_eci(f["lonlat"][:, 1], f["lonlat"][:, 0], f["frame"])
        pc = np.einsum("kji,kj->ki", R[ii], xyz - pos_gt[ii])
        z = np.maximum(pc[:, 2], 0.1)
        proj = np.empty((len(ii), 2))
        proj[:, 0] = intr[0] * pc[:, 0] / z + intr[2]
        proj[:, 1] = intr[1] * pc[:, 1] / z + intr[3]
        mask = ((proj[:, 0] > 0) & (proj[:, 1] > 0) & (proj[:, 0] < 4700) & (proj[:, 1] < 2600)
                & (np.linalg.norm(proj - uv, axis=-1) < 1000) & (conf > 0.8))
    # remove_elems (od_pipe.py:253-288): keep poses that still own an observation, and knots
    all_kept = bool(mask.all())
    ii_kept = ii if all_kept else ii[mask]
    keep = np.zeros(time_idx.shape[0], dtype=bool)
    keep[ii_kept] = True                    # (every pose that still owns a row)
    keep |= (time_idx % KNOT_PERIOD == 0)
    if keep.all():
        ii_new = ii_kept
    else:
        new_index = np.cumsum(keep) - 1
        ii_new = new_index[ii_kept]
        time_idx = time_idx[keep]
        pos_gt, quat_gt = pos_gt[keep], quat_gt[keep]
    # IMU pre-integration (od_pipe.py:945-961): only the rotation accumulated over each gap is
    # consumed downstream (BA_utils.py:295), zero-rate padding being the identity.
    T = len(pos_gt)
    gaps = np.diff(time_idx)
    max_gap = int(gaps.max()) if T > 1 else 1
    # The per-second increments exp(dt * omega) are formed for every second at once; their product over each gap is a serial
    # recurrence per pose (up to 1000 factors) and runs in the library's host code (vba_host_gap_rotations: every product and sum
    # rounded as the array expression rounds it -- the bits of the reference's cum_rots[:, :, -1]).
    omega = quat.omega_from_quats(quat_full, dt)
    cum = gap_rotations(quat.qexp(dt * omega), time_idx)
    sel = (lambda a: np.ascontiguousarray(a)) if all_kept else (lambda a: a[mask])
    return Window(time_idx=time_idx, ii=ii_new, landmarks_uv=sel(uv), landmarks_xyz=sel(xyz),
                  confidences=sel(conf), intrinsics=intr_rows, poses_gt=np.concatenate([pos_gt, quat_gt], 1),
                  vel_gt_full=vel_full, quat_gt_full=quat_full, omega_gt=omega, cumrot_last=cum,
                  max_gap=max_gap, mask=mask, extras=dict(proj_gt=sel(proj)))


def _lib_host():
    from . import _lib
    return _lib, _lib.load()


def gap_rotations(rot, time_idx):
    """cum[i] = rot[t_i] (x) rot[t_i + 1] (x) ... (x) rot[t_{i+1} - 1], identity for the last pose (reference
    ``precompute_cum_rotations`` BA_utils.py:278-288 as the driver uses it, od_pipe.py:945-961; only ``[..., -1]`` is read)."""
    _lib, lib = _lib_host()
    rot = np.ascontiguousarray(rot, dtype=np.float64)
    t = np.ascontiguousarray(time_idx, dtype=np.int64)
    cum = np.empty((t.size, 4))
    _lib.check(lib.vba_host_gap_rotations(_lib.as_pd(rot), rot.shape[0], _lib.as_pi64(t), t.size, _lib.as_pd(cum)), lib)
    return cum


def initial_guess(win: Window, seed=0):
    """Ground truth + Gaussian perturbation (100 km, 0.2 rad, 10 % |v|), od_pipe.py:962-969.

    Uses torch's CPU generator so the draw is the one the reference makes after
    ``torch.manual_seed(seed)``.
    """
    import torch
    # a generator of its own, seeded like the reference seeds the global one (od_pipe.py:913): the same draws, and sequences can
    # be prepared on several host threads at once (prepared_runs)
    gen = torch.Generator()
    gen.manual_seed(seed)
    T = win.poses_gt.shape[0]
    vel = win.velocities
    pos_off = (torch.randn((T, 3), generator=gen) * 100).double().numpy()
    ori_off = (torch.randn([T, 3], generator=gen) * 0.2).double().numpy()
    vmean = torch.tensor(vel).abs().mean()
    vel_off = (torch.randn([T, 3], generator=gen) * vmean * 0.1).numpy()
    position = win.poses_gt[:, :3] + pos_off
    orientation = quat.qexp(quat.qlog(win.poses_gt[:, 3:]) + ori_off)
    return np.concatenate([position, orientation, vel + vel_off], -1)


def next_batch(ii, time_idx, i):
    """Cut where the frame gap exceeds 200 s after >4 near-contiguous observations.

    Reference ``identify_next_batch_new`` od_pipe.py:898-905.  Returns
    ``(t_final, i_final, seq_end)``.
    """
    gaps = np.diff(time_idx[ii[i:]])                    # gaps[k] = t_obs[i + 1 + k] - t_obs[i + k]
    contiguous = np.cumsum(gaps < 100)                  # (a gap cannot be both < 100 and > 200: counting first is the loop's order)
    cut = np.flatnonzero((gaps > 200) & (contiguous > 4))
    if cut.size:
        j = i + 1 + int(cut[0])
        return int(ii[j - 1]) + 1, j, False
    return int(ii[-1]) + 1, len(ii), True


def propagate_between_batches(state, velocity, omega, tdiff, duration, rk4_step=None):
    """Dead-reckon the last estimate across a gap (reference ``propagate_dynamics_init``
    BA_utils.py:114-129): ``tdiff`` steps to reach the first new frame, then ``duration``
    more, returning the per-second states from the first new frame on, [duration+1, 10].

    As in the reference the orbit is started from the estimated POSITION but from the ``velocities``
    tensor the driver carries beside the states (od_pipe.py:1011), which ``BA`` hands back untouched
    (BA_filtering.py:98) -- not from the velocity inside the state vector.
    """
    K = tdiff + duration
    if rk4_step is not None:        # the interpreted loop (kept for the comparison in tests/test_od_pipe_host.py)
        x = np.concatenate([state[:3], velocity])
        q = state[3:7].copy()
        out = []
        for k in range(K):
            x = rk4_step(x)
            q = quat.qmul(q, quat.qexp(1.0 * omega[k]))
            if k >= tdiff - 1:
                out.append(np.concatenate([x[:3], q, x[3:]]))
        return np.stack(out)
    # both chains are serial recurrences of up to ~1000 steps: in the library's host code (vba_host_orbit_chain: the RK4 of the
    # device path's host build; vba_host_quat_chain: the bits of the array expression)
    _lib, lib = _lib_host()
    x0 = np.ascontiguousarray(np.concatenate([state[:3], velocity]), dtype=np.float64)
    xs = np.empty((K, 6))
    _lib.check(lib.vba_host_orbit_chain(_lib.as_pd(x0), K, _lib.as_pd(xs)), lib)
    rot = np.ascontiguousarray(quat.qexp(1.0 * np.asarray(omega[:K], dtype=np.float64)))
    q0 = np.ascontiguousarray(state[3:7], dtype=np.float64)
    qs = np.empty((K, 4))
    _lib.check(lib.vba_host_quat_chain(_lib.as_pd(q0), _lib.as_pd(rot), K, _lib.as_pd(qs)), lib)
    out = np.empty((duration + 1, 10))
    out[:, :3], out[:, 3:7], out[:, 7:] = xs[tdiff - 1:, :3], qs[tdiff - 1:], xs[tdiff - 1:, 3:]
    return out


class SequenceRun:
    """One sequence of the reference's ``streaming_version`` (od_pipe.py:911-1062), cut at its ``BA`` loops: ``next_patch``
    does what the driver does in front of the 20 calls of a batch (batch cut, dead reckoning across the gap, error
    bookkeeping) and returns their arguments, ``finish_patch`` takes their result.  The sequential driver and the batched
    one (:func:`streaming_batched`: the patches of many sequences as windows of ONE handle) share this code."""

    def __init__(self, detections, orbit_np, device=None):
        import torch
        win = self.win = prepare_window(detections, orbit_np, device=device)
        states = initial_guess(win, seed=0)
        self.time_idx, self.ii = win.time_idx, win.ii
        T = self.T = len(self.time_idx)
        # the reference hands BA an imu tensor [1,T,N,10] of which only [..., -1, 6:10] is read
        self.imu = torch.zeros((1, T, 1, 10), dtype=torch.float64)
        self.imu[0, :, 0, 6:10] = torch.from_numpy(win.cumrot_last)
        self.uv = torch.from_numpy(win.landmarks_uv)[None]
        self.xyz = torch.from_numpy(win.landmarks_xyz)[None]
        self.intr = torch.from_numpy(win.intrinsics)[None]
        self.conf = torch.from_numpy(win.confidences)
        self.poses_gt = torch.from_numpy(win.poses_gt)
        self.vel_all = torch.from_numpy(win.velocities)[None]
        self.states_all = torch.from_numpy(states)[None]
        self.t = self.i = 0
        self.seq_end = False
        self.patch = 0
        self.errors, self.times = [], []
        self.first_detection = None
        self.states_t = self.vel_t = None

    def next_patch(self):
        """Arguments of the next batch's BA calls, or None when the sequence has ended."""
        import torch
        if self.seq_end:
            return None
        win, time_idx, ii = self.win, self.time_idx, self.ii
        t_init = self.t
        self.t, self.i, self.seq_end = next_batch(ii, time_idx, self.i)
        t, i = self.t, self.i
        if self.patch == 0:
            self.states_t = self.states_all[:, :t]
            self.vel_t = self.vel_all[:, :t]
            self.first_detection = time_idx[:t][-1]
        else:
            omega = win.omega_gt[time_idx[t_init - 1]:time_idx[t - 1]]
            tdiff = int(time_idx[t_init] - time_idx[t_init - 1])
            duration = int(time_idx[t - 1] - time_idx[t_init])
            prop = propagate_between_batches(self.states_t[0, -1].numpy(), self.vel_t[0, -1].numpy(), omega, tdiff, duration)
            sel = time_idx[t_init:t] - time_idx[t_init]
            prop = torch.from_numpy(prop[sel])[None]
            self.states_t = torch.cat([self.states_t, prop], dim=1)
            self.vel_t = torch.cat([self.vel_t, prop[..., 7:]], dim=1)
            err_prop = (prop[0, :, :3] - self.poses_gt[t - prop.shape[1]:t, :3]).norm(dim=-1)[:-1]
            self.times.append(time_idx[t - prop.shape[1]:t][:-1])
            self.errors.append(err_prop)
        return dict(first=self.patch == 0, states=self.states_t, velocities=self.vel_t, imu=self.imu[:, :t], uv=self.uv[:, :i],
                    xyz=self.xyz[:, :i], ii=ii[:i], time_idx=time_idx[:t], intr=self.intr[:, :t], conf=self.conf[:i],
                    poses_gt=self.poses_gt[:t], lam=1e-4)

    def finish_patch(self, states_t, vel_t):
        """The estimate after the batch's BA calls: bookkeeping behind the loop (od_pipe.py:1041-1060)."""
        import torch
        win, time_idx, t, T = self.win, self.time_idx, self.t, self.T
        self.states_t, self.vel_t = states_t, vel_t
        self.patch += 1
        self.errors.append((states_t[0, -1:, :3] - self.poses_gt[t - 1:t, :3]).norm(dim=-1))
        self.times.append(time_idx[t - 1:t])
        if self.seq_end and t < T:
            t_init, t = t, T
            omega = win.omega_gt[time_idx[t_init - 1]:time_idx[t - 1]]
            tdiff = int(time_idx[t_init] - time_idx[t_init - 1])
            duration = int(time_idx[t - 1] - time_idx[t_init])
            prop = propagate_between_batches(states_t[0, -1].numpy(), vel_t[0, -1].numpy(), omega, tdiff, duration)
            prop = torch.from_numpy(prop[time_idx[t_init:t] - time_idx[t_init]])
            self.errors.append((prop[:, :3] - self.poses_gt[t_init:t, :3]).norm(dim=-1))
            self.times.append(time_idx[-prop.shape[0]:])

    def result(self):
        import torch
        return torch.cat(self.errors), self.first_detection, self.times


def prepared_runs(sources, threads=None, ahead=None, device=None):
    """``SequenceRun`` objects for ``sources`` -- ``(detections, orbit_np)`` array pairs or ``(detections_file, orbit_file)`` path
    pairs -- in order, prepared on ``threads`` host threads (default: three) up to ``ahead`` sequences in
    front of the consumer.  Data preparation is ~9 ms of NumPy per 50 000-row sequence against ~1 ms of BA calls: the array
    kernels and the library's host helpers release the interpreter lock, so the preparation of the next sequences runs beside
    the consumer's BA calls and beside each other."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    sources = list(sources)
    if threads is None:
        threads = max(1, min(3, len(os.sched_getaffinity(0)), len(sources)))     # (measured: 1.5 x at three threads, nothing beyond -- the interpreter lock)
    if ahead is None:
        ahead = 2 * threads

    def make(src):
        det, orb = src
        if isinstance(det, (str, bytes, os.PathLike)):
            det, orb = np.load(det, allow_pickle=True), np.load(orb, allow_pickle=True)
        return SequenceRun(det, orb, device=device)

    if threads <= 1 or len(sources) <= 1:
        for src in sources:
            yield make(src)
        return
    with ThreadPoolExecutor(max_workers=threads) as pool:
        pending = []
        nxt = 0
        while nxt < len(sources) or pending:
            while nxt < len(sources) and len(pending) < ahead:
                pending.append(pool.submit(make, sources[nxt]))
                nxt += 1
            yield pending.pop(0).result()


def _load_sequence(detections, orbit_np, orbit_file_name, detections_file_name):
    if detections is None:
        detections = np.load(detections_file_name, allow_pickle=True)
    if orbit_np is None:
        orbit_np = np.load(orbit_file_name, allow_pickle=True)
    return detections, orbit_np


class _Clock:
    """Wall time by phase of a driver run (``timing`` argument of the drivers): seconds added to ``prep`` (data preparation of a
    sequence: read, ground truth, mask, IMU, initial guess), ``ba`` (inside the BA calls, uploads and result copies included) and
    ``bookkeeping`` (batch cut, dead reckoning across the gap, error records), plus the number of ``ba_calls``."""

    def __init__(self, timing):
        import time
        self.t, self.now = timing, time.perf_counter
        if timing is not None:
            for k in ("prep", "ba", "bookkeeping", "ba_calls"):
                timing.setdefault(k, 0)

    def __call__(self, key, t0, calls=0):
        if self.t is not None:
            self.t[key] += self.now() - t0
            self.t["ba_calls"] += calls
        return self.now()


def streaming_version(detections=None, orbit_np=None, orbit_file_name=None, detections_file_name=None,
                      ba=None, num_iters=NUM_ITERS, record=None, timing=None, run=None, device=0):
    """Drop-in for the reference's ``streaming_version`` (od_pipe.py:911-1062).

    ``ba`` defaults to the HIP-backed :func:`vinsat_amd.ba.BA`; tests may inject another
    callable with the reference signature.  With the default ``ba`` and no ``record`` list the ``num_iters`` calls
    of a batch are issued as one chained device call (:func:`vinsat_amd.ba.BA_window`, same bits).
    ``timing`` (a dict) receives the wall time by phase (:class:`_Clock`).  ``run``: a :class:`SequenceRun` prepared elsewhere
    (:func:`prepared_runs`) instead of the input arrays / files.  With the default ``ba`` the per-row part of the data preparation
    runs on GPU ``device`` as well (:func:`prepare_window`); an injected ``ba`` keeps the preparation on the host.
    """
    ba_window = None
    rows_device = None
    if ba is None:
        rows_device = device
        from .ba import BA as ba
        if record is None:
            from .ba import BA_window as ba_window
    clk = _Clock(timing)
    t0 = clk.now()
    if run is None:
        np.random.seed(0)           # (as the reference's driver does, od_pipe.py:913; nothing here draws from it)
        run = SequenceRun(*_load_sequence(detections, orbit_np, orbit_file_name, detections_file_name), device=rows_device)
    t0 = clk("prep", t0)
    while True:
        p = run.next_patch()
        t0 = clk("bookkeeping", t0)
        if p is None:
            break
        states_t, vel_t, lam = p["states"], p["velocities"], p["lam"]
        inits = [(it < 10) if p["first"] else False for it in range(num_iters)]
        if ba_window is not None:
            states_t, vel_t, lam, last_h = ba_window(range(num_iters), inits, states_t, vel_t, p["imu"], p["uv"], p["xyz"],
                                                     p["ii"], p["time_idx"], p["intr"], p["conf"], lam)
        for it in range(num_iters if ba_window is None else 0):
            states_t, vel_t, lam, last_h = ba(it, states_t, vel_t, p["imu"], p["uv"], p["xyz"], p["ii"],
                                              p["time_idx"], p["intr"], p["conf"], 1e-3, 1e-3, lam,
                                              p["poses_gt"], initialize=inits[it])
            if record is not None:
                record.append(dict(patch=run.patch, iter=it, states=states_t.clone(), lamda=lam))
        t0 = clk("ba", t0, num_iters)
        run.finish_patch(states_t, vel_t)
    out = run.result()
    clk("bookkeeping", t0)
    return out


def streaming_batched(sequences, num_iters=NUM_ITERS, ba_window=None, record=None, timing=None, threads=None, device=0):
    """Many sequences at once -- the reference's outer loop over sequence files (od_pipe.py:1069-1077) turned into the batch
    dimension of ``BA``: round r runs batch r of EVERY sequence that still has one as the windows of ONE ragged handle
    (:func:`vinsat_amd.ba.BA_window` on lists: every kernel launch covers all of them), sequences that have ended drop out.

    ``sequences``: list of ``(detections, orbit_np)`` pairs.  Returns the list of ``streaming_version`` results.  With equal
    handle settings (``vinsat_amd.ba.configure``) every sequence gets the bits of its own ``streaming_version`` run.
    ``record`` (a list) receives ``dict(round, sequence, states, lamda)`` after every round; ``timing`` (a dict) the wall time
    by phase (:class:`_Clock`).
    """
    rows_device = None
    if ba_window is None:
        rows_device = device            # (the HIP BA is in use: the per-row preparation runs on its device as well)
        from .ba import BA_window as ba_window
    clk = _Clock(timing)
    t0 = clk.now()
    runs = list(prepared_runs(sequences, threads=threads, ahead=len(sequences), device=rows_device))     # (on several host threads)
    t0 = clk("prep", t0)
    rnd = 0
    while True:
        live = [(k, r, r.next_patch()) for k, r in enumerate(runs)]
        live = [(k, r, p) for k, r, p in live if p is not None]
        t0 = clk("bookkeeping", t0)
        if not live:
            break
        # batch 0 of a sequence is the only one with landmark-only calls (od_pipe.py:1038): all sequences are in the same
        # round, so `initialize` is one value per call as in the reference
        first = live[0][2]["first"]
        assert all(p["first"] == first for _, _, p in live)
        inits = [(it < 10) if first else False for it in range(num_iters)]
        ps = [p for _, _, p in live]
        if len(ps) == 1:        # a single window left: the one-window path (pipelined, latency mode)
            p = ps[0]
            st, vel, lam, _ = ba_window(range(num_iters), inits, p["states"], p["velocities"], p["imu"], p["uv"], p["xyz"], p["ii"],
                                        p["time_idx"], p["intr"], p["conf"], p["lam"])
            st, lam = [st], [lam]
        else:
            st, _, lam, _ = ba_window(range(num_iters), inits, [p["states"] for p in ps], [p["velocities"] for p in ps],
                                      [p["imu"] for p in ps], [p["uv"] for p in ps], [p["xyz"] for p in ps], [p["ii"] for p in ps],
                                      [p["time_idx"] for p in ps], [p["intr"] for p in ps], [p["conf"] for p in ps],
                                      [p["lam"] for p in ps])
        t0 = clk("ba", t0, num_iters * len(ps))
        for (k, r, p), s_new, l_new in zip(live, st, lam):
            if record is not None:
                record.append(dict(round=rnd, sequence=k, states=s_new.clone(), lamda=l_new))
            r.finish_patch(s_new, p["velocities"])
        rnd += 1
    out = [r.result() for r in runs]
    clk("bookkeeping", t0)
    return out
