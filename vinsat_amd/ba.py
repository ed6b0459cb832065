"""``BA``: the reference's bundle-adjustment call surface, executed by the HIP kernels.

Same positional signature and 4-tuple return as the reference's
``estimation/BA/BA_filtering.py:4, 98``::

    states_new, velocities, lamda_init, last_hessian = BA(iter, states, velocities, imu_meas, landmarks,
        landmarks_xyz, ii, time_idx, intrinsics, confidences, Sigma, V, lamda_init, poses_gt_eci, initialize=False)

Behaviour kept from the reference: ``Sigma``/``V`` are ignored (overwritten at :26-27),
``velocities`` is returned untouched, only ``imu_meas[0, :, -1, 6:10]`` is read
(``BA_utils.py:295``), a "lamda too large" outcome keeps the last trial (:75-77).
Inputs may be torch tensors or NumPy arrays; outputs are fp64 torch tensors shaped like
the reference's (``[1,n,10]``, ``[1,9,9]``).

The observation arrays are constant over the 20 calls of a window, so the engine is cached
and re-uploaded only when they change (cheap fingerprint).  No CPU fallback exists.
"""
from __future__ import annotations

import numpy as np

from .engine import BAEngine

_cache = {}


def _np(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().double().numpy()
    except ImportError:
        pass
    return np.asarray(x, dtype=np.float64)


def _fingerprint(*arrays):
    parts = []
    for a in arrays:
        parts.append((a.shape, a.dtype.str, a.ctypes.data, float(a.reshape(-1)[:: max(1, a.size // 64)].sum())))
    return tuple(parts)


def _engine_for(xyz, uv, conf, ii, K, cum, t, device):
    n, m = K.shape[0], xyz.shape[0]
    eng = _cache.get("eng")
    if eng is None or eng.n_max < n or eng.m_max < m or eng.device != device:
        if eng is not None:
            eng.close()
        eng = BAEngine(max(n, 16), max(m, 256), windows=1, device=device)
        eng.n_max, eng.m_max, eng.device = max(n, 16), max(m, 256), device
        _cache["eng"] = eng
        _cache["fp"] = None
    fp = (n, m, ii.tobytes() if m <= 4096 else (int(ii.sum()), int(ii[0]), int(ii[-1])),
          t.tobytes(), float(xyz.sum()), float(uv.sum()), float(conf.sum()), float(K.sum()), float(cum.sum()))
    if _cache.get("fp") != fp:
        eng.upload_observations(xyz, uv, conf, ii, n)
        eng.upload_window(K, cum, t)
        _cache["fp"] = fp
    return eng


def BA(iter, states, velocities, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences, Sigma, V,
       lamda_init, poses_gt_eci, initialize=False, device=0):
    import torch
    st = _np(states)
    if st.ndim != 3 or st.shape[0] != 1 or st.shape[2] != 10:
        raise ValueError("states must be [1, n, 10] (the reference hard-codes batch index 0, BA_filtering.py:24,37)")
    n = st.shape[1]
    imu = _np(imu_meas)
    cum = np.ascontiguousarray(imu[0, :, -1, 6:10])
    uv = _np(landmarks).reshape(-1, 2)
    xyz = _np(landmarks_xyz).reshape(-1, 3)
    K = _np(intrinsics).reshape(-1, 4)
    conf = _np(confidences).reshape(-1)
    ii = np.ascontiguousarray(np.asarray(ii), dtype=np.int64).reshape(-1)
    t = np.ascontiguousarray(np.asarray(time_idx), dtype=np.int64).reshape(-1)
    if not (K.shape[0] == n and cum.shape[0] == n and t.shape[0] == n):
        raise ValueError("intrinsics / imu_meas / time_idx must have one row per pose")
    eng = _engine_for(xyz, uv, conf, ii, K, cum, t, device)
    out, lam, hess, n_trials, flags = eng.iterate(int(iter), bool(initialize), float(lamda_init), st[0])
    if flags & 1:
        print("lamda too large")          # reference BA_filtering.py:76
    BA.last = dict(n_trials=n_trials, flags=flags, ms=eng.last_step_ms())
    return (torch.from_numpy(out)[None], velocities, lam, torch.from_numpy(hess)[None])


BA.last = {}


def BA_reg(iter, states, velocities, states_prior, velocity_prior, hessian_state_t, hessian_rot_t, imu_meas, landmarks,
           landmarks_xyz, ii, time_idx, intrinsics, confidences, Sigma, V, lamda_init, poses_gt_eci, initialize=False,
           use_reg=True, device=0):
    """Counterpart of the reference's ``BA_reg`` (``BA_filtering.py:100-210``): ``BA`` with a propagated-covariance
    prior per pose.  Same positional signature and 4-tuple.  As in the reference ``Sigma``, ``V``, ``use_reg`` and
    ``velocity_prior`` are not read (the prior velocity is ``states_prior[..., 7:]``, ``BA_utils.py:614``), and
    ``hessian_rot_t`` has no effect on the result (see ``include/vinsat_ba.h``: the rotation term of ``prior_gpu`` is
    a constant).  The integrator is the reference's CPU branch (``predict``) unless the engine is switched."""
    import torch
    st = _np(states)
    if st.ndim != 3 or st.shape[0] != 1 or st.shape[2] != 10:
        raise ValueError("states must be [1, n, 10]")
    n = st.shape[1]
    cum = np.ascontiguousarray(_np(imu_meas)[0, :, -1, 6:10])
    uv = _np(landmarks).reshape(-1, 2)
    xyz = _np(landmarks_xyz).reshape(-1, 3)
    K = _np(intrinsics).reshape(-1, 4)
    conf = _np(confidences).reshape(-1)
    ii = np.ascontiguousarray(np.asarray(ii), dtype=np.int64).reshape(-1)
    t = np.ascontiguousarray(np.asarray(time_idx), dtype=np.int64).reshape(-1)
    sp = _np(states_prior).reshape(-1, 10)
    Hs = _np(hessian_state_t).reshape(-1, 6, 6)
    if not (K.shape[0] == n and cum.shape[0] == n and t.shape[0] == n and sp.shape[0] == n and Hs.shape[0] == n):
        raise ValueError("intrinsics / imu_meas / time_idx / prior must have one row per pose")
    eng = _engine_for(xyz, uv, conf, ii, K, cum, t, device)
    eng.upload_prior(sp, Hs)
    eng.set_prior(True)
    try:
        out, lam, hess, n_trials, flags = eng.iterate(int(iter), bool(initialize), float(lamda_init), st[0])
    finally:
        eng.set_prior(False)
    if flags & 1:
        print("lamda too large")          # reference BA_filtering.py:186
    BA_reg.last = dict(n_trials=n_trials, flags=flags, ms=eng.last_step_ms())
    return (torch.from_numpy(out)[None], velocities, lam, torch.from_numpy(hess)[None])


BA_reg.last = {}


def BA_window(iters, initializes, states, velocities, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences,
              lamda_init, device=0):
    """The driver's loop ``for iter in range(num_iters): states, ... = BA(iter, states, ...)`` (reference
    ``od_pipe.py:1036-1040``) as ONE call: the states stay on the device between the calls and the calls are chained
    there (``vba_run_schedule``).  Bit-identical to calling :func:`BA` ``len(iters)`` times.

    Returns ``(states_new, velocities, lamda, last_hessian)`` of the last call, shaped like ``BA``'s.
    """
    import torch
    st = _np(states)
    if st.ndim != 3 or st.shape[0] != 1 or st.shape[2] != 10:
        raise ValueError("states must be [1, n, 10]")
    n = st.shape[1]
    cum = np.ascontiguousarray(_np(imu_meas)[0, :, -1, 6:10])
    uv = _np(landmarks).reshape(-1, 2)
    xyz = _np(landmarks_xyz).reshape(-1, 3)
    K = _np(intrinsics).reshape(-1, 4)
    conf = _np(confidences).reshape(-1)
    ii = np.ascontiguousarray(np.asarray(ii), dtype=np.int64).reshape(-1)
    t = np.ascontiguousarray(np.asarray(time_idx), dtype=np.int64).reshape(-1)
    eng = _engine_for(xyz, uv, conf, ii, K, cum, t, device)
    eng.set_states(st[0], float(lamda_init))
    eng.run_schedule(list(iters), list(initializes))
    out, lam, hess, n_trials, flags = eng.get_states()
    if flags & 1:
        print("lamda too large")
    return (torch.from_numpy(out)[None], velocities, lam, torch.from_numpy(hess)[None])
