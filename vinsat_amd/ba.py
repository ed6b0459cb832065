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

What makes the reference's loop ``for iter in range(20): states, ... = BA(iter, states, ...)``
(``od_pipe.py:1036-1040``) cheap here:

* the window arguments (observations, per-pose constants) are uploaded once: a call whose arguments are the SAME
  buffers with the SAME content as the previous call's skips the upload.  For torch tensors "same" is decided by identity:
  buffer address, shape, strides, dtype and torch's in-place modification counter ``_version`` (a write through a
  ``.numpy()`` alias or ``.data`` does not bump that counter: call :func:`invalidate` after such a write).  NumPy arrays
  carry no such counter, so their CONTENT is compared, byte for byte, with a private copy taken at the upload (``ii`` and
  ``time_idx`` are what the reference passes as ndarrays: 0.4 MB at 500 / 50k, ~8 us of memcmp) -- an in-place edit is
  seen and uploads the window again.  The arguments of the last upload are kept referenced, so an address cannot be
  recycled by another live array;
* a call whose ``states`` IS the tensor the previous call returned (and whose ``lamda_init`` is the value it
  returned) uploads nothing at all: the device already holds both (``vba_iterate_resident``).

Batch dimension.  The reference's ``BA`` carries ``bsz = states.shape[0]`` (``BA_filtering.py:14``) but hard-codes batch index
0 at ``:24`` and ``:37``, and its real batch axis is the loop over sequences (``od_pipe.py:1069-1077``).  Here ``BA`` /
``BA_window`` take a batch in either of two forms and run all its windows with every kernel launch covering the lot (one
ragged handle of ``bsz`` windows, ``vba_step`` / ``vba_run_schedule``):

* dense: ``states [B, n, 10]``, ``imu_meas [B, n, g, 10]``, ``landmarks [B, m, 2]``, ``landmarks_xyz [B, m, 3]``,
  ``intrinsics [B, n, 4]`` (or ``[1, n, 4]``), ``ii`` / ``time_idx`` / ``confidences`` shared (``[m]``, ``[n]``, ``[m]``) or per
  window (``[B, m]`` ...), ``lamda_init`` a float or ``B`` floats; returns ``states_new [B, n, 10]``, ``velocities``, a list of
  ``B`` dampings and ``last_hessian [B, 9, 9]``;
* ragged: every per-window argument a list of ``B`` items shaped as the reference's single-window arguments (windows of
  different ``n``, ``m``); returns lists.

``iter`` and ``initialize`` are per call, as in the reference.  Equal handle settings (:func:`configure`) give a window the
same bits in a batch as alone.

No CPU fallback exists.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .engine import BAEngine

_memcmp = ctypes.CDLL(None).memcmp
_memcmp.argtypes = (ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
_memcmp.restype = ctypes.c_int

_cache = {}
_WATCH_SLOTS = 8         # vba_set_host_watch: one per array argument of BA()


def _np(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().double().numpy()
    except ImportError:
        pass
    return np.asarray(x, dtype=np.float64)


def _token(x):
    """Identity of an argument's buffer (torch: plus its in-place counter); the content of NumPy arrays is checked
    separately (host watch, see :func:`_engine_for`), see the module docstring."""
    ver = getattr(x, "_version", None)
    if ver is not None:                                    # torch.Tensor
        return (x.data_ptr(), tuple(x.shape), x.stride(), x.dtype, ver)
    if isinstance(x, np.ndarray):
        return (x.__array_interface__["data"][0], x.shape, x.strides, x.dtype.str)
    return ("obj", id(x))


def _same_objects(args):
    """The cheap test that almost every call of a driver loop passes: the very same Python objects as at the upload, torch
    tensors with an unchanged in-place counter (ndarrays: their content is compared by the library while the device
    works, see ``vba_set_host_watch``)."""
    refs = _cache.get("refs")
    if refs is None:
        return False
    vers = _cache["vers"]
    for a, r, v in zip(args, refs, vers):
        if a is not r or getattr(a, "_version", None) != v:
            return False
    return True


def _same_bytes(a, c):
    """Equal content of two ndarrays of equal shape and dtype; contiguous ones by one ``memcmp`` (``np.array_equal`` builds a
    boolean array first: ~5 x the time at 400 kB)."""
    if a.flags.c_contiguous and c.flags.c_contiguous and a.dtype.kind in "iuf":
        if a.dtype.kind != "f":
            return _memcmp(a.ctypes.data, c.ctypes.data, a.nbytes) == 0
        # (floats: equal bytes are equal values; different bytes may still be equal values, -0.0 == 0.0 -- only then look closer)
        return _memcmp(a.ctypes.data, c.ctypes.data, a.nbytes) == 0 or np.array_equal(a, c)
    return np.array_equal(a, c)


def _watchable(a):
    """The host buffer of an argument whose CONTENT is compared with the uploaded copy: an ndarray itself; with
    ``configure(strict=True)`` also a torch CPU tensor (as the ndarray view of its storage: a write through ``.numpy()`` or any
    other alias does not bump ``_version`` and would otherwise go unseen); None for everything else."""
    if isinstance(a, np.ndarray):
        return a
    if _cache.get("strict") and getattr(a, "_version", None) is not None:
        try:
            if a.device.type == "cpu" and not a.requires_grad:
                return a.detach().numpy()
        except (RuntimeError, TypeError):
            pass
    return None


def _numpy_unchanged(args):
    """True if every watchable argument (:func:`_watchable`) still holds the bytes of the private copy taken when it was uploaded."""
    for a, c in zip(args, _cache.get("np_copies", ())):
        if c is None:
            continue
        a = _watchable(a)
        if a is None or not (a.shape == c.shape and a.dtype == c.dtype and _same_bytes(a, c)):
            return False
    return True


def configure(integrator=None, lanes=None, fusion=None, solver=None, mode=None, strict=None):
    """Settings of the cached engines that the reference's call surface has no argument for.

    ``integrator``: ``"rk4"`` (default) = one-second RK4 steps, the reference's CPU branch ``predict``
    (``BA_utils.py:73-87``) -- the parity target; ``"hop"`` = the <=100 s hops of ``predict_gpu``
    (``BA_utils.py:52-71, 529-602``), which is what the reference itself runs when it sees a GPU
    (``BA_filtering.py:16-17``).  Takes effect from the next call on.

    ``lanes`` / ``fusion`` / ``solver`` / ``mode`` pin what a handle otherwise chooses from its own geometry -- lanes per
    pose of the accumulation (``VBA_OPT_ACCUMULATE_LANES``: the shape of its reduction tree), the kernel-fusion mask
    (``VBA_OPT_FUSION``), the chain partition ``(chunk, chunk2)`` (``vba_set_solver2``; ``0`` = sequential walk) and the kernel
    set (``vba_create_mode``) -- so that a window gets the same bits alone and in a batch of any size.  ``"auto"`` returns
    one of them to the handle's choice.  The cached engines are rebuilt.

    ``strict=True``: torch CPU arguments are content-checked like ndarrays (private copy at the upload, compared by the library
    while the device works) instead of being trusted on address / shape / ``_version`` -- a write through ``tensor.numpy()`` or
    another alias is then seen by itself; costs one more copy of the window's arguments at every upload."""
    if strict is not None and bool(strict) != bool(_cache.get("strict")):
        _cache["strict"] = bool(strict)
        invalidate()
    if integrator is not None:
        if integrator not in ("rk4", "hop"):
            raise ValueError("integrator must be 'rk4' or 'hop'")
        _cache["hop"] = integrator == "hop"
        for key in ("eng", "beng"):
            eng = _cache.get(key)
            if eng is not None:
                eng.set_integrator(_cache["hop"])
        _cache["resident"] = None
        _cache["bresident"] = None
    pins = dict(_cache.get("pins", {}))
    changed = False
    for name, val in (("lanes", lanes), ("fusion", fusion), ("solver", solver), ("mode", mode)):
        if val is None:
            continue
        if val == "auto":
            changed |= pins.pop(name, None) is not None
        else:
            changed |= pins.get(name) != val
            pins[name] = val
    if changed:
        _cache["pins"] = pins
        release()


def _new_engine(n_max, m_max, windows, device):
    """An engine with the pinned settings (:func:`configure`) applied."""
    pins = _cache.get("pins", {})
    eng = BAEngine(n_max, m_max, windows=windows, device=device, mode={"lat": 1, "bw": 0}.get(pins.get("mode"), pins.get("mode", -1)))
    eng.set_integrator(bool(_cache.get("hop", False)))
    if "lanes" in pins:
        eng.set_accumulate_lanes(pins["lanes"])
    if "fusion" in pins:
        eng.set_fusion(pins["fusion"])
    if "solver" in pins:
        sv = pins["solver"]
        if isinstance(sv, (tuple, list)):
            eng.set_solver(*sv)
        else:
            eng.set_solver(sv)
    return eng


def invalidate():
    """Forget what is on the device: the next call uploads its window again (needed after a torch argument was written
    through an alias that does not bump its ``_version``; edits of NumPy arguments are seen by themselves)."""
    # the watch slots first: clearing them waits for the library's comparison helper, which may still be reading the live
    # buffers and the copies that are dropped below
    eng = _cache.get("eng")
    if eng is not None and getattr(eng, "h", None):
        for k in range(_WATCH_SLOTS):
            eng.set_host_watch(k)
    _cache.pop("key", None)
    _cache.pop("refs", None)
    _cache.pop("vers", None)
    _cache.pop("nm", None)
    _cache.pop("np_copies", None)
    _cache.pop("views", None)
    _cache.pop("all_watched", None)
    _cache.pop("bkey", None)
    _cache.pop("bleaves", None)
    _cache.pop("brefs", None)
    _cache.pop("bcopies", None)
    _cache.pop("bform", None)
    _cache.pop("bns", None)
    _cache["resident"] = None
    _cache["bresident"] = None


def release():
    """Close the cached engines (device memory, streams) -- e.g. before another handle takes over the device."""
    for key in ("eng", "beng"):
        eng = _cache.pop(key, None)
        if eng is not None:
            eng.close()
    invalidate()


def _engine_for(args, n, m, device, compare_now=False):
    """The cached engine with the window given by ``args`` = (imu_meas, landmarks, landmarks_xyz, ii, time_idx,
    intrinsics, confidences) on the device.  ``compare_now``: the caller's device call does not evaluate the library's host
    watch (``vba_run_schedule``: :func:`BA_window`), so the watched buffers are compared here, before it."""
    eng = _cache.get("eng")
    if eng is None or eng.n_max < n or eng.m_max < m or eng.device != device:
        if eng is not None:
            eng.close()
        eng = _new_engine(max(n, 16), max(m, 256), 1, device)
        _cache["eng"] = eng
        invalidate()
    if _cache.get("nm") == (n, m) and _same_objects(args):
        # ndarray contents: compared by the library while the device works (resident calls) -- if every ndarray has a watch slot
        # (contiguous, at most 8 of them); else here and now
        if (_cache.get("all_watched") and not compare_now) or _numpy_unchanged(args):
            return eng
    key = (n, m) + tuple(_token(a) for a in args)
    if _cache.get("key") != key or not _numpy_unchanged(args):
        imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences = args
        cum = np.ascontiguousarray(_np(imu_meas)[0, :, -1, 6:10])
        uv = _np(landmarks).reshape(-1, 2)
        xyz = _np(landmarks_xyz).reshape(-1, 3)
        K = _np(intrinsics).reshape(-1, 4)
        conf = _np(confidences).reshape(-1)
        ii_ = np.ascontiguousarray(np.asarray(ii), dtype=np.int64).reshape(-1)
        t = np.ascontiguousarray(np.asarray(time_idx), dtype=np.int64).reshape(-1)
        if not (K.shape[0] == n and cum.shape[0] == n and t.shape[0] == n):
            raise ValueError("intrinsics / imu_meas / time_idx must have one row per pose")
        if xyz.shape[0] != m:
            raise ValueError("landmarks and landmarks_xyz disagree on the number of rows")
        eng.upload_observations(xyz, uv, conf, ii_, n)
        eng.upload_window(K, cum, t)
        _cache["key"] = key
        _cache["np_copies"] = tuple(w.copy() if w is not None else None for w in map(_watchable, args))
        _cache["resident"] = None
    # (also when only the Python objects are new -- fresh slices of the same buffers, as the reference's driver makes them)
    _cache["nm"] = (n, m)
    _cache["refs"] = args               # keeps the buffers alive: their addresses cannot be reused while cached
    _cache["vers"] = tuple(getattr(a, "_version", None) for a in args)
    # the library compares the live ndarrays with the copies that were uploaded during every resident call
    slot, all_watched, views = 0, True, []
    for a, c in zip(args, _cache["np_copies"]):
        if c is None:
            continue
        a = _watchable(a)
        views.append(a)
        if slot < _WATCH_SLOTS and a.flags.c_contiguous and c.flags.c_contiguous:
            eng.set_host_watch(slot, a, c)
            slot += 1
        else:
            all_watched = False         # (a strided view: compared on the Python side before every call instead)
    _cache["views"] = views             # (the ndarray views of watched torch tensors stay alive with the watch)
    for k in range(slot, _WATCH_SLOTS):
        eng.set_host_watch(k)
    _cache["all_watched"] = all_watched
    return eng


def _shape_of(states):
    shp = tuple(states.shape)
    if len(shp) != 3 or shp[0] != 1 or shp[2] != 10:
        raise ValueError("states must be [1, n, 10] here (BA and BA_window also take batches: [B, n, 10] or a list of windows)")
    return shp[1]


def _rows(landmarks):
    shp = landmarks.shape
    if len(shp) == 3:
        return shp[0] * shp[1]
    return int(np.prod(shp[:-1])) if len(shp) > 1 else shp[0] // 2


def _take_resident(states, lamda_init, reg):
    """True if the device holds exactly these states and this damping (the previous call's result).  The record is
    consumed either way: it is valid again only once the next call has returned."""
    r = _cache.get("resident")
    _cache["resident"] = None
    return (r is not None and r[0] is states and getattr(states, "_version", None) == r[1]
            and float(lamda_init) == r[2] and r[3] == reg)


def _wrap(out, lam, hess, reg):
    import torch
    st = torch.from_numpy(out)[None]
    _cache["resident"] = (st, st._version, lam, reg)
    return st, torch.from_numpy(hess)[None]


_HOST_CHANGED = 1 << 30


def _call(eng, reg, iter, initialize, states, lamda_init, reupload):
    """One call on the cached engine: resident if the states are the previous result, else with the states sent up.  A
    resident call during which the library found a watched ndarray edited is repeated on the window as it is now."""
    if _take_resident(states, lamda_init, reg):
        out, lam, hess, n_trials, flags = eng.iterate_resident(iter, initialize)
        if flags & _HOST_CHANGED:
            invalidate()
            eng = reupload()
            out, lam, hess, n_trials, flags = eng.iterate(iter, initialize, float(lamda_init), _np(states)[0])
    else:
        # the first call of a window's loop: the calls that follow feed its result back, so it opens the pipelined chain
        # (vba_iterate_open), and like a resident call it has the library compare the watched ndarrays while the device works;
        # ndarrays without a watch slot are compared here and now
        watched = _cache.get("all_watched")
        if not watched and not _numpy_unchanged(_cache.get("refs", ())):
            invalidate()
            eng = reupload()
        out, lam, hess, n_trials, flags = eng.iterate(iter, initialize, float(lamda_init), _np(states)[0], opening=True)
        if watched and flags & _HOST_CHANGED:
            invalidate()
            eng = reupload()
            out, lam, hess, n_trials, flags = eng.iterate(iter, initialize, float(lamda_init), _np(states)[0])
    return out, lam, hess, n_trials, flags & 7


# ------------------------------------------------------------------------------------------------ batch dimension
def _is_batch(states):
    if isinstance(states, (list, tuple)):
        return True
    shp = getattr(states, "shape", None)
    return shp is not None and len(shp) == 3 and shp[0] > 1


def _item(x, b, B, what):
    """Window ``b`` of a per-window argument given as a list of ``B`` items."""
    if not isinstance(x, (list, tuple)) or len(x) != B:
        raise ValueError(f"ragged batch: {what} must be a list of {B} items (one per window)")
    return x[b]


def _split_batch(states, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences, lamda_init):
    """Per-window views of a batch.  Returns (form, windows) with ``form`` "dense" or "ragged" and, per window, a dict of
    NumPy arrays: states [n,10], cum [n,4], uv [m,2], xyz [m,3], ii [m] int64, t [n] int64, K [n,4], conf [m], lam."""
    ragged = isinstance(states, (list, tuple))
    B = len(states) if ragged else states.shape[0]
    if B < 1:
        raise ValueError("empty batch")
    lams = [float(x) for x in lamda_init] if isinstance(lamda_init, (list, tuple, np.ndarray)) else [float(lamda_init)] * B
    if len(lams) != B:
        raise ValueError("lamda_init must be a float or one float per window")
    wins = []
    if ragged:
        for b in range(B):
            st = _np(states[b]).reshape(-1, 10)
            imu = _np(_item(imu_meas, b, B, "imu_meas"))
            imu = imu[0] if imu.ndim == 4 else imu
            wins.append(dict(states=st, cum=np.ascontiguousarray(imu[:, -1, 6:10]),
                             uv=_np(_item(landmarks, b, B, "landmarks")).reshape(-1, 2),
                             xyz=_np(_item(landmarks_xyz, b, B, "landmarks_xyz")).reshape(-1, 3),
                             ii=np.ascontiguousarray(np.asarray(_item(ii, b, B, "ii")), dtype=np.int64).reshape(-1),
                             t=np.ascontiguousarray(np.asarray(_item(time_idx, b, B, "time_idx")), dtype=np.int64).reshape(-1),
                             K=_np(_item(intrinsics, b, B, "intrinsics")).reshape(-1, 4),
                             conf=_np(_item(confidences, b, B, "confidences")).reshape(-1), lam=lams[b]))
    else:
        st_all, imu_all = _np(states), _np(imu_meas)
        uv_all, xyz_all, K_all = _np(landmarks), _np(landmarks_xyz), _np(intrinsics)
        ii_all, t_all, c_all = np.asarray(ii), np.asarray(time_idx), _np(confidences)
        if imu_all.ndim != 4 or imu_all.shape[0] not in (1, B) or uv_all.shape[0] != B or xyz_all.shape[0] != B:
            raise ValueError("dense batch: imu_meas [B,n,g,10], landmarks [B,m,2] and landmarks_xyz [B,m,3] need the batch size of states")
        for b in range(B):
            wins.append(dict(states=st_all[b], cum=np.ascontiguousarray(imu_all[b if imu_all.shape[0] == B else 0][:, -1, 6:10]),
                             uv=uv_all[b].reshape(-1, 2), xyz=xyz_all[b].reshape(-1, 3),
                             ii=np.ascontiguousarray(ii_all[b] if ii_all.ndim == 2 else ii_all, dtype=np.int64).reshape(-1),
                             t=np.ascontiguousarray(t_all[b] if t_all.ndim == 2 else t_all, dtype=np.int64).reshape(-1),
                             K=(K_all[b] if K_all.shape[0] == B else K_all[0]).reshape(-1, 4),
                             conf=(c_all[b] if c_all.ndim == 2 else c_all).reshape(-1), lam=lams[b]))
    for b, w in enumerate(wins):
        n, m = w["states"].shape[0], w["xyz"].shape[0]
        if not (w["K"].shape[0] == n and w["cum"].shape[0] == n and w["t"].shape[0] == n):
            raise ValueError(f"window {b}: intrinsics / imu_meas / time_idx must have one row per pose")
        if not (w["uv"].shape[0] == m and w["ii"].shape[0] == m and w["conf"].shape[0] == m):
            raise ValueError(f"window {b}: landmarks, landmarks_xyz, ii and confidences disagree on the number of rows")
    return ("ragged" if ragged else "dense"), wins


def _flat(x):
    """The leaves of a (possibly list-valued) argument, for identity tokens."""
    return tuple(x) if isinstance(x, (list, tuple)) else (x,)


def _leaves(args):
    return tuple(leaf for a in args for leaf in _flat(a))


def _batch_hit(args, device):
    """The cached batch engine if it holds exactly these windows: same argument buffers (identity tokens) and, for ndarray
    arguments, the bytes that were uploaded (each distinct array compared once)."""
    eng = _cache.get("beng")
    if eng is None or eng.device != device or "bkey" not in _cache:
        return None
    leaves = _leaves(args)
    # the cheap test first (what every call of a driver loop passes): the very same leaf objects as at the upload, torch tensors
    # with an unchanged in-place counter -- 154 leaves at 22 windows, whose tokens cost ~0.25 ms to build
    old = _cache.get("bleaves")
    same = old is not None and len(old[0]) == len(leaves) and all(a is b for a, b in zip(leaves, old[0])) \
        and tuple(getattr(a, "_version", None) for a in leaves) == old[1]
    if not same and _cache["bkey"] != tuple(_token(a) for a in leaves):
        return None
    seen = set()
    for a, c in zip(leaves, _cache.get("bcopies", ())):
        if c is None or id(a) in seen:
            continue
        seen.add(id(a))
        if not (a.shape == c.shape and a.dtype == c.dtype and _same_bytes(a, c)):
            return None
    if not same:        # (fresh objects over the same buffers: remember them for the next call)
        _cache["bleaves"] = (leaves, tuple(getattr(a, "_version", None) for a in leaves))
    return eng


def _batch_engine_for(args, wins, device):
    """A batch engine (one ragged handle of len(wins) windows) with these windows uploaded; becomes the cached one."""
    B = len(wins)
    n_max = max(16, max(w["states"].shape[0] for w in wins))
    m_max = max(256, max(w["xyz"].shape[0] for w in wins))
    eng = _cache.get("beng")
    if eng is None or eng.windows != B or eng.n_max < n_max or eng.m_max < m_max or eng.device != device:
        if eng is not None:
            eng.close()
        eng = _new_engine(n_max, m_max, B, device)
        _cache["beng"] = eng
    for b, w in enumerate(wins):
        eng.upload_observations(w["xyz"], w["uv"], w["conf"], w["ii"], w["states"].shape[0], window=b)
        eng.upload_window(w["K"], w["cum"], w["t"], window=b)
    leaves = _leaves(args)
    _cache["bkey"] = tuple(_token(a) for a in leaves)
    copies, made = [], {}
    for a in leaves:
        if isinstance(a, np.ndarray):
            if id(a) not in made:
                made[id(a)] = a.copy()
            copies.append(made[id(a)])
        else:
            copies.append(None)
    _cache["bcopies"] = tuple(copies)
    _cache["bleaves"] = (leaves, tuple(getattr(a, "_version", None) for a in leaves))
    _cache["brefs"] = args              # keeps the buffers alive: their addresses cannot be reused while cached
    return eng


def _batch_resident(states, lamda_init):
    """True if the device holds exactly these states and dampings: ``states`` IS what the previous batched call returned."""
    r = _cache.get("bresident")
    _cache["bresident"] = None
    if r is None:
        return False
    prev, vers, lams = r
    if isinstance(prev, list):
        same = isinstance(states, (list, tuple)) and len(states) == len(prev) and all(a is b for a, b in zip(states, prev))
        now = tuple(getattr(a, "_version", None) for a in states) if same else None
    else:
        same = states is prev
        now = (getattr(states, "_version", None),) if same else None
    lam_in = [float(x) for x in lamda_init] if isinstance(lamda_init, (list, tuple, np.ndarray)) else None
    return bool(same and now == vers and lam_in is not None and lam_in == lams)


def _batch_result(eng, form, ns):
    import torch
    S, lam, hess, ntr, flags = eng.get_states_all()
    lams = [float(x) for x in lam]
    if form == "dense":
        st = torch.from_numpy(np.ascontiguousarray(S[:, :ns[0]]))
        hs = torch.from_numpy(hess)
        _cache["bresident"] = (st, (st._version,), lams)
    else:
        st = [torch.from_numpy(np.ascontiguousarray(S[b, :n]))[None] for b, n in enumerate(ns)]
        hs = [torch.from_numpy(hess[b].copy())[None] for b in range(len(ns))]
        _cache["bresident"] = (st, tuple(x._version for x in st), lams)
    return st, lams, hs, [int(x) for x in ntr], [int(x) & 7 for x in flags]


def _batch_states_up(eng, states, lamda_init, ns):
    B = len(ns)
    lams = [float(x) for x in lamda_init] if isinstance(lamda_init, (list, tuple, np.ndarray)) else [float(lamda_init)] * B
    if len(lams) != B:
        raise ValueError("lamda_init must be a float or one float per window")
    S = np.zeros((eng.windows, eng.n_max, 10))
    for b, n in enumerate(ns):
        st = _np(states[b]).reshape(-1, 10)
        if st.shape[0] != n:
            raise ValueError(f"window {b}: states have {st.shape[0]} poses, the window {n}")
        S[b, :n] = st
    eng.set_states_all(S, lams)


def _BA_batched(iters, inits, states, velocities, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences, lamda_init, device):
    """``len(iters)`` BA() calls on every window of a batch: one ``vba_step`` (a single call) or one chained ``vba_run_schedule``."""
    args = (imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences)
    resident = _batch_resident(states, lamda_init)
    eng = _batch_hit(args, device)
    B = len(states) if isinstance(states, (list, tuple)) else states.shape[0]
    if eng is not None and _cache.get("bform") is not None and len(_cache["bns"]) == B:
        form, ns = _cache["bform"], _cache["bns"]         # the windows are on the device already
    else:
        form, wins = _split_batch(states, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences, lamda_init)
        eng = _batch_engine_for(args, wins, device)
        ns = [w["states"].shape[0] for w in wins]
        _cache["bform"], _cache["bns"] = form, ns
        resident = False
    if not resident:        # (else: the device holds these states and dampings -- the previous call's result)
        _batch_states_up(eng, states, lamda_init, ns)
    if len(iters) == 1:
        eng.step(int(iters[0]), bool(inits[0]))
    else:
        eng.run_schedule([int(x) for x in iters], [bool(x) for x in inits])
    st, lams, hs, ntr, flags = _batch_result(eng, form, ns)
    for f in flags:
        if f & 1:
            print("lamda too large")          # reference BA_filtering.py:76
    return st, lams, hs, ntr, flags


def BA(iter, states, velocities, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences, Sigma, V,
       lamda_init, poses_gt_eci, initialize=False, device=0):
    if _is_batch(states):       # bsz > 1 (dense) or a list of windows (ragged): see the module docstring
        st, lams, hs, ntr, flags = _BA_batched([iter], [initialize], states, velocities, imu_meas, landmarks, landmarks_xyz, ii,
                                               time_idx, intrinsics, confidences, lamda_init, device)
        BA.last = dict(n_trials=ntr, flags=flags)
        return (st, velocities, lams, hs)
    shp = states.shape
    if len(shp) != 3 or shp[0] != 1 or shp[2] != 10:
        _shape_of(states)
    n = shp[1]
    args = (imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences)
    m = _rows(landmarks)
    eng = _engine_for(args, n, m, device)
    out, lam, hess, n_trials, flags = _call(eng, False, int(iter), bool(initialize), states, lamda_init,
                                            lambda: _engine_for(args, n, m, device))
    if flags & 1:
        print("lamda too large")          # reference BA_filtering.py:76
    BA.last = dict(n_trials=n_trials, flags=flags)
    st, hs = _wrap(out, lam, hess, False)
    return (st, velocities, lam, hs)


BA.last = {}


def BA_reg(iter, states, velocities, states_prior, velocity_prior, hessian_state_t, hessian_rot_t, imu_meas, landmarks,
           landmarks_xyz, ii, time_idx, intrinsics, confidences, Sigma, V, lamda_init, poses_gt_eci, initialize=False,
           use_reg=True, device=0):
    """Counterpart of the reference's ``BA_reg`` (``BA_filtering.py:100-210``): ``BA`` with a propagated-covariance
    prior per pose.  Same positional signature and 4-tuple.  As in the reference ``Sigma``, ``V``, ``use_reg`` and
    ``velocity_prior`` are not read (the prior velocity is ``states_prior[..., 7:]``, ``BA_utils.py:614``), and
    ``hessian_rot_t`` has no effect on the result (see ``include/vinsat_ba.h``: the rotation term of ``prior_gpu`` is
    a constant).  The integrator is the reference's CPU branch (``predict``) unless the engine is switched."""
    n = _shape_of(states)
    eng = _engine_for((imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences), n, _rows(landmarks), device)
    pkey = (_token(states_prior), _token(hessian_state_t))
    pc = _cache.get("prior_copies", (None, None))
    same_bytes = all(c is None or np.array_equal(a, c) for a, c in zip((states_prior, hessian_state_t), pc))
    if _cache.get("prior_key") != pkey or _cache.get("prior_for") != _cache.get("key") or not same_bytes:
        sp = _np(states_prior).reshape(-1, 10)
        Hs = _np(hessian_state_t).reshape(-1, 6, 6)
        if not (sp.shape[0] == n and Hs.shape[0] == n):
            raise ValueError("the prior must have one row per pose")
        eng.upload_prior(sp, Hs)
        _cache["prior_key"], _cache["prior_for"], _cache["prior_refs"] = pkey, _cache.get("key"), (states_prior, hessian_state_t)
        _cache["prior_copies"] = tuple(a.copy() if isinstance(a, np.ndarray) else None for a in (states_prior, hessian_state_t))
        _cache["resident"] = None
    eng.set_prior(True)
    try:
        args = (imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences)
        out, lam, hess, n_trials, flags = _call(eng, True, int(iter), bool(initialize), states, lamda_init,
                                                lambda: _engine_for(args, n, _rows(landmarks), device))
    finally:
        eng.set_prior(False)
    if flags & 1:
        print("lamda too large")          # reference BA_filtering.py:186
    BA_reg.last = dict(n_trials=n_trials, flags=flags)
    st, hs = _wrap(out, lam, hess, True)
    return (st, velocities, lam, hs)


BA_reg.last = {}


def BA_window(iters, initializes, states, velocities, imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences,
              lamda_init, device=0):
    """The driver's loop ``for iter in range(num_iters): states, ... = BA(iter, states, ...)`` (reference
    ``od_pipe.py:1036-1040``) as ONE call: the states stay on the device between the calls and the calls are chained
    there (``vba_run_schedule``).  Bit-identical to calling :func:`BA` ``len(iters)`` times.

    Returns ``(states_new, velocities, lamda, last_hessian)`` of the last call, shaped like ``BA``'s.  A batch (dense
    ``[B, n, 10]`` or a list of windows, see the module docstring) runs the schedule on every window at once.
    """
    if _is_batch(states):
        st, lams, hs, ntr, flags = _BA_batched(list(iters), list(initializes), states, velocities, imu_meas, landmarks,
                                               landmarks_xyz, ii, time_idx, intrinsics, confidences, lamda_init, device)
        BA_window.last = dict(n_trials=ntr, flags=flags)
        return (st, velocities, lams, hs)
    n = _shape_of(states)
    # (vba_run_schedule does not evaluate the host watch: an ndarray edited in place since the upload is looked for here)
    eng = _engine_for((imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences), n, _rows(landmarks), device, compare_now=True)
    if not _take_resident(states, lamda_init, False):
        eng.set_states(_np(states)[0], float(lamda_init))
    eng.run_schedule(list(iters), list(initializes))
    out, lam, hess, n_trials, flags = eng.get_states()
    if flags & 1:
        print("lamda too large")
    st, hs = _wrap(out, lam, hess, False)
    return (st, velocities, lam, hs)


BA_window.last = {}
