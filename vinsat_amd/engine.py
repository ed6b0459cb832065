"""Thin object wrapper over the C ABI: one :class:`BAEngine` = one ``vba_handle``."""
from __future__ import annotations

import ctypes
from ctypes import byref, c_double, c_float, c_int, c_int64, c_uint, c_void_p

import numpy as np

from . import _lib
from ._lib import PD, PI64

DBG = dict(est=0, weight=1, H=2, b=3, Phi=4, r_pred=5, qgrad=6, Hq=7, bands=8, rhs=9, dpose=10, scalars=11, Jg=12)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(PD)


class BAEngine:
    """Device-resident bundle-adjustment context holding ``windows`` independent windows."""

    def __init__(self, n_max, m_max, windows=1, device=0, mode=-1):
        """``mode``: -1 = kernel set chosen by the window count, 0 = bandwidth-mode kernels, 1 = latency-mode kernels
        (``vba_create_mode``)."""
        self.lib = _lib.load()
        self.h = c_void_p()
        _lib.check(self.lib.vba_create_mode(device, windows, int(n_max), int(m_max), int(mode), byref(self.h)), self.lib)
        self.windows = windows
        self.n_max, self.m_max, self.device = int(n_max), int(m_max), device
        self.n = [0] * windows
        self.m = [0] * windows
        # out-parameters of the per-call entry points, made once: lamda, n_trials, flags and their addresses
        lam, nt, fl = c_double(), c_int(), c_uint()
        self._scal = (lam, nt, fl, ctypes.addressof(lam), ctypes.addressof(nt), ctypes.addressof(fl))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _option(self, name, value):
        """``vba_set_option``: the settings a caller of ``BA`` never needs (``include/vinsat_ba.h`` ``VBA_OPT_*``)."""
        _lib.check(self.lib.vba_set_option(self.h, _lib.OPT[name], int(value)), self.lib)

    def mode(self):
        """(kernel set: 1 latency / 0 bandwidth, chunk size of the solver partition: 0 = sequential walk)."""
        a, b = c_int(), c_int()
        _lib.check(self.lib.vba_get_mode(self.h, byref(a), byref(b)), self.lib)
        return a.value, b.value

    def set_solver(self, chunk, chunk2=None):
        """0 = sequential chain, 2..60 = partitioned with that chunk size (chunk2: second-level chunk size, or -1 =
        reduced system by cyclic reduction), -1 = default, -2 = sequential without packing."""
        if chunk2 is None:
            _lib.check(self.lib.vba_set_solver(self.h, int(chunk)), self.lib)
        else:
            _lib.check(self.lib.vba_set_solver2(self.h, int(chunk), int(chunk2)), self.lib)

    def set_integrator(self, hop100):
        """False: 1 s RK4 steps (reference CPU branch, default); True: <=100 s hops (the reference's predict_gpu)."""
        _lib.check(self.lib.vba_set_integrator(self.h, int(bool(hop100))), self.lib)

    def set_accumulate_lanes(self, lanes):
        """Lanes per pose of the accumulation kernel (0 = automatic)."""
        self._option("accumulate_lanes", int(lanes))

    def set_trial_tiles(self, tiles):
        """Tiles of 256 rows per observation block of the latency-mode trial kernel (0 = automatic); results do not depend on it."""
        self._option("trial_tiles", int(tiles))

    def set_schedule_graph(self, on):
        """Replay of a chained schedule's launches as a hipGraph (latency-mode handles; default on).  Same bits either way."""
        self._option("schedule_graph", int(bool(on)))

    def schedule_graph_stats(self):
        """(graphs captured, replays) of :meth:`run_schedule` so far."""
        a, b = c_int(), c_int()
        _lib.check(self.lib.vba_schedule_graph_stats(self.h, byref(a), byref(b)), self.lib)
        return a.value, b.value

    def set_key_carry(self, on):
        """True (default): an accepted trial leaves the next call's |r| keys behind; False: every call recomputes them."""
        self._option("key_carry", int(bool(on)))

    def set_fusion(self, mask):
        """Kernel fusion mask (include/vinsat_ba.h: bit 0 step inside the trial kernel, bit 1 blocks formed inside the chunk
        elimination, ... bits 5 / 6 the solve as one grid of waiting blocks -- measured slower, comparison only)."""
        self._option("fusion", int(mask))

    def set_chunk_waves(self, waves):
        """Partitioned solve: 2 (default) = every chunk is eliminated from both ends by two waves, 1 = one wave."""
        self._option("chunk_waves", int(waves))

    def set_warm_select(self, on):
        """True / 1 (default): carried keys are selected warm -- from the bucket of one warm bin inside the accumulation
        (latency mode), by one warm pass otherwise -- and chained calls fold their accept test into that; False / 0: exact
        digit passes and a decide launch per call; 2: every warm select is forced to miss (test knob); 3: the warm select
        stays a kernel of its own."""
        self._option("warm_select", int(on) if int(on) in (2, 3) else int(bool(on)))

    def set_warm_shift(self, shift):
        """Tuning / test knob: log2 of the warm-bin width in bit patterns (52 = a binade)."""
        self._option("warm_shift", int(shift))

    def set_pipeline(self, on):
        """True (default): ``iterate_resident`` overlaps the caller's turnaround with a speculatively enqueued next call."""
        self._option("pipeline", int(bool(on)))

    def pipeline_stats(self):
        """(speculated calls used, speculated calls dropped)."""
        a, b = c_int(), c_int()
        _lib.check(self.lib.vba_pipeline_stats(self.h, byref(a), byref(b)), self.lib)
        return a.value, b.value

    def set_bucket_cap(self, cap):
        """Test knob: keys a bin bucket can hold (0 = default); a fuller bin makes the call that needs it miss."""
        self._option("bucket_cap", int(cap))

    def warm_select_misses(self):
        c = c_int()
        _lib.check(self.lib.vba_warm_select_misses(self.h, byref(c)), self.lib)
        return c.value

    def set_pivoting(self, always):
        """False (default): unpivoted fast path with checked pivots and automatic fallback; True: always pivot."""
        self._option("pivoting", int(bool(always)))

    def solver_fallbacks(self):
        c = c_int()
        _lib.check(self.lib.vba_solver_fallbacks(self.h, byref(c)), self.lib)
        return c.value

    # ------------------------------------------------------------------ uploads
    def upload_observations(self, landmarks_xyz, landmarks_uv, confidences, ii, n, window=0):
        xyz, uv, conf, ii = _f64(landmarks_xyz).reshape(-1, 3), _f64(landmarks_uv).reshape(-1, 2), _f64(confidences).reshape(-1), _i64(ii).reshape(-1)
        m = xyz.shape[0]
        if not (uv.shape[0] == m and conf.shape[0] == m and ii.shape[0] == m):
            raise ValueError("observation arrays disagree on the number of rows")
        _lib.check(self.lib.vba_upload_observations(self.h, window, int(n), m, _p(xyz), _p(uv), _p(conf),
                                                    ii.ctypes.data_as(PI64)), self.lib)
        self.n[window], self.m[window] = int(n), m

    def upload_window(self, intrinsics, cumrot_last, time_idx, window=0):
        K, c, t = _f64(intrinsics).reshape(-1, 4), _f64(cumrot_last).reshape(-1, 4), _i64(time_idx).reshape(-1)
        n = K.shape[0]
        if not (c.shape[0] == n and t.shape[0] == n):
            raise ValueError("per-pose arrays disagree on the number of poses")
        _lib.check(self.lib.vba_upload_window(self.h, window, n, _p(K), _p(c), t.ctypes.data_as(PI64)), self.lib)
        self.n[window] = n

    def upload_prior(self, states_prior, hessian_state, window=0):
        """Per-pose prior of the reference's ``BA_reg``: states_prior [n,10], hessian_state_t [n,6,6]."""
        sp, H = _f64(states_prior).reshape(-1, 10), _f64(hessian_state).reshape(-1, 36)
        if sp.shape[0] != H.shape[0]:
            raise ValueError("prior arrays disagree on the number of poses")
        _lib.check(self.lib.vba_upload_prior(self.h, window, sp.shape[0], _p(sp), _p(H)), self.lib)

    def set_prior(self, on):
        """True: the following calls are ``BA_reg`` calls (need ``upload_prior``); False (default): ``BA``."""
        _lib.check(self.lib.vba_set_prior(self.h, int(bool(on))), self.lib)

    def set_states(self, states, lamda, window=0):
        s = _f64(states).reshape(-1, 10)
        _lib.check(self.lib.vba_set_states(self.h, window, _p(s), float(lamda)), self.lib)

    def get_states(self, window=0):
        n = self.n[window]
        s = np.empty((n, 10))
        lam = c_double()
        hess = np.empty((9, 9))
        nt, fl = c_int(), c_uint()
        _lib.check(self.lib.vba_get_states(self.h, window, _p(s), byref(lam), _p(hess), byref(nt), byref(fl)), self.lib)
        return s, lam.value, hess, nt.value, fl.value

    def set_states_all(self, states, lamdas):
        """Every window at once: ``states`` [W, n_max, 10] (rows beyond a window's pose count ignored), ``lamdas`` [W]."""
        s, lam = _f64(states), _f64(lamdas).reshape(-1)
        if s.shape != (self.windows, self.n_max, 10) or lam.shape[0] != self.windows:
            raise ValueError("states must be [windows, n_max, 10] and lamdas [windows]")
        _lib.check(self.lib.vba_set_states_all(self.h, _p(s), _p(lam)), self.lib)

    def get_states_all(self):
        """(states [W, n_max, 10], lamda [W], last_hessian [W, 9, 9], n_trials [W], flags [W]) of every window, one copy."""
        W = self.windows
        s = np.empty((W, self.n_max, 10))
        lam, hess = np.empty(W), np.empty((W, 9, 9))
        nt, fl = (c_int * W)(), (c_uint * W)()
        _lib.check(self.lib.vba_get_states_all(self.h, _p(s), _p(lam), _p(hess), nt, fl), self.lib)
        return s, lam, hess, np.array(nt[:], dtype=np.int64), np.array(fl[:], dtype=np.int64)

    # ------------------------------------------------------------------ compute
    def step(self, it, initialize):
        _lib.check(self.lib.vba_step(self.h, int(it), int(bool(initialize))), self.lib)

    def run_schedule(self, iters, inits):
        """len(iters) consecutive BA() calls chained on the device (one host call); returns the LM trials issued."""
        n = len(iters)
        a = (c_int * n)(*[int(x) for x in iters])
        b = (c_int * n)(*[int(bool(x)) for x in inits])
        t = c_int()
        _lib.check(self.lib.vba_run_schedule(self.h, n, a, b, byref(t)), self.lib)
        return t.value

    def iterate(self, it, initialize, lamda, states, opening=False):
        """One ``BA()`` call on window 0: returns (states_new, lamda_out, last_hessian, n_trials, flags).  ``opening``: the
        first call of a driver loop whose following calls will be :meth:`iterate_resident` (``vba_iterate_open``)."""
        s = _f64(states).reshape(-1, 10)
        out = np.empty_like(s)
        lam = c_double()
        hess = np.empty((9, 9))
        nt, fl = c_int(), c_uint()
        f = self.lib.vba_iterate_open if opening else self.lib.vba_iterate
        _lib.check(f(self.h, int(it), int(bool(initialize)), float(lamda), _p(s), _p(out), byref(lam), _p(hess), byref(nt), byref(fl)), self.lib)
        return out, lam.value, hess, nt.value, fl.value

    def iterate_resident(self, it, initialize):
        """``BA()`` on window 0 from the states and damping the previous call left on the device."""
        out = np.empty((self.n[0], 10))
        hess = np.empty((9, 9))
        sc = self._scal
        rc = self.lib.vba_iterate_resident(self.h, int(it), 1 if initialize else 0, out.__array_interface__["data"][0],
                                           sc[3], hess.__array_interface__["data"][0], sc[4], sc[5])
        if rc:
            _lib.check(rc, self.lib)
        return out, sc[0].value, hess, sc[1].value, sc[2].value

    def set_host_watch(self, slot, live=None, copy=None):
        """Watch a caller's ndarray against the copy that was uploaded (see ``vba_set_host_watch``); ``live=None`` clears."""
        if live is None:
            _lib.check(self.lib.vba_set_host_watch(self.h, int(slot), None, None, 0), self.lib)
        else:
            _lib.check(self.lib.vba_set_host_watch(self.h, int(slot), live.ctypes.data, copy.ctypes.data, live.nbytes), self.lib)

    # "begin" = two back-to-back event records (the overhead every class contains), no kernel
    KERNELS = ("begin", "residual", "select", "accumulate", "dynamics", "assemble", "solve", "trial", "decide")

    def step_profiled(self, it, initialize):
        """Like :meth:`step`, returning {kernel class: milliseconds} measured with HIP events on the library stream."""
        ms = (c_float * len(self.KERNELS))()
        _lib.check(self.lib.vba_step_profiled(self.h, int(it), int(bool(initialize)), ms), self.lib)
        return dict(zip(self.KERNELS, [float(x) for x in ms]))

    CHAIN_CLASSES = ("accumulate", "solve", "trial")

    def set_chain_profile(self, on):
        """HIP events at the class boundaries of every call of the following ``run_schedule`` calls (``VBA_OPT_CHAIN_PROFILE``)."""
        self._option("chain_profile", int(bool(on)))

    def chain_profile(self, reset=True):
        """{class: (average ms per interval, intervals)} of the chained schedules run since the last reset."""
        ms = (c_double * 3)()
        cnt = (c_int64 * 3)()
        _lib.check(self.lib.vba_chain_profile(self.h, ms, cnt, int(bool(reset))), self.lib)
        return {k: ((ms[i] / cnt[i]) if cnt[i] else 0.0, int(cnt[i])) for i, k in enumerate(self.CHAIN_CLASSES)}

    def last_step_ms(self):
        ms = c_float()
        _lib.check(self.lib.vba_last_step_ms(self.h, byref(ms)), self.lib)
        return ms.value

    def debug(self, what, window=0):
        n, m = self.n[window], self.m[window]
        shapes = dict(est=(m, 2), weight=(m,), H=(n, 6, 6), b=(n, 6), Phi=(n, 6, 6), r_pred=(n - 1, 7), qgrad=(n, 3),
                      Hq=(n, 3, 3, 3), bands=(n, 3, 9, 9), rhs=(n, 9), dpose=(n, 9), scalars=(8,), Jg=(m, 2, 6))
        shp = shapes[what]
        out = np.empty(int(np.prod(shp)))
        cnt = c_int64()
        _lib.check(self.lib.vba_debug_fetch(self.h, window, DBG[what], _p(out), out.size, byref(cnt)), self.lib)
        assert cnt.value == out.size
        return out.reshape(shp)
