"""ctypes binding of libvinsat_ba.so (C ABI declared in include/vinsat_ba.h).

There is no CPU fallback: if the shared library is missing or no MI355X is visible the
loader / the first call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VBA_LIB selects another build of the same library (A/B timing of two builds on one device)
LIB_PATH = os.environ.get("VBA_LIB") or os.path.join(_HERE, "libvinsat_ba.so")

PD = POINTER(c_double)
PI64 = POINTER(c_int64)

# name -> (restype, argtypes); mirrors include/vinsat_ba.h one to one
SIGNATURES = {
    "vba_version": (c_int, []),
    "vba_has_variants": (c_int, []),
    "vba_last_error": (c_char_p, []),
    "vba_device_count": (c_int, [POINTER(c_int)]),
    "vba_create": (c_int, [c_int, c_int, c_int, c_int64, POINTER(c_void_p)]),
    "vba_create_mode": (c_int, [c_int, c_int, c_int, c_int64, c_int, POINTER(c_void_p)]),
    "vba_get_mode": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "vba_destroy": (c_int, [c_void_p]),
    "vba_set_stream": (c_int, [c_void_p, c_void_p, c_int]),
    "vba_set_option": (c_int, [c_void_p, c_int, c_int]),
    "vba_set_solver": (c_int, [c_void_p, c_int]),
    "vba_set_solver2": (c_int, [c_void_p, c_int, c_int]),
    "vba_upload_prior": (c_int, [c_void_p, c_int, c_int, PD, PD]),
    "vba_set_prior": (c_int, [c_void_p, c_int]),
    "vba_set_host_watch": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64]),
    "vba_pipeline_stats": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "vba_warm_select_misses": (c_int, [c_void_p, POINTER(c_int)]),
    "vba_schedule_graph_stats": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "vba_set_integrator": (c_int, [c_void_p, c_int]),
    "vba_solver_fallbacks": (c_int, [c_void_p, POINTER(c_int)]),
    "vba_upload_observations": (c_int, [c_void_p, c_int, c_int, c_int64, PD, PD, PD, PI64]),
    "vba_upload_window": (c_int, [c_void_p, c_int, c_int, PD, PD, PI64]),
    "vba_set_states": (c_int, [c_void_p, c_int, PD, c_double]),
    "vba_get_states": (c_int, [c_void_p, c_int, PD, PD, PD, POINTER(c_int), POINTER(c_uint)]),
    "vba_set_states_all": (c_int, [c_void_p, PD, PD]),
    "vba_get_states_all": (c_int, [c_void_p, PD, PD, PD, POINTER(c_int), POINTER(c_uint)]),
    "vba_step": (c_int, [c_void_p, c_int, c_int]),
    "vba_run_schedule": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "vba_iterate": (c_int, [c_void_p, c_int, c_int, c_double, PD, PD, PD, PD, POINTER(c_int), POINTER(c_uint)]),
    "vba_iterate_open": (c_int, [c_void_p, c_int, c_int, c_double, PD, PD, PD, PD, POINTER(c_int), POINTER(c_uint)]),
    # (buffers as plain addresses: building a typed pointer object per call costs more than the call)
    "vba_iterate_resident": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vba_debug_fetch": (c_int, [c_void_p, c_int, c_int, PD, c_int64, PI64]),
    "vba_last_step_ms": (c_int, [c_void_p, POINTER(c_float)]),
    "vba_step_profiled": (c_int, [c_void_p, c_int, c_int, POINTER(c_float)]),
    "vba_chain_profile": (c_int, [c_void_p, PD, PI64, c_int]),
    "vba_sh_partial_count": (c_int64, [c_int]),
    "vba_sh_stage1": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p]),
    "vba_sh_stage2": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "vba_sh_stage3": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "vba_sh_stage4": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_int)]),
    "vba_sh_unique_id": (c_int, [c_char_p, c_void_p]),
    "vba_sh_comm_init": (c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int]),
    "vba_sh_call": (c_int, [c_void_p, c_int, c_int, c_int64, POINTER(c_int)]),
    "vba_sh_run_schedule": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), c_int64, POINTER(c_int)]),
    "vba_sh_set_protocol": (c_int, [c_void_p, c_int]),
    "vba_sh_stats": (c_int, [c_void_p, PI64, PI64, PI64]),
    "vba_sh_comm_destroy": (c_int, [c_void_p]),
    # host-side helpers of the driver (no device)
    "vba_host_orbit_chain": (c_int, [PD, c_int, PD]),
    "vba_host_quat_chain": (c_int, [PD, PD, c_int, PD]),
    "vba_host_gap_rotations": (c_int, [PD, c_int64, PI64, c_int, PD]),
    "vba_prepare_rows": (c_int, [c_int, c_int64, PD, PI64, c_int, PD, PD, PD, PD, PD, c_void_p]),
    # free-landmark Schur add-on (parity unpinned: no counterpart in the reference)
    "vba_schur_last_error": (c_char_p, []),
    "vba_schur_create": (c_int, [c_int, c_int, c_int64, c_int, c_int, c_int64, POINTER(c_void_p)]),
    "vba_schur_destroy": (c_int, [c_void_p]),
    "vba_schur_upload": (c_int, [c_void_p] + [c_void_p] * 15 + [c_double]),
    "vba_schur_set_state": (c_int, [c_void_p, PD, PD]),
    "vba_schur_get_state": (c_int, [c_void_p, PD, PD]),
    "vba_schur_iterate": (c_int, [c_void_p, c_double, PD, PD, POINTER(c_int)]),
    "vba_schur_last_info": (c_int, [c_void_p, POINTER(c_int)]),
    "vba_schur_last_ms": (c_int, [c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_float)]),
    "vba_schur_debug_fetch": (c_int, [c_void_p, c_int, PD, c_int64]),
}

# VBA_OPT_* of include/vinsat_ba.h (vba_set_option)
OPT = dict(accumulate_lanes=1, trial_tiles=2, key_carry=3, warm_select=4, warm_shift=5, bucket_cap=6, fusion=7, chunk_waves=8,
           pivoting=9, pipeline=10, schedule_graph=11, chain_profile=12)

_lib = None


class VbaError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64; a process that
    loads the system copy first (through this library) and torch's afterwards ends up with two ROCr instances, and the
    second one cannot open the device (``No HIP GPUs are available``).  Both copies carry the SONAME libamdhip64.so.7,
    so loading torch's copy first makes the dynamic loader bind this library to it, whichever is imported first."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load the shared library and set the prototypes (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VbaError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C vinsat_amd/csrc` (there is no CPU fallback)")
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def as_pd(a):
    """A C-contiguous float64 ndarray as ``double*`` (the array must outlive the call)."""
    return a.ctypes.data_as(PD)


def as_pi64(a):
    return a.ctypes.data_as(PI64)


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        msg = lib.vba_last_error()
        raise VbaError(f"libvinsat_ba error {rc}: {msg.decode() if msg else ''}")
