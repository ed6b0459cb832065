"""The C-ABI library must load without a GPU and export every symbol include/vinsat_ba.h declares, and the ctypes
prototype table must cover exactly that set (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "vinsat_ba.h")


@pytest.fixture(scope="module")
def lib_path():
    path = os.path.join(ROOT, "vinsat_amd", "libvinsat_ba.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "vinsat_amd", "csrc"), "-j4"])
    return path


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vba_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in vinsat_ba.h but not exported"


def test_ctypes_table_matches_header(lib_path):
    from vinsat_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.load()
    assert lib.vba_version() >= 100
    assert lib.vba_sh_partial_count(10) == 272


def test_no_gpu_is_reported_not_hidden(lib_path):
    """Without a device the library must fail loudly (there is no CPU fallback)."""
    from vinsat_amd import _lib
    lib = _lib.load()
    cnt = ctypes.c_int(-1)
    assert lib.vba_device_count(ctypes.byref(cnt)) == 0
    if cnt.value == 0:
        h = ctypes.c_void_p()
        rc = lib.vba_create(0, 1, 16, 256, ctypes.byref(h))
        assert rc != 0 and not h.value
        assert b"device" in lib.vba_last_error().lower()
        from vinsat_amd.engine import BAEngine
        with pytest.raises(_lib.VbaError):
            BAEngine(16, 256)


def test_one_hip_runtime_in_the_process_whichever_is_imported_first():
    """The wheel of PyTorch bundles its own HIP runtime; the library binds to that copy (vinsat_amd/_lib.py), so that a
    process that calls BA() on numpy arrays first and touches torch.cuda later does not end up with two ROCr instances
    (the second one cannot open the device)."""
    import subprocess, sys
    code = ("from vinsat_amd import _lib; _lib.load(); import torch; "
            "maps = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}; print(len(maps))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "1"
