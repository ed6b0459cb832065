import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "variants: needs the comparison build of the library (make VARIANTS=1; deselected otherwise)")


def pytest_collection_modifyitems(config, items):
    """Tests of the comparison-only solver variants (marker ``variants``) run against a library built with them
    (make VARIANTS=1, VBA_LIB=...); against the default library they are deselected."""
    has = False
    try:
        from vinsat_amd import _lib
        has = bool(_lib.load().vba_has_variants())
    except Exception:
        has = False
    if has:
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("variants") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def pytest_sessionstart(session):
    """The shared library is built in-tree by ``__graft_entry__.build()`` and is not under version control; a fresh
    checkout builds it here (hipcc cross-compiles without a GPU).  A failed build is left for the tests to report."""
    lib = os.path.join(ROOT, "vinsat_amd", "libvinsat_ba.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "vinsat_amd", "csrc")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def golden_inputs(g):
    """BA inputs of a fixture that stores them (C1, C2), without batch dimension."""
    return dict(xyz=g["in_landmarks_xyz"][0], uv=g["in_landmarks"][0], ii=g["in_ii"], time_idx=g["in_time_idx"],
                K=g["in_intrinsics"][0], conf=g["in_confidences"], cumrot=g["in_cumrot_last"])


@pytest.fixture(scope="session")
def c1():
    return load_golden("c1")


@pytest.fixture(scope="session")
def c2():
    return load_golden("c2")
