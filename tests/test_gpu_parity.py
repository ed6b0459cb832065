"""Parity of the HIP path (through the C ABI) with the reference's golden vectors and with the oracle."""
import numpy as np
import pytest

from conftest import golden_inputs, load_golden, rel_err
from oracle import ba_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_c1(c1):
    from vinsat_amd.engine import BAEngine
    inp = golden_inputs(c1)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    e = BAEngine(n, m)
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_c2(c2):
    from vinsat_amd.engine import BAEngine
    inp = golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    e = BAEngine(n, m)
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    yield e
    e.close()


def _check_call(eng, g, k, full):
    n = eng.n[0]
    st_in = g[f"states_in_{k}"][0] if f"states_in_{k}" in g else (g["states0"][0] if k == 0 else g[f"states_out_{k-1}"][0])
    init = bool(g["initialize"][k])
    out, lam, hess, ntr, flags = eng.iterate(int(g["iters"][k]), init, float(g["lamda_in"][k]), st_in)
    assert flags == 0
    assert ntr == g["n_trials"][k]                      # integer: exact
    assert lam == g["lamda_out"][k]
    ref = g[f"states_out_{k}"][0]
    assert np.abs(out[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-9
    assert rel_err(out, ref) < 1e-8
    assert rel_err(hess, g[f"last_hessian_{k}"][0]) < 1e-10
    if not full:
        return
    assert rel_err(eng.debug("est"), g[f"landmark_est_{k}"][0]) < 1e-13
    assert rel_err(eng.debug("Jg"), g[f"Jg_{k}"][:, :, :6]) < 1e-12
    A = eng.debug("bands")
    sc = eng.debug("scalars")
    A[:, 1] += sc[4] * np.eye(9)
    assert sc[4] == float(np.float32(g["lamda_in"][k]))
    assert rel_err(A, g[f"A_bands_{k}"][0]) < 1e-11
    assert rel_err(eng.debug("rhs"), g[f"JTr_{k}"][0].reshape(n, 9)) < 1e-9
    assert rel_err(eng.debug("dpose"), g[f"dpose_{k}"][0].reshape(n, 9)) < 2e-7
    if not init:
        assert np.abs(eng.debug("r_pred") - g[f"r_pred_{k}"][0]).max() < 1e-9
        D = np.array([1, 1, 1, 100.0, 100, 100])
        Phi = eng.debug("Phi")
        Jf = g[f"Jf_blocks_{k}"][:, 0]
        assert rel_err((D[None, :, None] * Phi[:-1])[:, :, :3], Jf[:, :, :3]) < 1e-12
        assert rel_err((D[None, :, None] * Phi[:-1])[:, :, 3:], Jf[:, :, 6:]) < 1e-12
        assert rel_err(eng.debug("qgrad"), g[f"qgrad_{k}"][0][:, 3:6]) < 1e-9
        Hq = eng.debug("Hq")
        ref_hq = g[f"Hq_bands_{k}"][:, :, 3:6, 3:6]
        assert rel_err(Hq[:, 1], ref_hq[:, 1]) < 1e-12
        assert rel_err(Hq[:-1, 2], ref_hq[:-1, 2]) < 1e-12
        assert rel_err(Hq[1:, 0], ref_hq[1:, 0]) < 1e-12


@pytest.mark.parametrize("k", range(20))
def test_c1_every_call_with_intermediates(eng_c1, c1, k):
    _check_call(eng_c1, c1, k, full=True)


@pytest.mark.parametrize("k", [0, 9, 10, 19])
def test_c2_calls_with_intermediates(eng_c2, c2, k):
    _check_call(eng_c2, c2, k, full=True)


@pytest.mark.parametrize("k", [1, 2, 3, 5, 11, 15])
def test_c2_calls(eng_c2, c2, k):
    _check_call(eng_c2, c2, k, full=False)


def test_c2_weights_and_blocks_vs_oracle(eng_c2, c2):
    g, inp = c2, golden_inputs(c2)
    for k in (1, 2, 12):
        st = g[f"states_out_{k-1}"][0]
        dbg = {}
        O.ba_iteration(int(g["iters"][k]), st, inp["cumrot"], inp["uv"], inp["xyz"], inp["ii"], inp["time_idx"], inp["K"],
                       inp["conf"], float(g["lamda_in"][k]), initialize=bool(g["initialize"][k]), debug=dbg)
        eng_c2.iterate(int(g["iters"][k]), bool(g["initialize"][k]), float(g["lamda_in"][k]), st)
        sc = eng_c2.debug("scalars")
        assert sc[0] == dbg["c_obs"]                      # exact lower median: bit-exact
        assert rel_err(sc[1], dbg["wmax"]) < 1e-14
        assert rel_err(eng_c2.debug("weight"), dbg["w"]) < 1e-13
        assert rel_err(eng_c2.debug("H"), dbg["H"]) < 1e-12
        assert rel_err(eng_c2.debug("b"), dbg["b"]) < 1e-10
        assert rel_err(sc[2], dbg["init_residual"]) < 1e-13
        assert rel_err(sc[3], dbg["trials"][-1]["residual"]) < 1e-9


def test_c2_chained_20_calls(eng_c2, c2):
    """The BASELINE parity bar: states after the 20 calls of a window within 1e-6 of the reference."""
    g = c2
    st, lam = g["states0"][0], 1e-4
    for k in range(20):
        st, lam, _, ntr, flags = eng_c2.iterate(int(g["iters"][k]), bool(g["initialize"][k]), lam, st)
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k] and flags == 0
        assert rel_err(st, g[f"states_out_{k}"][0]) < 1e-7
    ref = g["states_out_19"][0]
    assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-8
    q, qr = st[:, 3:7], ref[:, 3:7]
    ang = 2 * np.arccos(np.clip(np.abs((q * qr).sum(-1)), 0, 1))
    assert ang.max() < 1e-6


def test_run_to_run_bit_stable(eng_c2, c2):
    g = c2
    a = eng_c2.iterate(12, False, 1e-4, g["states_out_11"][0])[0]
    b = eng_c2.iterate(12, False, 1e-4, g["states_out_11"][0])[0]
    assert np.array_equal(a, b)
