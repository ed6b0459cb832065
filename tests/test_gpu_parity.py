"""Parity of the HIP path (through the C ABI) with the reference's golden vectors and with the oracle."""
import numpy as np
import pytest

from conftest import ROOT, golden_inputs, load_golden, rel_err
from oracle import ba_oracle as O

pytestmark = pytest.mark.gpu

# dpose of one solve against the reference's dense LU (torch.linalg.solve), relative to max |dpose| of the system.
# <= 1e-7.  Measured per solver variant on every captured system (profiles/r02_dpose_error_by_variant.json,
# tools/dpose_error_table.py): the default and every other variant WITHOUT row exchanges stay below 3.6e-9 -- closer to the
# reference than LAPACK's banded LU on the same captured matrix (4.3e-8); the variants that exchange rows inside a 9x9
# block (the fallback for indefinite blocks, never taken on these systems by default) reach 1.43e-7 on the worst one (C1
# call 10, cond ~1e14).  The bar is 1e-7 for the default path and 2e-7 for the pivoted fallbacks.
DPOSE_TOL = 1e-7
DPOSE_TOL_PIVOTED = 2e-7


# solver: -1 = default (chain cut into ~sqrt(n) chunks), 0 = one wave walks the whole chain, 7 = odd chunk size
@pytest.fixture(scope="module", params=[(-1, False), (0, False), (-1, True)], ids=["partitioned", "sequential", "partitioned-pivot"])
def eng_c1(c1, request):
    from vinsat_amd.engine import BAEngine
    inp = golden_inputs(c1)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    e = BAEngine(n, m)
    e.set_solver(request.param[0])
    e.set_pivoting(request.param[1])
    e.pivoted = request.param[1]
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    yield e
    e.close()


@pytest.fixture(scope="module", params=[(-1, False), (0, False), (7, False), (-1, True), (0, True), ((5, 4), False), ((3, 2), True),
                                        ((7, -1), False), ((3, -1), True), ((5, 0), False)],
                ids=["default", "sequential", "chunk7", "default-pivot", "sequential-pivot", "two-level-5-4", "two-level-3-2-pivot",
                     "cyclic-7", "cyclic-3-pivot", "one-level-5"])
def eng_c2(c2, request):
    from vinsat_amd.engine import BAEngine
    inp = golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    e = BAEngine(n, m)
    if isinstance(request.param[0], tuple):
        e.set_solver(*request.param[0])
    else:
        e.set_solver(request.param[0])
    e.set_pivoting(request.param[1])
    e.pivoted = request.param[1]
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
    assert rel_err(eng.debug("dpose"), g[f"dpose_{k}"][0].reshape(n, 9)) < (DPOSE_TOL_PIVOTED if getattr(eng, "pivoted", False) else DPOSE_TOL)
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
        # the select is exact: recomputing the lower median on the host from the device's own reprojection
        # gives the same bits (the oracle's value differs in the last ulp because its residuals do)
        a = np.abs(inp["uv"] - eng_c2.debug("est")).reshape(-1)
        assert sc[0] == np.sort(a)[(a.size - 1) // 2]
        # vs the oracle: the reference rotates with quaternion products (BA_utils.py:1052-1069), the kernels with the matrix of
        # the normalised quaternion; one ulp of a matrix entry is ~2e-12 px here (f ~ 17 500 px, range ~ 600 km)
        assert rel_err(sc[0], dbg["c_obs"]) < 1e-11
        assert rel_err(sc[1], dbg["wmax"]) < 1e-14
        assert rel_err(eng_c2.debug("weight"), dbg["w"]) < 1e-11     # pow() differs in the last bits
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


# ------------------------------------------------------------------------------------------------ full-size windows
def _window_from_seed(name):
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence(name)
    win = od_pipe.prepare_window(det, orb)
    return win


def _digest(v):
    v = np.asarray(v, dtype=np.float64).reshape(-1)
    return np.array([v.size, v.sum(), np.abs(v).sum(), v[0], v[v.size // 2], v[-1]])


@pytest.mark.parametrize("solver", [-1, 0], ids=["partitioned", "sequential"])
@pytest.mark.parametrize("name", ["c3", "c4"])
def test_headline_windows_chained_vs_reference_states(name, solver):
    """C3 (500 poses / 50k observations) and C4 (500 / 200k): inputs regenerated from the seed, checked
    against the digest of what the reference was fed, then all 20 calls chained on the GPU and compared with
    the reference's states after calls 0, 9, 10, 14, 19."""
    import os
    from conftest import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, f"{name}.npz")):
        pytest.skip(f"{name} fixture not generated")
    from vinsat_amd.engine import BAEngine
    g = load_golden(name)
    win = _window_from_seed(name.upper())
    for key, arr in (("landmarks", win.landmarks_uv), ("landmarks_xyz", win.landmarks_xyz), ("confidences", win.confidences),
                     ("intrinsics", win.intrinsics), ("cumrot_last", win.cumrot_last)):
        assert rel_err(_digest(arr), g["digest_" + key]) < 1e-12, key
    assert np.array_equal(win.time_idx, g["in_time_idx"])
    assert np.array_equal(np.array([win.ii.size, win.ii.sum(), win.ii[0], win.ii[-1]]), g["in_ii_digest"])
    n, m = win.time_idx.shape[0], win.ii.shape[0]
    eng = BAEngine(n, m)
    eng.set_solver(solver)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    st, lam = g["states0"][0], 1e-4
    for k in range(20):
        st, lam, _, ntr, flags = eng.iterate(int(g["iters"][k]), bool(g["initialize"][k]), lam, st)
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k] and flags == 0
        if f"states_out_{k}" in g:
            ref = g[f"states_out_{k}"][0]
            assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6, k
            q, qr = st[:, 3:7], ref[:, 3:7]
            assert (2 * np.arccos(np.clip(np.abs((q * qr).sum(-1)), 0, 1))).max() < 1e-6, k
            assert rel_err(st, ref) < 1e-6, k
    eng.close()


# ------------------------------------------------------------------------------------------------ LM loop
def _oracle_vs_gpu(eng, win_args, it, init, lam, st, tol=1e-7):
    cum, uv, xyz, ii, t, K, conf = win_args
    ref, lam_ref, hess_ref, ntr_ref = O.ba_iteration(it, st, cum, uv, xyz, ii, t, K, conf, lam, initialize=init)
    out, lam_g, hess, ntr, flags = eng.iterate(it, init, lam, st)
    assert ntr == ntr_ref
    assert lam_g == lam_ref
    assert rel_err(out, ref) < tol
    assert rel_err(hess, hess_ref) < 1e-9
    return out, lam_g, ntr, flags


def test_rejected_trials_and_lambda_exhaustion(eng_c2, c2):
    """Drive the LM loop through several trials (BA_filtering.py:52-77).  Near the optimum with unit
    confidences the accept test (weighted trial vs unweighted initial residual, :51 vs :66) rejects the small
    dampings; with confidences of 3 it can never pass, so lamda runs out after 9 trials and the last trial is kept."""
    g, inp = c2, golden_inputs(c2)
    n = g["states0"].shape[1]
    base = [inp["cumrot"], inp["uv"], inp["xyz"], inp["ii"], inp["time_idx"], inp["K"], None]
    try:
        conf = np.ones_like(inp["conf"])
        eng_c2.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
        base[6] = conf
        st, lam = g["states_out_19"][0].copy(), 1e-4
        seen = []
        for rep in range(4):
            st, lam, ntr, flags = _oracle_vs_gpu(eng_c2, tuple(base), 0, True, lam, st, tol=1e-6)
            seen.append(ntr)
            assert flags == 0
        assert max(seen) >= 4, seen
        st5, _, ntr, _ = _oracle_vs_gpu(eng_c2, tuple(base), 5, True, 1e-4, g["states_out_19"][0], tol=1e-6)
        assert ntr > 1
        conf = np.full_like(inp["conf"], 3.0)
        eng_c2.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
        base[6] = conf
        out, lam_o, ntr, flags = _oracle_vs_gpu(eng_c2, tuple(base), 0, True, 1e-4, g["states_out_19"][0], tol=1e-6)
        assert ntr == 9 and (flags & 1) and lam_o == 0.1
        out, lam_o, ntr, flags = _oracle_vs_gpu(eng_c2, tuple(base), 12, False, 1e-2, g["states_out_19"][0], tol=1e-6)
        assert ntr == 7 and (flags & 1)
    finally:
        eng_c2.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)


# ------------------------------------------------------------------------------------------------ edge cases
def test_unsorted_rows_empty_poses_and_ragged_gaps():
    """Rows in arbitrary order, poses without any observation, uneven time gaps (1..37 s) and a very uneven
    number of observations per pose; compared with the oracle."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence(synth.WindowConfig("edge", 24, 37, 5), seed=3)
    win = od_pipe.prepare_window(det, orb)
    rng = np.random.default_rng(11)
    # drop observations: poses 3 and 17 get none, pose 5 keeps one row, others a random subset
    keep = rng.random(win.ii.size) < 0.7
    keep[(win.ii == 3) | (win.ii == 17)] = False
    idx5 = np.nonzero(win.ii == 5)[0]
    keep[idx5] = False
    keep[idx5[0]] = True
    order = rng.permutation(np.nonzero(keep)[0])
    xyz, uv, conf, ii = win.landmarks_xyz[order], win.landmarks_uv[order], win.confidences[order], win.ii[order]
    conf = conf * rng.uniform(0.5, 1.0, size=conf.shape)
    # ragged gaps: re-time the poses (dynamics residuals become large; parity is what matters here)
    t = np.cumsum(np.concatenate([[10], rng.integers(1, 38, size=win.time_idx.size - 1)])).astype(np.int64)
    n = t.size
    eng = BAEngine(n, ii.size)
    eng.upload_observations(xyz, uv, conf, ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, t)
    args = (win.cumrot_last, uv, xyz, ii, t, win.intrinsics, conf)
    st = od_pipe.initial_guess(win)
    lam = 1e-4
    for it, init in ((0, True), (1, True), (2, True), (3, True), (10, False), (11, False)):
        st, lam, ntr, flags = _oracle_vs_gpu(eng, args, it, init, lam, st, tol=1e-7)
    H = eng.debug("H")
    assert np.all(H[3] == 0) and np.all(H[17] == 0)
    eng.close()


def test_minimum_window_two_poses_one_observation_each():
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence(synth.WindowConfig("tiny", 2, 1, 5), seed=1)
    win = od_pipe.prepare_window(det, orb)
    eng = BAEngine(2, 2)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, 2)
    eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    args = (win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii, win.time_idx, win.intrinsics, win.confidences)
    st = win.states_gt.copy()
    st[:, :3] += 1.0
    _oracle_vs_gpu(eng, args, 0, True, 1e-4, st, tol=1e-7)
    _oracle_vs_gpu(eng, args, 12, False, 1e-2, st, tol=1e-7)
    eng.close()


def test_argument_errors_are_reported():
    from vinsat_amd._lib import VbaError
    from vinsat_amd.engine import BAEngine
    eng = BAEngine(8, 64)
    with pytest.raises(VbaError):
        eng.step(0, True)                                  # nothing uploaded
    with pytest.raises(VbaError):
        eng.upload_observations(np.zeros((4, 3)), np.zeros((4, 2)), np.ones(4), np.array([0, 1, 9, 2]), 8)   # ii out of range
    with pytest.raises(VbaError):
        eng.upload_window(np.ones((8, 4)), np.ones((8, 4)), np.array([0, 1, 2, 2, 3, 4, 5, 6]))              # repeated time
    with pytest.raises(VbaError):
        eng.upload_observations(np.zeros((100, 3)), np.zeros((100, 2)), np.ones(100), np.zeros(100, dtype=np.int64), 8)  # m > m_max
    # vba_set_option: an option that does not exist, values an option does not take
    from vinsat_amd import _lib
    lib = _lib.load()
    assert lib.vba_set_option(eng.h, 999, 0) != 0 and b"option" in lib.vba_last_error()
    for name, value in (("trial_tiles", 3), ("accumulate_lanes", 5), ("chunk_waves", 7), ("warm_shift", 99)):
        assert lib.vba_set_option(eng.h, _lib.OPT[name], value) != 0, name
    assert lib.vba_set_option(eng.h, _lib.OPT["fusion"], 32) != 0          # (bits 4 .. 6: the comparison build only)
    for name, value in (("trial_tiles", 2), ("accumulate_lanes", 16), ("chunk_waves", 1), ("pipeline", 0), ("schedule_graph", 0)):
        assert lib.vba_set_option(eng.h, _lib.OPT[name], value) == 0, name
    eng.close()


# ------------------------------------------------------------------------------------------------ batched windows
def test_batched_windows_equal_single_window_runs(c2):
    """W windows in one handle (different data, different sizes) give bit-identical results to W separate runs."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    wins = []
    for seed, cfg in ((0, synth.WindowConfig("a", 40, 30, 5)), (1, synth.WindowConfig("b", 64, 17, 5)),
                      (2, synth.WindowConfig("c", 33, 50, 5))):
        det, orb = synth.make_sequence(cfg, seed=seed)
        wins.append(od_pipe.prepare_window(det, orb))
    n_max = max(w.time_idx.size for w in wins)
    m_max = max(w.ii.size for w in wins)
    sched = [(0, True), (1, True), (4, True), (10, False), (11, False)]
    singles = []
    for w in wins:
        e = BAEngine(n_max, m_max)
        e.set_accumulate_lanes(8)       # same reduction tree in both runs
        e.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, w.time_idx.size)
        e.upload_window(w.intrinsics, w.cumrot_last, w.time_idx)
        e.set_states(od_pipe.initial_guess(w), 1e-4)
        for it, init in sched:
            e.step(it, init)
        singles.append(e.get_states())
        e.close()
    e = BAEngine(n_max, m_max, windows=3)
    e.set_accumulate_lanes(8)
    for k, w in enumerate(wins):
        e.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, w.time_idx.size, window=k)
        e.upload_window(w.intrinsics, w.cumrot_last, w.time_idx, window=k)
        e.set_states(od_pipe.initial_guess(w), 1e-4, window=k)
    for it, init in sched:
        e.step(it, init)
    for k in range(3):
        s, lam, hess, ntr, flags = e.get_states(window=k)
        assert np.array_equal(s, singles[k][0]) and lam == singles[k][1] and np.array_equal(hess, singles[k][2])
    e.close()


# ------------------------------------------------------------------------------------------------ drop-in surface
def test_BA_call_surface_matches_reference_signature(c1):
    import torch
    from vinsat_amd.ba import BA
    g, inp = c1, golden_inputs(c1)
    n = inp["K"].shape[0]
    imu = torch.zeros((1, n, 5, 10), dtype=torch.float64)
    imu[0, :, -1, 6:] = torch.from_numpy(inp["cumrot"])
    vel = torch.from_numpy(g["in_velocities"])
    states, lam = torch.from_numpy(g["states0"]), 1e-4
    for k in range(20):
        states, v_out, lam, hess = BA(int(g["iters"][k]), states, vel, imu, torch.from_numpy(inp["uv"])[None],
                                      torch.from_numpy(inp["xyz"])[None], inp["ii"], inp["time_idx"],
                                      torch.from_numpy(inp["K"])[None], torch.from_numpy(inp["conf"]), 1e-3, 1e-3, lam,
                                      torch.from_numpy(g["in_poses_gt_eci"]), initialize=bool(g["initialize"][k]))
        assert v_out is vel and states.shape == (1, n, 10) and hess.shape == (1, 9, 9) and isinstance(lam, float)
    assert rel_err(states[0].numpy(), g["states_out_19"][0]) < 1e-7


# ------------------------------------------------------------------------------------------------ sharded stages
class _EmulatedRanks:
    """R ranks of the observation-sharded mode on ONE GPU: one engine per rank holding its row slice, the three RCCL
    all-gathers replaced by device-side concatenation.  Drives the HIP stage entry points vba_sh_stage1..4 exactly as
    vinsat_amd/dist.py:ShardedBA.step does."""

    def __init__(self, n, m, ranks, xyz, uv, conf, ii, K, cumrot, time_idx):
        import torch
        from vinsat_amd.dist import HipStageEngine, shard_bounds
        from vinsat_amd.engine import BAEngine
        self.torch, self.n, self.m, self.ranks = torch, n, m, ranks
        b = shard_bounds(m, ranks)
        m_pad = -(-m // ranks)
        self.engs = []
        for r in range(ranks):
            lo, hi = int(b[r]), int(b[r + 1])
            e = BAEngine(n, hi - lo)
            e.upload_observations(xyz[lo:hi], uv[lo:hi], conf[lo:hi], ii[lo:hi], n)
            e.upload_window(K, cumrot, time_idx)
            self.engs.append(HipStageEngine(e))
        pc = self.engs[0].partial_count(n)
        f64 = dict(dtype=torch.float64, device="cuda")
        self.abs_l = [torch.full((2 * m_pad,), float("inf"), **f64) for _ in range(ranks)]
        self.part_l = [torch.empty(pc, **f64) for _ in range(ranks)]
        self.trial_l = [torch.empty(2, **f64) for _ in range(ranks)]

    def set_states(self, st, lam):
        for e in self.engs:
            e.set_states(st, lam)

    def call(self, it, init):
        """One BA() call on every emulated rank; returns the number of stage-3 rounds (LM trials + pivoted repeats)."""
        torch, engs = self.torch, self.engs
        for r, e in enumerate(engs):
            e.stage1(it, init, self.m, self.abs_l[r])
        abs_all = torch.cat(self.abs_l)
        for r, e in enumerate(engs):
            e.stage2(abs_all, self.part_l[r])
        part_all = torch.cat(self.part_l)
        first, rounds = True, 0
        while True:
            for r, e in enumerate(engs):
                e.stage3(part_all if first else None, self.ranks, self.trial_l[r])
            trial_all = torch.cat(self.trial_l)
            done = [e.stage4(trial_all, self.ranks) for e in engs]
            first = False
            rounds += 1
            assert all(d == done[0] for d in done)
            if done[0]:
                return rounds
            assert rounds < 12

    def results(self):
        outs = [e.get_states() for e in self.engs]
        for o in outs:      # every rank holds bit-identical normal equations, so bit-identical states and damping
            assert np.array_equal(o[0], outs[0][0]) and o[1] == outs[0][1]
        return outs[0]

    def close(self):
        for e in self.engs:
            e.close()


@pytest.mark.parametrize("ranks", [2, 3])
def test_sharded_stage_kernels_on_one_gpu(c2, ranks):
    """The HIP stage entry points (vba_sh_stage1..4) driven for R emulated ranks on one GPU: buffers are
    concatenated on the device in place of the RCCL all-gathers.  Must agree with the unsharded HIP path to
    rounding and every emulated rank must end with bit-identical states."""
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n = inp["K"].shape[0]
    m = inp["xyz"].shape[0] - 3
    xyz, uv, conf, ii = inp["xyz"][:m], inp["uv"][:m], inp["conf"][:m].copy(), inp["ii"][:m]
    conf[::7] = 2.5
    if ranks == 3:
        conf[ii % 5 == 0] = -0.4       # indefinite blocks: the unpivoted path must fall back on every emulated rank alike
    em = _EmulatedRanks(n, m, ranks, xyz, uv, conf, ii, inp["K"], inp["cumrot"], inp["time_idx"])
    single = BAEngine(n, m)
    single.upload_observations(xyz, uv, conf, ii, n)
    single.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    st, lam = g["states0"][0], 1e-4
    ref, lam_ref = st.copy(), lam
    em.set_states(st, lam)
    saw_multi = False
    for it, init in [(0, True), (1, True), (2, True), (5, True), (10, False), (11, False), (12, False)]:
        rounds = em.call(it, init)
        ref, lam_ref, hess_ref, ntr_ref, flags_ref = single.iterate(it, init, lam_ref, ref)
        out = em.results()
        assert ntr_ref == out[3] and rounds >= ntr_ref      # a fallback round repeats a trial without counting it
        saw_multi |= rounds > 1
        assert out[1] == lam_ref
        assert rel_err(out[0], ref) < 1e-9
        assert rel_err(out[2], hess_ref) < 1e-7     # different accumulation tree (handle geometry differs)
    if ranks == 3:
        assert all(e.eng.solver_fallbacks() > 0 for e in em.engs)
    em.close()
    single.close()


def test_sharded_c4_window_on_eight_emulated_ranks_vs_reference_states():
    """BASELINE config 4 -- 500 poses / 200 000 rows, landmarks sharded over 8 ranks -- through the HIP stage kernels
    with 8 emulated ranks on one GPU: all 20 calls of the driver's schedule chained, trial counts and lamda exact and the
    states after calls 0, 9, 10, 14, 19 against the reference's own run (tests/golden/c4.npz)."""
    g = load_golden("c4")
    win = _window_from_seed("C4")
    assert np.array_equal(np.array([win.ii.size, win.ii.sum(), win.ii[0], win.ii[-1]]), g["in_ii_digest"])
    n, m = win.time_idx.shape[0], win.ii.shape[0]
    assert (n, m) == (500, 200000)
    em = _EmulatedRanks(n, m, 8, win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, win.intrinsics, win.cumrot_last,
                        win.time_idx)
    em.set_states(g["states0"][0], 1e-4)
    for k in range(20):
        em.call(int(g["iters"][k]), bool(g["initialize"][k]))
        st, lam, _, ntr, flags = em.results()
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k] and flags == 0, k
        if f"states_out_{k}" in g:
            ref = g[f"states_out_{k}"][0]
            assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6, k
            q, qr = st[:, 3:7], ref[:, 3:7]
            assert (2 * np.arccos(np.clip(np.abs((q * qr).sum(-1)), 0, 1))).max() < 1e-6, k
            assert rel_err(st, ref) < 1e-6, k
    em.close()


def test_median_with_massive_ties_and_signed_zero():
    """All residuals identical (and a block of exact zeros): the radix select must still return the exact lower
    median; exercises the compaction-list overflow path of the select."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence(synth.WindowConfig("ties", 16, 40, 5), seed=2)
    win = od_pipe.prepare_window(det, orb)
    n, m = win.time_idx.size, win.ii.size
    st = win.states_gt.copy()
    # measurements = exact reprojection at these states, shifted by the same offset everywhere -> |r| all equal
    est = O.landmark_project(st, win.landmarks_xyz, win.intrinsics, win.ii, jacobian=False)
    for shift in (0.0, 3.0):
        uv = est + shift
        eng = BAEngine(n, m)
        eng.upload_observations(win.landmarks_xyz, uv, win.confidences, win.ii, n)
        eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
        if shift == 0.0:
            continue        # c_obs ~ 1e-13: weights overflow in the reference too; only the non-degenerate case is compared
        out, lam, hess, ntr, flags = eng.iterate(3, True, 1e-4, st)
        sc = eng.debug("scalars")
        a = np.abs(uv - eng.debug("est")).reshape(-1)
        assert sc[0] == np.sort(a)[(a.size - 1) // 2]
        ref = O.ba_iteration(3, st, win.cumrot_last, uv, win.landmarks_xyz, win.ii, win.time_idx, win.intrinsics,
                             win.confidences, 1e-4, initialize=True)
        assert ntr == ref[3] and rel_err(out, ref[0]) < 1e-7
        eng.close()


# ------------------------------------------------------------------------------------------------ driver on the GPU
def test_streaming_driver_two_pass_on_gpu():
    """The drop-in driver + HIP BA on the two-pass sequence against the reference's own run: 40 BA calls, growing
    window (12 then 25 poses), a pose without observations, RK4 chains of ~900 steps inside the dynamics kernel."""
    from vinsat_amd import od_pipe, synth
    g = load_golden("gap")
    det, orb = synth.make_two_pass_sequence()
    rec = []
    errors, first_det, times = od_pipe.streaming_version(detections=det, orbit_np=orb, record=rec)
    assert [r["states"].shape[1] for r in rec] == list(g["n_poses_per_call"])
    for k in range(40):
        ref = g[f"states_out_{k}"][0]
        st = rec[k]["states"][0].numpy()
        assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6, k
        assert rel_err(st, ref) < 1e-6, k
        assert rec[k]["lamda"] == g["lamda_out"][k]
    assert rel_err(errors.numpy(), g["errors"]) < 1e-5
    assert int(first_det) == int(g["first_detection"])
    assert np.array_equal(np.concatenate([np.atleast_1d(t) for t in times]), g["times"])


def test_streaming_driver_single_pass_on_gpu(c1):
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence("C1")
    rec = []
    errors, first_det, times = od_pipe.streaming_version(detections=det, orbit_np=orb, record=rec)
    assert rel_err(rec[-1]["states"][0].numpy(), c1["states_out_19"][0]) < 1e-7
    assert rel_err(errors.numpy(), c1["errors"]) < 1e-6


def test_unpivoted_fast_path_falls_back_when_a_pivot_check_fails(c2):
    """Negative confidences make the normal equations indefinite: the positive-definite fast path must notice
    (pivot check), repeat the solve with row pivoting, and agree with the oracle's LAPACK banded LU."""
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    conf = inp["conf"].copy()
    conf[inp["ii"] % 3 == 0] = -0.5
    for solver in (-1, 0):
        eng = BAEngine(n, m)
        eng.set_solver(solver)
        eng.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
        eng.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
        args = (inp["cumrot"], inp["uv"], inp["xyz"], inp["ii"], inp["time_idx"], inp["K"], conf)
        st = g["states_out_19"][0]
        for it, init, lam in ((3, True, 1e-4), (12, False, 1e-2)):
            ref, lam_ref, hess_ref, ntr_ref = O.ba_iteration(it, st, *args, lam, initialize=init)
            out, lam_g, hess, ntr, flags = eng.iterate(it, init, lam, st)
            assert ntr == ntr_ref and lam_g == lam_ref
            assert rel_err(out, ref) < 1e-6
        assert eng.solver_fallbacks() >= 2
        eng.close()
    # and a healthy window never needs the fallback
    eng = BAEngine(n, m)
    eng.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    eng.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    st, lam = g["states0"][0], 1e-4
    for k in range(20):
        st, lam, _, ntr, flags = eng.iterate(int(g["iters"][k]), bool(g["initialize"][k]), lam, st)
    assert eng.solver_fallbacks() == 0
    eng.close()


def test_hop_integrator_mode_vs_oracle():
    """vba_set_integrator(1): the <=100 s hop schedule of the reference's predict_gpu.  Two passes with gaps of
    945 / 555 s; compared with the oracle run with the same integrator (itself pinned to the reference function)."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_two_pass_sequence()
    win = od_pipe.prepare_window(det, orb)
    n, m = win.time_idx.size, win.ii.size
    eng = BAEngine(n, m)
    eng.set_integrator(True)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    st, lam = od_pipe.initial_guess(win), 1e-4
    ref, lam_ref = st.copy(), lam
    for it, init in ((0, True), (1, True), (2, True), (3, True), (10, False), (11, False), (12, False)):
        ref, lam_ref, hess_ref, ntr_ref = O.ba_iteration(it, ref, win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii,
                                                         win.time_idx, win.intrinsics, win.confidences, lam_ref,
                                                         initialize=init, hop=True)
        st, lam, hess, ntr, flags = eng.iterate(it, init, lam, st)
        assert ntr == ntr_ref and lam == lam_ref
        assert rel_err(st, ref) < 1e-7
    # the state-transition blocks of the last call against the oracle's (same integrator)
    x_in = eng.debug("dpose")          # exercised for shape only
    assert x_in.shape == (n, 9)
    eng.close()


def test_c5_full_orbit_window_vs_oracle():
    """BASELINE config 5: 2000 poses / 500 000 observations (3 s stride, ~1 orbit), all 20 calls of the driver's schedule
    chained.  The reference cannot run this size (dense (9n)^2 objects, >100 GB); parity is against the oracle, which
    is pinned to the reference up to 500 poses -- including a 500-pose sub-window of this very orbit, next test."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence("C5")
    win = od_pipe.prepare_window(det, orb)
    n, m = win.time_idx.size, win.ii.size
    assert (n, m) == (2004, 500000)      # 2000 frames + 4 knot poses (2000, 3000, 5000, 6000 s are not frame times)
    eng = BAEngine(n, m)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    st, lam = od_pipe.initial_guess(win), 1e-4
    ref, lam_ref = st.copy(), lam
    for it in range(20):
        init = it < 10
        ref, lam_ref, hess_ref, ntr_ref = O.ba_iteration(it, ref, win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii,
                                                         win.time_idx, win.intrinsics, win.confidences, lam_ref, initialize=init)
        st, lam, hess, ntr, flags = eng.iterate(it, init, lam, st)
        assert ntr == ntr_ref and lam == lam_ref and flags == 0, it
        assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6, it
        assert rel_err(st, ref) < 1e-6, it
    assert eng.solver_fallbacks() == 0
    eng.close()


def test_c5_subwindow_of_500_poses_vs_reference_states():
    """the first 500 poses of the C5 orbit (3 s stride, 250 rows per pose = 125 000 rows) is the largest
    piece of config 5 the reference can run; its 20 calls (tests/golden/c5s.npz, made by the reference's driver) pin the
    GPU path -- and, in tests/test_oracle_golden.py, the oracle the full C5 window is checked against."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    g = load_golden("c5s")
    det, orb = synth.make_subwindow("C5", 500)
    win = od_pipe.prepare_window(det, orb)
    for key, arr in (("landmarks", win.landmarks_uv), ("landmarks_xyz", win.landmarks_xyz), ("confidences", win.confidences),
                     ("intrinsics", win.intrinsics), ("cumrot_last", win.cumrot_last)):
        assert rel_err(_digest(arr), g["digest_" + key]) < 1e-12, key
    assert np.array_equal(win.time_idx, g["in_time_idx"])
    assert np.array_equal(np.array([win.ii.size, win.ii.sum(), win.ii[0], win.ii[-1]]), g["in_ii_digest"])
    n, m = win.time_idx.shape[0], win.ii.shape[0]
    eng = BAEngine(n, m)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    st, lam = g["states0"][0], 1e-4
    for k in range(20):
        st, lam, _, ntr, flags = eng.iterate(int(g["iters"][k]), bool(g["initialize"][k]), lam, st)
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k] and flags == 0, k
        if f"states_out_{k}" in g:
            ref = g[f"states_out_{k}"][0]
            assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6, k
            q, qr = st[:, 3:7], ref[:, 3:7]
            assert (2 * np.arccos(np.clip(np.abs((q * qr).sum(-1)), 0, 1))).max() < 1e-6, k
            assert rel_err(st, ref) < 1e-6, k
    eng.close()


@pytest.mark.parametrize("solver", [-1, 0], ids=["default", "sequential"])
def test_plain_BA_with_rejected_trials_vs_reference(solver):
    """Plain ``BA`` whose LM loop rejects trials (BA_filtering.py:52-77), pinned to the REFERENCE (not only to the oracle):
    the 100-pose / 5k window with confidences of 3 run through the reference's driver (tests/golden/rej.npz; 1 to 9 trials
    per call, two lamda exhaustions).  Every call from the reference's own input states: trial count and lamda exact,
    states; for the calls captured in full the LAST trial's system and solution."""
    from vinsat_amd.engine import BAEngine
    g = load_golden("rej")
    inp = golden_inputs(g)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    eng = BAEngine(n, m)
    eng.set_solver(solver)
    eng.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    eng.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    assert g["n_trials"].max() == 9 and sorted(set(g["n_trials"].tolist())) == [1, 2, 3, 4, 5, 6, 9]
    for k in range(20):
        st_in = g[f"states_in_{k}"][0] if f"states_in_{k}" in g else (g["states0"][0] if k == 0 else g[f"states_out_{k-1}"][0])
        out, lam, hess, ntr, flags = eng.iterate(int(g["iters"][k]), bool(g["initialize"][k]), float(g["lamda_in"][k]), st_in)
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k], k
        # "lamda too large" (BA_filtering.py:75-77) only when the damping really ran out: lamda_in * 10^trials > 1e4
        assert flags in (0, 1) and (flags == 0 or float(g["lamda_in"][k]) * 10.0 ** ntr > 1e4), (k, flags)
        ref = g[f"states_out_{k}"][0]
        assert np.abs(out[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-8, k
        assert rel_err(out, ref) < 1e-7, k
        assert rel_err(hess, g[f"last_hessian_{k}"][0]) < 1e-10, k
        if f"A_bands_{k}" in g:
            A = eng.debug("bands")
            sc = eng.debug("scalars")
            A[:, 1] += sc[4] * np.eye(9)
            assert sc[4] == float(np.float32(g["lamda_in"][k] * 10.0 ** (ntr - 1)))     # fp32 damping of the last trial
            assert rel_err(A, g[f"A_bands_{k}"][-1]) < 1e-11, k
            assert rel_err(eng.debug("rhs"), g[f"JTr_{k}"][0].reshape(n, 9)) < 1e-9, k
            assert rel_err(eng.debug("dpose"), g[f"dpose_{k}"][-1].reshape(n, 9)) < DPOSE_TOL, k
    # and chained on its own outputs (the BASELINE bar)
    st, lam = g["states0"][0], 1e-4
    for k in range(20):
        st, lam, _, ntr, flags = eng.iterate(int(g["iters"][k]), bool(g["initialize"][k]), lam, st)
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k], k
    ref = g["states_out_19"][0]
    assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6
    assert (2 * np.arccos(np.clip(np.abs((st[:, 3:7] * ref[:, 3:7]).sum(-1)), 0, 1))).max() < 1e-6
    eng.close()


def test_run_schedule_equals_step_by_step(c2):
    """vba_run_schedule chains the 20 calls on the device; it must give the same bits as 20 vba_step calls --
    also when some windows need several LM trials (confidence 3 -> lamda exhaustion) or the pivoted fallback
    (negative confidences) in the middle of the chain while others sail through."""
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    confs = [inp["conf"], np.full_like(inp["conf"], 3.0), np.where(inp["ii"] % 3 == 0, -0.5, inp["conf"]), inp["conf"] * 0.9]
    iters = list(range(20))
    inits = [k < 10 for k in range(20)]

    def make(W):
        e = BAEngine(n, m, windows=W)
        e.set_accumulate_lanes(8)
        return e

    # reference: one engine per window, stepped call by call
    ref = []
    for conf in confs:
        e = make(1)
        e.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
        e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
        e.set_states(g["states0"][0], 1e-4)
        for it, init in zip(iters, inits):
            e.step(it, init)
        ref.append(e.get_states())
        e.close()
    # all four windows in one handle, one chained call
    e = make(4)
    for w, conf in enumerate(confs):
        e.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n, window=w)
        e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=w)
        e.set_states(g["states0"][0], 1e-4, window=w)
    trials = e.run_schedule(iters, inits)
    assert trials > 20
    for w in range(4):
        s, lam, hess, ntr, flags = e.get_states(window=w)
        assert np.array_equal(s, ref[w][0]) and lam == ref[w][1] and np.array_equal(hess, ref[w][2]), w
        assert ntr == ref[w][3] and flags == ref[w][4], w     # a window stalled at a later call than another one must not run that call twice
    assert rel_err(e.get_states(window=0)[0], g["states_out_19"][0]) < 1e-7
    # a second run on the same handle (state of the call counters is reset)
    for w in range(4):
        e.set_states(g["states0"][0], 1e-4, window=w)
    e.run_schedule(iters, inits)
    assert np.array_equal(e.get_states(window=3)[0], ref[3][0])
    e.close()


@pytest.mark.parametrize("seed", range(6))
def test_randomised_windows_with_long_gaps_vs_oracle(seed):
    """The same with up to four gaps of 65 .. 1300 s among the random ones: the long edges go parallel in time."""
    test_randomised_windows_vs_oracle(seed, long_gaps=True)


@pytest.mark.parametrize("seed", range(8))
def test_randomised_windows_vs_oracle(seed, long_gaps=False):
    """Random window shapes (2..70 poses, 0..60 rows per pose, gaps 1..60 s, confidences 0.3..1.2, shuffled rows),
    random solver settings and a random call schedule, against the oracle."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe
    import random_windows
    win, xyz, uv, ii, conf, t, _ = random_windows.make(seed, long_gaps=long_gaps)
    n = win.time_idx.size
    eng = BAEngine(n, ii.size)
    mode = seed % 4
    if mode == 1:
        eng.set_solver(0)
    elif mode == 2 and n >= 6:
        eng.set_solver(3, 2)
    elif mode == 3:
        eng.set_pivoting(True)
    eng.upload_observations(xyz, uv, conf, ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, t)
    args = (win.cumrot_last, uv, xyz, ii, t, win.intrinsics, conf)
    st = od_pipe.initial_guess(win, seed=seed)
    lam = 1e-4
    sched = random_windows.SCHEDULE
    ref, lam_ref = st.copy(), lam
    for it, init in sched:
        # every call starts from the oracle's state, so each comparison is like for like (ill-conditioned random
        # windows amplify a 1e-9 state difference into 1e-6 differences of the next call's blocks)
        out, lam_g, hess, ntr, flags = eng.iterate(it, init, lam_ref, ref)
        ref, lam_ref, hess_ref, ntr_ref = O.ba_iteration(it, ref, *args, lam_ref, initialize=init)
        assert ntr == ntr_ref and lam_g == lam_ref, (it, ntr, ntr_ref)
        assert rel_err(out, ref) < 1e-6, it
        assert rel_err(hess, hess_ref) < 1e-8
    # step by step on the device's own states ...
    eng.set_states(od_pipe.initial_guess(win, seed=seed), 1e-4)
    for it, init in sched:
        eng.step(it, init)
    st = eng.get_states()[0]
    assert rel_err(st, ref) < 1e-5
    # ... and the same calls chained on the device give the same bits
    eng.set_states(od_pipe.initial_guess(win, seed=seed), 1e-4)
    eng.run_schedule([c[0] for c in sched], [c[1] for c in sched])
    assert np.array_equal(eng.get_states()[0], st)
    eng.close()


def test_driver_with_chained_window_calls_matches_call_by_call():
    """streaming_version's default path issues the 20 calls of a batch as one chained device call (BA_window); the
    errors must equal the call-by-call path (forced by passing a record list) bit for bit."""
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_two_pass_sequence()
    rec = []
    e1, f1, t1 = od_pipe.streaming_version(detections=det, orbit_np=orb, record=rec)      # call by call
    e2, f2, t2 = od_pipe.streaming_version(detections=det, orbit_np=orb)                  # chained
    assert np.array_equal(e1.numpy(), e2.numpy()) and f1 == f2
    assert all(np.array_equal(a, b) for a, b in zip(t1, t2))


# ------------------------------------------------------------------------------------------------ carried keys
def _carry_engine(inp, n, m, conf, carry, windows=1):
    from vinsat_amd.engine import BAEngine
    e = BAEngine(n, m, windows=windows)
    e.set_accumulate_lanes(8)
    e.set_key_carry(carry)
    for w in range(windows):
        e.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n, window=w)
        e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=w)
    return e


@pytest.mark.parametrize("confmode", ["golden", "rejections", "pivot-fallback"])
def test_carried_keys_give_the_bits_of_recomputed_keys(c2, confmode):
    """An accepted trial leaves the next call's |r| keys, exponent histogram and sum |r| behind (k_trial<true>);
    a call that starts from them must produce the bits of a call that recomputes them (k_obs_residual), call by
    call, also through rejected trials, lamda exhaustion and the pivoted repeat."""
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    conf = {"golden": inp["conf"], "rejections": np.full_like(inp["conf"], 3.0),
            "pivot-fallback": np.where(inp["ii"] % 3 == 0, -0.5, inp["conf"])}[confmode]
    outs = {}
    for carry in (False, True):
        e = _carry_engine(inp, n, m, conf, carry)
        e.set_states(g["states0"][0], 1e-4)
        seq = []
        for k in range(20):
            e.step(k, k < 10)
            s, lam, hess, ntr, flags = e.get_states()
            seq.append((s.copy(), lam, hess.copy(), ntr, flags))
        outs[carry] = seq
        e.close()
    for k, (a, b) in enumerate(zip(outs[False], outs[True])):
        assert np.array_equal(a[0], b[0]), k
        assert a[1] == b[1] and np.array_equal(a[2], b[2]) and a[3] == b[3] and a[4] == b[4], k
    if confmode == "golden":
        assert rel_err(outs[True][-1][0], g["states_out_19"][0]) < 1e-7
    if confmode == "rejections":
        assert max(o[3] for o in outs[True]) > 1


def test_carried_keys_are_dropped_when_the_states_are_replaced(c2):
    """vba_set_states / uploads / vba_iterate between two calls invalidate what the last trial left behind
    (including its half-consumed histogram); the following call must equal a call on a fresh handle."""
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    fresh = _carry_engine(inp, n, m, inp["conf"], False)
    e = _carry_engine(inp, n, m, inp["conf"], True)

    def fresh_call(st, lam, it, init):
        fresh.set_states(st, lam)
        fresh.step(it, init)
        return fresh.get_states()

    e.set_states(g["states0"][0], 1e-4)
    e.step(0, True)
    e.step(1, True)                                  # carried
    st, lam = g["states_out_9"][0], 1e-3
    e.set_states(st, lam)                            # drops the carry, leaves a dirty histogram
    e.step(10, False)
    a, b = e.get_states(), fresh_call(st, lam, 10, False)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[3] == b[3]
    e.step(11, False)                                # carried again
    b2 = (fresh.step(11, False), fresh.get_states())[1]
    assert np.array_equal(e.get_states()[0], b2[0])
    # host round-trip call in the middle (does not emit), then stepping again
    out, lam_o, hess, ntr, flags = e.iterate(3, True, 1e-4, g["states_out_2"][0])
    ref = fresh_call(g["states_out_2"][0], 1e-4, 3, True)
    assert np.array_equal(out, ref[0]) and lam_o == ref[1]
    e.step(4, True)
    fresh.step(4, True)
    assert np.array_equal(e.get_states()[0], fresh.get_states()[0])
    # new observation rows: carry dropped
    conf2 = inp["conf"] * 0.8
    e.upload_observations(inp["xyz"], inp["uv"], conf2, inp["ii"], n)
    fresh.upload_observations(inp["xyz"], inp["uv"], conf2, inp["ii"], n)
    e.step(5, True)
    fresh.step(5, True)
    assert np.array_equal(e.get_states()[0], fresh.get_states()[0])
    # chained schedule after single steps, and single steps after a schedule
    e.run_schedule(list(range(6, 12)), [k < 10 for k in range(6, 12)])
    for k in range(6, 12):
        fresh.step(k, k < 10)
    assert np.array_equal(e.get_states()[0], fresh.get_states()[0])
    e.step(12, False)
    fresh.step(12, False)
    assert np.array_equal(e.get_states()[0], fresh.get_states()[0])
    e.close()
    fresh.close()


def test_carried_keys_in_a_batch_with_stalling_windows(c2):
    """Four windows in one handle, some needing several trials: schedule with carried keys == without."""
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    confs = [inp["conf"], np.full_like(inp["conf"], 3.0), np.where(inp["ii"] % 3 == 0, -0.5, inp["conf"]), inp["conf"] * 0.9]
    iters, inits = list(range(20)), [k < 10 for k in range(20)]
    res = {}
    for carry in (False, True):
        e = _carry_engine(inp, n, m, inp["conf"], carry, windows=4)
        for w, conf in enumerate(confs):
            e.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n, window=w)
            e.set_states(g["states0"][0], 1e-4, window=w)
        e.run_schedule(iters[:7], inits[:7])
        e.run_schedule(iters[7:], inits[7:])          # second chain starts from carried keys
        res[carry] = [e.get_states(window=w) for w in range(4)]
        e.close()
    for w in range(4):
        a, b = res[False][w], res[True][w]
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2]), w


# ------------------------------------------------------------------------------------------------ BA_reg
@pytest.mark.parametrize("base", ["c1", "c2"])
@pytest.mark.parametrize("solver", [-1, 0], ids=["default", "sequential"])
def test_BA_reg_every_call_vs_reference(base, solver):
    """The reference's ``BA_reg`` (BA_filtering.py:100-210) call by call against fixtures captured from the reference
    function itself: states, damping, last Hessian block, number of LM trials (lamda exhaustion on most full calls:
    the trial mean carries the constant 100 per pose), and on three calls the matrix and right-hand side handed to
    ``torch.linalg.solve``."""
    from vinsat_amd.engine import BAEngine
    g, b = load_golden("reg_" + base), load_golden(base)
    inp = golden_inputs(b)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    e = BAEngine(n, m)
    e.set_solver(solver)
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    e.upload_prior(g["states_prior"][0], g["hessian_state_t"][0])
    e.set_prior(True)
    for k in range(len(g["iters"])):
        out, lam, hess, ntr, flags = e.iterate(int(g["iters"][k]), bool(g["initialize"][k]), float(g["lamda_in"][k]), g[f"states_in_{k}"][0])
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k], k
        assert (flags & 1) == (1 if ntr == 9 or (ntr > 1 and lam == 0.1) else 0), k
        ref = g[f"states_out_{k}"][0]
        assert rel_err(out, ref) < 1e-8, k
        assert rel_err(hess, g[f"last_hessian_{k}"][0]) < 1e-9, k
        if f"A_bands_{k}" in g:
            A = e.debug("bands").reshape(n, 3, 9, 9).copy()
            ref_A = g[f"A_bands_{k}"][0].copy()
            lam32 = float(np.float32(g["lamda_in"][k]))
            for i in range(n):
                ref_A[i, 1] -= lam32 * np.eye(9)          # the fixture holds the damped matrix of the first trial
            assert rel_err(A, ref_A) < 1e-10, k
            assert rel_err(e.debug("rhs").reshape(-1), g[f"JTr_{k}"][0].reshape(-1)) < 1e-9, k
    # the same handle goes back to plain BA
    e.set_prior(False)
    k = 10
    out, lam, hess, ntr, flags = e.iterate(int(b["iters"][k]), bool(b["initialize"][k]), float(b["lamda_in"][k]), b[f"states_out_{k-1}"][0])
    assert rel_err(out, b[f"states_out_{k}"][0]) < 1e-8 and ntr == b["n_trials"][k]
    e.close()


def test_BA_reg_call_surface_and_chained_schedule(c1):
    """``vinsat_amd.ba.BA_reg`` keeps the reference's positional signature; a 20-call chain on the device
    (``run_schedule`` with the prior switched on) equals the call-by-call results."""
    import torch
    from vinsat_amd.ba import BA_reg
    from vinsat_amd.engine import BAEngine
    g, b = load_golden("reg_c1"), c1
    inp = golden_inputs(b)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    imu = np.zeros(tuple(b["in_imu_shape"]))
    imu[0, :, -1, 6:10] = inp["cumrot"]
    k = 12
    res = BA_reg(int(g["iters"][k]), torch.tensor(g[f"states_in_{k}"]), torch.tensor(b["in_velocities"]), torch.tensor(g["states_prior"]),
                 torch.tensor(g["velocity_prior"]), torch.tensor(g["hessian_state_t"]), torch.tensor(g["hessian_rot_t"]), torch.tensor(imu),
                 torch.tensor(b["in_landmarks"]), torch.tensor(b["in_landmarks_xyz"]), inp["ii"], inp["time_idx"],
                 torch.tensor(b["in_intrinsics"]), torch.tensor(inp["conf"]), None, None, float(g["lamda_in"][k]),
                 torch.tensor(b["in_poses_gt_eci"]), initialize=False)
    assert res[0].shape == (1, n, 10) and res[3].shape == (1, 9, 9) and res[2] == g["lamda_out"][k]
    assert rel_err(res[0][0].numpy(), g[f"states_out_{k}"][0]) < 1e-8
    e = BAEngine(n, m)
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    e.upload_prior(g["states_prior"][0], g["hessian_state_t"][0])
    e.set_prior(True)
    e.set_states(g["states0"][0], 1e-4)
    trials = e.run_schedule(list(range(20)), [k < 10 for k in range(20)])
    assert trials >= int(g["n_trials"].sum())       # enqueued trials: re-issued speculative ones count too
    s, lam, hess, ntr, flags = e.get_states()
    assert rel_err(s, g["states_out_19"][0]) < 1e-7 and lam == g["lamda_out"][19]
    e.close()


def test_BA_reg_with_a_propagated_prior_vs_oracle(c2):
    """Prior made the way the reference means it (``propagate_dynamics_cov_init`` from a last Hessian block: information
    matrices of condition ~1e4 that tighten along the window), sampled at the window's frames, through two full
    ``BA_reg`` calls on the GPU against the oracle."""
    from vinsat_amd import prior
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    t = inp["time_idx"]
    dur = int(t[-1] - t[0])
    rng = np.random.default_rng(3)
    omega = rng.normal(0, 1e-3, (dur + 1, 3))
    ref0 = g["states_out_19"][0]
    st_t, v_t, Hs_t, Hr_t = prior.propagate_dynamics_cov_init(ref0[0], ref0[0, 7:], g["last_hessian_19"][0], omega, 0, dur, 1)
    rows = (t - t[0]).astype(int)
    sp, Hs = st_t[rows], Hs_t[rows]
    e = BAEngine(n, m)
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    e.upload_prior(sp, Hs)
    e.set_prior(True)
    st, lam = g["states_out_9"][0], 1e-4
    for it in (10, 11):
        out, lam_g, hess, ntr, flags = e.iterate(it, False, lam, st)
        ref, lam_o, hess_o, ntr_o = O.ba_iteration(it, st, inp["cumrot"], inp["uv"], inp["xyz"], inp["ii"], inp["time_idx"], inp["K"],
                                                   inp["conf"], lam, initialize=False, prior=(sp, Hs))
        assert ntr == ntr_o and lam_g == lam_o
        assert rel_err(out, ref) < 1e-7 and rel_err(hess, hess_o) < 1e-9
        st, lam = out, lam_g
    e.close()


def test_BA_reg_batched_windows_equal_single_window_runs(c2):
    """Three windows with different priors in one handle (prior on), stepped together through rejected trials and
    lamda exhaustion, against one handle per window: same bits."""
    from vinsat_amd.engine import BAEngine
    g = load_golden("reg_c2")
    inp = golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    rng = np.random.default_rng(0)
    priors = []
    for w in range(3):
        sp = g["states_prior"][0].copy()
        sp[:, :3] += rng.normal(0, 0.02 * w, (n, 3))
        priors.append((sp, g["hessian_state_t"][0] * (1.0 + 0.5 * w)))
    calls = [(9, True), (10, False), (11, False), (12, False)]

    def run(ws, W):
        e = BAEngine(n, m, windows=W)
        e.set_accumulate_lanes(8)
        for k, w in enumerate(ws):
            e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n, window=k)
            e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=k)
            e.upload_prior(*priors[w], window=k)
            e.set_states(g["states_in_9"][0], 1e-4, window=k)
        e.set_prior(True)
        for it, init in calls:
            e.step(it, init)
        out = [e.get_states(window=k) for k in range(W)]
        e.close()
        return out

    batch = run([0, 1, 2], 3)
    for w in range(3):
        single = run([w], 1)[0]
        assert np.array_equal(batch[w][0], single[0]) and batch[w][1] == single[1] and batch[w][3] == single[3], w
    assert max(b[3] for b in batch) > 1


@pytest.mark.parametrize("chunk", [-1, 3, 2])
def test_windows_of_different_length_pick_their_own_reduction_variant(chunk):
    """Three windows of 100, 60 and 20 poses in one handle: with chunks of 3 their separator counts (33, 19, 6) fall on
    both sides of the size from which the first cyclic-reduction level runs as its own kernel, so every solve kernel of
    the family is launched and each window must be taken by exactly one.  Against single-window handles, bit for bit."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    wins = []
    for k, npose in enumerate((100, 60, 20)):
        det, orb = synth.make_sequence(synth.WindowConfig(f"w{k}", npose, 30, 5), seed=40 + k)
        wins.append(od_pipe.prepare_window(det, orb))
    n_max = max(w.time_idx.size for w in wins)
    m_max = max(w.ii.size for w in wins)
    sched = [(9, True), (10, False), (11, False), (14, False)]

    def run(idx):
        e = BAEngine(n_max, m_max, windows=len(idx))
        e.set_accumulate_lanes(8)
        if chunk > 0:
            e.set_solver(chunk, -1)
        for k, i in enumerate(idx):
            w = wins[i]
            e.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, w.time_idx.size, window=k)
            e.upload_window(w.intrinsics, w.cumrot_last, w.time_idx, window=k)
            e.set_states(od_pipe.initial_guess(w, seed=i), 1e-4, window=k)
        for it, init in sched:
            e.step(it, init)
        out = [e.get_states(window=k) for k in range(len(idx))]
        e.close()
        return out

    batch = run([0, 1, 2])
    for i in range(3):
        single = run([i])[0]
        assert np.array_equal(batch[i][0], single[0]) and batch[i][1] == single[1], i
        assert np.isfinite(batch[i][0]).all()


# ------------------------------------------------------------------------------------------------ warm select / folded accept test
def _schedule_states(eng, st0, iters, inits, chained):
    eng.set_states(st0, 1e-4)
    if chained:
        eng.run_schedule(iters, inits)
    else:
        for it, init in zip(iters, inits):
            eng.step(it, init)
    return eng.get_states()


@pytest.mark.parametrize("confmode", ["golden", "rejections"])
def test_warm_select_and_folded_accept_test_give_the_bits_of_the_exact_path(c2, confmode):
    """Carried keys are selected from the bucket of ONE warm bin in the prologue of the accumulation (or, setting 3, by one
    warm pass of a select kernel), which in a chained schedule also evaluates the accept test of the call in front.  The median stays exact and the accept test is the same arithmetic, so: chained / stepped, warm / exact
    digits, and a warm select forced to miss on every call (the repeat path) all end in identical bits -- also when calls
    reject trials and exhaust lamda in the middle of the chain."""
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    conf = inp["conf"] if confmode == "golden" else np.full_like(inp["conf"], 3.0)
    iters, inits = list(range(20)), [k < 10 for k in range(20)]
    outs = {}
    for name, warm, chained in (("exact-stepped", 0, False), ("exact-chained", 0, True), ("warm-stepped", 1, False),
                                ("warm-chained", 1, True), ("miss-stepped", 2, False), ("miss-chained", 2, True),
                                ("kernel-stepped", 3, False), ("kernel-chained", 3, True)):      # 3: warm select as its own kernel
        e = BAEngine(n, m)
        e.set_warm_select(warm)
        e.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
        e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
        outs[name] = _schedule_states(e, g["states0"][0], iters, inits, chained)
        misses = e.warm_select_misses()
        assert misses == 0 if warm == 0 else (misses >= 19 if warm == 2 else misses <= 2), (name, misses)   # a real miss is rare
        e.close()
    ref = outs["exact-stepped"]
    for name, o in outs.items():
        assert np.array_equal(o[0], ref[0]) and o[1] == ref[1] and o[3] == ref[3] and o[4] == ref[4], name
    if confmode == "golden":
        assert rel_err(ref[0], g["states_out_19"][0]) < 1e-7


def test_warm_select_with_long_bins_and_ties():
    """Keys that pile up in one warm bin (half of all residuals identical): the list is longer than what is ranked by
    counting and goes through the radix digits of the in-bin offset; the median must still be the exact lower median."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    cfg = synth.WindowConfig("ties", 24, 400, 5)
    det, orb = synth.make_sequence(cfg, seed=5, pixel_noise=0.0)
    win = od_pipe.prepare_window(det, orb)
    uv = win.landmarks_uv.copy()
    uv[::2] += 0.75                    # every second row is off by exactly the same amount in both components
    n, m = win.time_idx.size, win.ii.size
    st = win.states_gt.copy()
    outs = []
    for warm in (0, 1):
        e = BAEngine(n, m)
        e.set_warm_select(warm)
        e.upload_observations(win.landmarks_xyz, uv, win.confidences, win.ii, n)
        e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
        e.set_states(st, 1e-4)
        meds = []
        for it in (5, 6, 7, 8):
            e.step(it, True)
            sc = e.debug("scalars")
            a = np.abs(uv - e.debug("est")).reshape(-1)
            assert sc[0] == np.sort(a)[(a.size - 1) // 2]       # exact lower median of the device's own residuals
            meds.append(sc[0])
        outs.append((e.get_states(), meds))
        e.close()
    assert np.array_equal(outs[0][0][0], outs[1][0][0]) and outs[0][1] == outs[1][1]


def test_many_windows_per_launch_agree_with_single_window_runs():
    """From 16 windows per handle on, the bandwidth-mode kernels take over (assembly through memory, one wave per chain,
    dynamics on a second stream): 20 windows of different data, chained 20 calls, against one-window handles (latency-mode
    kernels) -- trial counts and lamda exact, states to rounding."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    cfg = synth.WindowConfig("w20", 40, 30, 5)
    wins = [od_pipe.prepare_window(*synth.make_sequence(cfg, seed=s)) for s in range(20)]
    n, m = wins[0].time_idx.size, wins[0].ii.size
    iters, inits = list(range(20)), [k < 10 for k in range(20)]
    big = BAEngine(n, m, windows=len(wins))
    for k, w in enumerate(wins):
        big.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, n, window=k)
        big.upload_window(w.intrinsics, w.cumrot_last, w.time_idx, window=k)
        big.set_states(od_pipe.initial_guess(w, seed=k), 1e-4, window=k)
    big.run_schedule(iters, inits)
    for k, w in enumerate(wins):
        e = BAEngine(n, m)
        e.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, n)
        e.upload_window(w.intrinsics, w.cumrot_last, w.time_idx)
        ref = _schedule_states(e, od_pipe.initial_guess(w, seed=k), iters, inits, True)
        e.close()
        got = big.get_states(window=k)
        assert got[3] == ref[3] and got[1] == ref[1] and got[4] == ref[4], k
        # different reduction trees (lanes per pose, one wave per chain): rounding differences grow with the conditioning
        # of these small windows over 20 calls; the bar is the BASELINE one
        assert np.abs(got[0][:, :3] - ref[0][:, :3]).max() / np.abs(ref[0][:, :3]).max() < 1e-8, k
        assert rel_err(got[0], ref[0]) < 1e-6, k
    big.close()


@pytest.mark.parametrize("mask", [1, 2, 3])
def test_latency_mode_fusions_vs_reference(c2, mask):
    """VBA_OPT_FUSION: the trial kernel forming the step itself (bit 0) and the chunk elimination forming its own blocks
    (bit 1).  Off by default (slower on MI355X); same arithmetic in another place, so: every call of the C2 chain against the
    reference's states, trial counts and lamda exact, and the rejection window against the unfused run."""
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    iters, inits = list(range(20)), [k < 10 for k in range(20)]
    e = BAEngine(n, m)
    e.set_fusion(mask)
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    st, lam = g["states0"][0], 1e-4
    for k in range(20):
        st, lam, hess, ntr, flags = e.iterate(iters[k], inits[k], lam, st)
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k] and flags == 0
        assert rel_err(st, g[f"states_out_{k}"][0]) < 1e-7
        assert rel_err(hess, g[f"last_hessian_{k}"][0]) < 1e-7
        if k in (0, 9, 10, 19):
            A = e.debug("bands")
            A[:, 1] += e.debug("scalars")[4] * np.eye(9)
            assert rel_err(A, g[f"A_bands_{k}"][0]) < 1e-10
            assert rel_err(e.debug("dpose"), g[f"dpose_{k}"][0].reshape(n, 9)) < DPOSE_TOL
    chained = _schedule_states(e, g["states0"][0], iters, inits, True)
    assert chained[3] == 1 and rel_err(chained[0], g["states_out_19"][0]) < 1e-7
    e.close()
    # rejections, lamda exhaustion and the pivoted fallback with the fusions on
    for conf in (np.full_like(inp["conf"], 3.0), np.where(inp["ii"] % 3 == 0, -0.5, inp["conf"])):
        outs = []
        for mk in (0, mask):
            e = BAEngine(n, m)
            e.set_fusion(mk)
            e.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
            e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
            outs.append(_schedule_states(e, g["states0"][0], iters, inits, True))
            e.close()
        assert outs[0][3] == outs[1][3] and outs[0][1] == outs[1][1] and outs[0][4] == outs[1][4]
        assert rel_err(outs[1][0], outs[0][0]) < 1e-7


def test_torch_cuda_initialises_after_the_library_has_used_the_gpu():
    """Load order: the library first (numpy call surface), torch.cuda afterwards, in a fresh process."""
    import subprocess, sys
    code = ("import numpy as np; from vinsat_amd.engine import BAEngine; e = BAEngine(4, 8); e.close(); "
            "import torch; x = torch.zeros(3, device='cuda'); print(int(x.sum().item()))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "0"


@pytest.mark.parametrize("solver", [0], ids=["four-per-wave"])
@pytest.mark.parametrize("reg", [False, True])
def test_batched_walk_forming_its_own_blocks_gives_the_bits_of_the_assembled_path(c2, reg, solver):
    """VBA_OPT_FUSION bit 2 (default for 16 windows and more, sequential driver -- itself the default from 128 windows on):
    in the full phase the sequential solve (k_solve_quad, four windows per wavefront; k_solve_forming with one) forms each block
    from the per-pose inputs itself and the assembly launch is gone.  Same entries, same elimination: 16 windows -- the
    golden one, one that rejects trials and exhausts lamda, one whose blocks send the unpivoted path to the pivoted
    kernels, perturbed copies -- through the chained schedule, bit for bit against the assembled path; plain BA and
    BA_reg (prior staged with the inputs)."""
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    gr = load_golden("reg_c2")
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    W = 16
    rng = np.random.default_rng(3)
    confs = [inp["conf"], np.full_like(inp["conf"], 3.0), np.where(inp["ii"] % 3 == 0, -0.5, inp["conf"])]
    confs += [inp["conf"] * rng.uniform(0.5, 1.5, m) for _ in range(W - 3)]
    iters, inits = list(range(20)), [k < 10 for k in range(20)]
    outs, dbg = [], []
    for mask in (1, 5):
        e = BAEngine(n, m, windows=W, mode=0)
        e.set_fusion(mask)
        e.set_solver(solver)
        for k in range(W):
            e.upload_observations(inp["xyz"], inp["uv"], confs[k], inp["ii"], n, window=k)
            e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=k)
            if reg:
                e.upload_prior(gr["states_prior"][0], gr["hessian_state_t"][0] * (1.0 + 0.1 * k), window=k)
            e.set_states(g["states0"][0], 1e-4, window=k)
        e.set_prior(reg)
        e.run_schedule(iters, inits)
        outs.append([e.get_states(window=k) for k in range(W)])
        dbg.append((e.debug("bands", window=0), e.debug("dpose", window=0)))
        e.close()
    for k in range(W):
        a, b = outs[0][k], outs[1][k]
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[3] == b[3] and a[4] == b[4], k
        assert np.array_equal(a[2], b[2]), k            # last_hessian: written by the walk itself
    assert np.array_equal(dbg[0][0], dbg[1][0]) and np.array_equal(dbg[0][1], dbg[1][1])
    if not reg:
        assert rel_err(outs[1][0][0], g["states_out_19"][0]) < 1e-7
        assert outs[1][1][3] > 1                         # the rejection window really rejected


@pytest.mark.parametrize("chunk", [4, 5, 8, 13])
@pytest.mark.parametrize("pivot", [False, True])
def test_two_sided_chunk_elimination_vs_one_wave(c2, chunk, pivot):
    """VBA_OPT_CHUNK_WAVES: chunks eliminated from both ends by two waves that meet in the middle (default) against one
    wave walking them left to right -- every chunk length the sizes produce (even, odd, the short last chunk that falls
    back to one wave), unpivoted and pivoted blocks, plain BA and the rejection window.  Same system, another
    elimination order: the step agrees to rounding, trial counts and lamda exactly, and each order repeats its own bits."""
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    iters, inits = list(range(20)), [k < 10 for k in range(20)]
    for conf in (inp["conf"], np.full_like(inp["conf"], 3.0)):
        outs, steps = [], []
        for waves in (1, 2, 2):
            e = BAEngine(n, m)
            e.set_solver(chunk, -1)
            e.set_pivoting(pivot)
            e.set_chunk_waves(waves)
            e.upload_observations(inp["xyz"], inp["uv"], conf, inp["ii"], n)
            e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
            e.iterate(10, False, float(g["lamda_in"][10]), g["states_out_9"][0])
            steps.append(e.debug("dpose"))
            outs.append(_schedule_states(e, g["states0"][0], iters, inits, True))
            e.close()
        assert rel_err(steps[1], steps[0]) < (2e-7 if pivot else 3e-8)     # row exchanges: each order is ~1e-7 from the dense LU
        assert np.array_equal(steps[1], steps[2]) and np.array_equal(outs[1][0], outs[2][0])
        assert outs[0][3] == outs[1][3] and outs[0][1] == outs[1][1] and outs[0][4] == outs[1][4]
        assert rel_err(outs[1][0], outs[0][0]) < 1e-7
        if conf is inp["conf"]:
            assert rel_err(steps[1], g["dpose_10"][0].reshape(n, 9)) < (DPOSE_TOL_PIVOTED if pivot else DPOSE_TOL)


@pytest.mark.parametrize("reg", [False, True])
@pytest.mark.parametrize("windows", [1, 16])
def test_uniform_pass_assembly_gives_the_bits_of_the_per_entry_form(c2, reg, windows):
    """Full-phase assembly: one wave per pose in seven uniform passes (vba_asm_fast.h, VBA_OPT_FUSION bit 3) against the
    per-entry form (band_entry / rhs_entry per thread, default): every band entry, the right-hand side and the result of
    the call, bit for bit -- latency and batched geometry, plain BA and BA_reg (prior terms)."""
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    gr = load_golden("reg_c2")
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    outs = []
    for mask in (1, 9):
        e = BAEngine(n, m, windows=windows, mode=1 if windows == 1 else 0)
        e.set_fusion(mask)
        for k in range(windows):
            e.upload_observations(inp["xyz"], inp["uv"], inp["conf"] * (1.0 + 0.01 * k), inp["ii"], n, window=k)
            e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=k)
            if reg:
                e.upload_prior(gr["states_prior"][0], gr["hessian_state_t"][0] * (1.0 + 0.1 * k), window=k)
            e.set_states(g["states_out_9"][0], float(g["lamda_in"][10]), window=k)
        e.set_prior(reg)
        e.step(10, False)
        e.step(11, False)
        wl = windows - 1
        outs.append((e.debug("bands", window=wl), e.debug("rhs", window=wl), e.get_states(window=wl), e.debug("bands", window=0)))
        e.close()
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][2], b[2][2]) and a[2][1] == b[2][1]
    assert np.abs(a[0]).max() > 0 and np.abs(a[0][:, 0]).max() > 0 and np.abs(a[0][:, 2]).max() > 0     # off-diagonal bands are populated


