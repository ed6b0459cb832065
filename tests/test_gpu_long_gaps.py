"""Long gaps of the pose chain (vinsat_amd/csrc/vba_long.hip: parallel-in-time propagation, ordered product of the chunks'
transition matrices) against the oracle's serial chain of 1 s RK4 steps (reference: BA_utils.py:73-87, 457-509)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import ba_oracle as O

pytestmark = pytest.mark.gpu

D = np.array([1.0, 1.0, 1.0, 100.0, 100.0, 100.0])


def _window(n, rows_per_pose=6, seed=5):
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence(synth.WindowConfig("long", n, rows_per_pose, 5), seed=seed)
    return od_pipe.prepare_window(det, orb)


def _engine(win, t):
    from vinsat_amd.engine import BAEngine
    n = t.size
    eng = BAEngine(n, win.ii.size)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, t)
    return eng


def _check_factor(eng, st, t, cumrot, phi_tol=1e-12, it=12, lam=1e-3):
    """One full-phase call on `st`; the factor it formed at those states against the oracle's serial walk."""
    out = eng.iterate(it, False, lam, st)
    r, E, F = O.orbit_factor(st, t, jacobian=True)
    Phi = eng.debug("Phi")[:-1]
    assert rel_err((D[None, :, None] * Phi)[:, :, :3], E[:, :, 0:3]) < phi_tol
    assert rel_err((D[None, :, None] * Phi)[:, :, 3:], E[:, :, 6:9]) < phi_tol
    # per edge as well: a long edge's block must not hide behind a large short one
    for i in range(Phi.shape[0]):
        assert rel_err(D[:, None] * Phi[i], np.concatenate([E[i][:, 0:3], E[i][:, 6:9]], -1)) < 10 * phi_tol, i
    rp = eng.debug("r_pred")
    # x_hat - x_next: positions of ~7000 km, velocities x 100: 1e-9 absolute is 1.4e-13 relative of the propagated state
    assert np.abs(rp[:, :6] - r).max() < 1e-9
    return out


def test_two_pass_window_factor_against_the_serial_chain():
    """The reference's two-pass window (gaps of 935 and 510 s), at the states the reference itself reached before call 25."""
    from vinsat_amd import od_pipe, synth
    g = load_golden("gap")
    win = od_pipe.prepare_window(*synth.make_two_pass_sequence())
    t = win.time_idx
    assert sorted(np.diff(t)[np.diff(t) > 64]) == [510, 935]
    eng = _engine(win, t)
    st = g["states_out_24"][0]
    assert not g["initialize"][25]
    out = _check_factor(eng, st, t, win.cumrot_last, it=int(g["iters"][25]), lam=float(g["lamda_in"][25]))
    ref = g["states_out_25"][0]
    assert g["n_trials"][25] == out[3] and g["lamda_out"][25] == out[1]
    assert rel_err(out[0], ref) < 1e-8
    eng.close()


@pytest.mark.parametrize("gaps", [(64, 65, 66), (100, 7, 1024, 3), (1025, 2, 2000), (3000,), (729, 730, 731)],
                         ids=["threshold", "mixed", "over-1024", "3000s", "plan-edge"])
def test_gap_lengths_around_every_rule_of_the_partition(gaps):
    """64 steps stay on the serial walk, 65 are cut into chunks; 1024 / 1025: the chunk count saturates at 32; 3000 s needs
    more than one sweep; short edges beside long ones keep their arithmetic."""
    n = len(gaps) + 3
    win = _window(n)
    steps = np.array([5] + list(gaps) + [4], dtype=np.int64)
    t = np.concatenate([[10], 10 + np.cumsum(steps)]).astype(np.int64)
    eng = _engine(win, t)
    rng = np.random.default_rng(2)
    st = win.states_gt.copy()
    st[:, :3] += rng.normal(0, 20.0, size=(n, 3))
    st[:, 7:] *= 1.0 + rng.normal(0, 0.01, size=(n, 3))
    _check_factor(eng, st, t, win.cumrot_last)
    eng.close()


def test_trial_residual_and_decisions_follow_the_oracle_across_long_gaps():
    """Whole calls (accept test with the long edges' residual slots, several trials) against the oracle."""
    from vinsat_amd import od_pipe
    from test_gpu_parity import _oracle_vs_gpu
    win = _window(9, rows_per_pose=12, seed=8)
    steps = np.array([5, 300, 5, 5, 90, 5, 700, 5], dtype=np.int64)
    t = np.concatenate([[10], 10 + np.cumsum(steps)]).astype(np.int64)
    eng = _engine(win, t)
    args = (win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii, t, win.intrinsics, win.confidences)
    st = od_pipe.initial_guess(win)
    lam = 1e-4
    for it, init in ((0, True), (1, True), (10, False), (11, False), (12, False), (13, False)):
        st, lam, ntr, flags = _oracle_vs_gpu(eng, args, it, init, lam, st, tol=1e-7)
    eng.close()


def test_more_long_edges_than_slots_fall_back_to_the_serial_walk():
    """kLongCap = 64 long edges per window go parallel in time, further ones take the ordinary lanes."""
    n = 72
    win = _window(n, rows_per_pose=3)
    steps = np.full(n - 1, 70, dtype=np.int64)
    t = np.concatenate([[10], 10 + np.cumsum(steps)]).astype(np.int64)
    eng = _engine(win, t)
    _check_factor(eng, win.states_gt.copy(), t, win.cumrot_last)
    eng.close()


def test_a_window_of_a_big_handle_runs_out_of_chain_room_and_walks_the_rest():
    """Handles of more than 256 windows keep 512 chain states per window and parity: three 1000 s gaps fit (157 states each), the others take
    the ordinary serial lanes.  Same factor either way (against the oracle)."""
    from vinsat_amd.engine import BAEngine
    n = 8
    win = _window(n)
    steps = np.array([1000, 1000, 1000, 1000, 1000, 1000, 5], dtype=np.int64)
    t = np.concatenate([[10], 10 + np.cumsum(steps)]).astype(np.int64)
    eng = BAEngine(n, win.ii.size, windows=257, mode=0)
    small = _window(n, seed=6)
    for w in range(257):
        src = win if w == 0 else small
        eng.upload_observations(src.landmarks_xyz, src.landmarks_uv, src.confidences, src.ii, n, window=w)
        eng.upload_window(src.intrinsics, src.cumrot_last, t if w == 0 else src.time_idx, window=w)
        eng.set_states(src.states_gt, 1e-3, window=w)
    eng.step(12, False)
    r, E, F = O.orbit_factor(win.states_gt, t, jacobian=True)
    Phi = eng.debug("Phi")[:-1]
    for i in range(n - 1):
        assert rel_err(D[:, None] * Phi[i], np.concatenate([E[i][:, 0:3], E[i][:, 6:9]], -1)) < 1e-11, i
    assert np.abs(eng.debug("r_pred")[:, :6] - r).max() < 1e-9
    eng.close()


@pytest.mark.parametrize("mode", [1, 0], ids=["latency-kernels", "bandwidth-kernels"])
def test_long_edges_in_a_batch_have_the_bits_of_the_window_alone(mode):
    """Windows with different numbers of long edges on one handle (the extra blocks of a window without one write zeros),
    with either kernel set (dynamics_block riding in the accumulation / k_dynamics_pair on the second stream)."""
    from vinsat_amd.engine import BAEngine
    wins, ts = [], []
    for k, gaps in enumerate([(5, 5, 5), (200, 5, 400), (5, 80, 5)]):
        win = _window(4, seed=20 + k)
        t = np.concatenate([[10], 10 + np.cumsum(np.array(gaps, dtype=np.int64))]).astype(np.int64)
        wins.append(win)
        ts.append(t)
    m_max = max(w.ii.size for w in wins)
    sched = ([0, 1, 10, 11, 12], [True, True, False, False, False])

    def run(idx):
        e = BAEngine(4, m_max, windows=len(idx), mode=mode)
        for slot, k in enumerate(idx):
            w = wins[k]
            e.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, 4, window=slot)
            e.upload_window(w.intrinsics, w.cumrot_last, ts[k], window=slot)
            st = w.states_gt.copy()
            st[:, :3] += 3.0
            e.set_states(st, 1e-4, window=slot)
        e.run_schedule(*sched)
        out = [e.get_states(window=slot)[0].copy() for slot in range(len(idx))]
        e.close()
        return out

    together = run([0, 1, 2])
    for k in range(3):
        alone = run([k])[0]
        assert np.array_equal(together[k], alone), k


# ------------------------------------------------------------------------------------------------ sharded mode across long gaps
def _second_batch_window():
    """The second batch of the two-pass sequence (25 poses, gaps of 935 and 510 s) with the states the driver starts it from."""
    from vinsat_amd import od_pipe, synth
    win = od_pipe.prepare_window(*synth.make_two_pass_sequence())
    return win, od_pipe.initial_guess(win, seed=3)


def test_observation_sharded_window_with_long_gaps_on_three_emulated_ranks():
    """vba_sh_stage1..4 (the caller-dispatched protocol) on the window with the long gaps: every rank propagates the long edges
    itself (k_long_chain / k_long_tangent / k_long_finish behind the accumulation, k_long_trial behind the trial kernel), their
    residual slots are summed with the pose-chain blocks' -- against the unsharded engine, call by call."""
    from test_gpu_parity import _EmulatedRanks
    from vinsat_amd.engine import BAEngine
    win, st0 = _second_batch_window()
    n, m = win.time_idx.size, win.ii.size
    em = _EmulatedRanks(n, m, 3, win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, win.intrinsics, win.cumrot_last, win.time_idx)
    single = BAEngine(n, m)
    single.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    single.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
    em.set_states(st0, 1e-4)
    ref, lam_ref = st0, 1e-4
    for it, init in [(0, True), (1, True), (10, False), (11, False), (12, False), (13, False), (14, False)]:
        em.call(it, init)
        ref, lam_ref, hess_ref, ntr_ref, _ = single.iterate(it, init, lam_ref, ref)
        out = em.results()
        assert out[3] == ntr_ref and out[1] == lam_ref, it
        assert rel_err(out[0], ref) < 1e-9, it
    em.close()
    single.close()


def _worker_native_long(rank, world, port, tmp):
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from vinsat_amd.dist import HipStageEngine, ShardedBA
    from vinsat_amd.engine import BAEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        win, st0 = _second_batch_window()
        n, m = win.time_idx.size, win.ii.size
        iters, inits = list(range(20)), [False] * 20

        def engine():
            e = BAEngine(n, m)
            e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
            e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
            return e
        res = {}
        for name, proto, chained in (("carried-chained", 1, True), ("carried", 1, False), ("round3", 0, False)):
            stage = HipStageEngine(engine(), torch_stream=False)
            stage.attach_rccl(dist)
            stage.set_protocol(proto)
            sb = ShardedBA(stage, n, m, m)
            sb.set_states(st0, 1e-4)
            if chained:
                sb.run_schedule(iters[:7], inits[:7])
                sb.run_schedule(iters[7:], inits[7:])
            else:
                for it, init in zip(iters, inits):
                    sb.step(it, init)
            res[name] = sb.get_states()
            sb.close()
        single = engine()
        single.set_states(st0, 1e-4)
        single.run_schedule(iters, inits)
        ref = single.get_states()
        single.close()
        base = res["carried"]
        for name, r in res.items():
            assert np.array_equal(r[0], base[0]) and r[1] == base[1] and r[3] == base[3], name
        assert base[1] == ref[1] and base[3] == ref[3] and np.abs(base[0] - ref[0]).max() / np.abs(ref[0]).max() < 1e-9
        np.save(os.path.join(tmp, "ok.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_library_issued_sharded_protocols_across_long_gaps(tmp_path):
    """The carried-keys protocol (chained and stepped) and the round-3 protocol, exchanges issued by the library over RCCL (one rank: a
    one-GPU box), on the window with the long gaps: one set of bits, and the unsharded schedule to 1e-9."""
    import os
    import torch.multiprocessing as mp
    port = 29300 + (os.getpid() % 200)
    mp.spawn(_worker_native_long, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def test_BA_reg_across_long_gaps_against_the_oracle():
    """``BA_reg`` (BA_filtering.py:100-210: the per-pose prior, its constant rotation residual, the trial's attitude residual scaled as
    the reference passes its coefficients) on a window with long gaps: the long edges' orbit residual comes from kernels of their
    own, everything else from the ordinary lanes -- whole calls against the oracle."""
    from vinsat_amd.engine import BAEngine
    win = _window(7, rows_per_pose=10, seed=9)
    steps = np.array([5, 400, 5, 5, 150, 5], dtype=np.int64)
    t = np.concatenate([[10], 10 + np.cumsum(steps)]).astype(np.int64)
    n = t.size
    rng = np.random.default_rng(6)
    sp = win.states_gt.copy()
    sp[:, :3] += rng.normal(0, 0.5, (n, 3))
    Hs = np.stack([np.eye(6) * s for s in rng.uniform(0.5, 3.0, n)])
    for i in range(n):
        B = rng.normal(0, 0.1, (6, 6))
        Hs[i] += B @ B.T
    eng = BAEngine(n, win.ii.size)
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, t)
    eng.upload_prior(sp, Hs)
    eng.set_prior(True)
    st = win.states_gt.copy()
    st[:, :3] += rng.normal(0, 2.0, (n, 3))
    lam = 1e-4
    for it in (10, 11, 12):
        out, lam_g, hess, ntr, flags = eng.iterate(it, False, lam, st)
        ref, lam_o, hess_o, ntr_o = O.ba_iteration(it, st, win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii, t, win.intrinsics,
                                                   win.confidences, lam, initialize=False, prior=(sp, Hs))
        assert ntr == ntr_o and lam_g == lam_o, it
        assert rel_err(out, ref) < 1e-7, it
        st, lam = out, lam_g
    eng.close()


def test_window_of_a_sixth_pass_with_ten_long_gaps_against_the_oracle():
    """What the window of a sequence's last batch looks like after six passes: 127 poses, ten gaps of 395 .. 1000 s (a knot every
    1000 s, od_pipe.py:213-221).  The factor and whole calls against the oracle's serial chain; the driver end to end against
    the same driver with the oracle standing in for the GPU BA."""
    from test_gpu_parity import _oracle_vs_gpu
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_multi_pass_sequence()
    win = od_pipe.prepare_window(det.copy(), orb.copy())
    gaps = np.diff(win.time_idx)
    assert win.time_idx.size == 127 and (gaps > 64).sum() == 10 and gaps.max() == 1000
    eng = _engine(win, win.time_idx)
    rng = np.random.default_rng(1)
    st = win.states_gt.copy()
    st[:, :3] += rng.normal(0, 5.0, st[:, :3].shape)
    _check_factor(eng, st, win.time_idx, win.cumrot_last)
    args = (win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii, win.time_idx, win.intrinsics, win.confidences)
    lam = 1e-4
    for it in (10, 11, 12):
        st, lam, ntr, flags = _oracle_vs_gpu(eng, args, it, False, lam, st, tol=1e-7)
    eng.close()
