"""The solver variants that were measured slower and are kept for comparison only (``make -C vinsat_amd/csrc VARIANTS=1`` ->
``libvinsat_ba_variants.so``; the default library does not carry them, ``vba_has_variants()`` == 0, and this module is
deselected): three windows per wavefront (``k_solve_packed``), one window per wavefront forming its own blocks
(``k_solve_forming``), one cyclic-reduction level in front (``k_cr_level0``), the solve as one grid of waiting blocks
(``k_solve_resident``).  Each is bit-compared with the path a default handle takes.  Run (on a GPU box):

    make -C vinsat_amd/csrc VARIANTS=1 && VBA_LIB=$PWD/vinsat_amd/libvinsat_ba_variants.so python -m pytest tests/test_gpu_variants.py -m gpu -q

DESIGN.md section 4 records why each lost."""
import numpy as np
import pytest

from conftest import golden_inputs, load_golden, rel_err  # noqa: F401

pytestmark = [pytest.mark.gpu, pytest.mark.variants]


@pytest.mark.parametrize("pivot", [False, True], ids=["unpivoted", "pivoted"])
def test_packed_sequential_solve_three_windows_per_wave(pivot):
    """Sequential driver with equal pose counts: three windows share a wavefront (k_solve_packed).  Five windows
    (3 + a partial group of 2) with different data must reproduce the one-window-per-wave results bit for bit."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    cfg = synth.WindowConfig("pk", 48, 25, 5)
    wins = [od_pipe.prepare_window(*synth.make_sequence(cfg, seed=s)) for s in range(5)]
    n, m = wins[0].time_idx.size, wins[0].ii.size
    sched = [(0, True), (1, True), (10, False), (11, False), (12, False)]

    def run(solver, W):
        outs = []
        groups = [list(enumerate(wins))] if W == 5 else [[(k, w)] for k, w in enumerate(wins)]
        for grp in groups:
            e = BAEngine(n, m, windows=len(grp))
            e.set_solver(solver)
            e.set_pivoting(pivot)
            e.set_accumulate_lanes(8)
            for k, (seed, w) in enumerate(grp):
                e.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, n, window=k)
                e.upload_window(w.intrinsics, w.cumrot_last, w.time_idx, window=k)
                e.set_states(od_pipe.initial_guess(w, seed=seed), 1e-4, window=k)
            for it, init in sched:
                e.step(it, init)
            outs += [e.get_states(window=k) for k in range(len(grp))]
            e.close()
        return outs

    packed = run(-3, 5)         # 5 windows, sequential with packing forced -> k_solve_packed
    single = run(-2, 1)         # one window per handle, one window per wavefront
    for a, b in zip(packed, single):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[3] == b[3]


@pytest.mark.parametrize("chunk", [4, 5, 6, 7, 8, 13])
def test_two_cyclic_reduction_levels_in_front_give_the_bits_of_one(chunk):
    """The reduced system's first TWO cyclic-reduction levels on their own CUs (k_cr_level01, default) against one level in
    front (k_cr_level0; VBA_OPT_FUSION bit 4): the same eliminations and folds in another place, so the same bits -- for
    separator counts of every residue mod 4 (a 300-pose window cut into chunks of 4 .. 13: 74, 59, 49, 42, 37, 23 separators,
    the last one below the size from which the levels are split off at all), unpivoted and pivoted."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    cfg = synth.WindowConfig("cr2", 300, 20, 5)
    win = od_pipe.prepare_window(*synth.make_sequence(cfg, seed=11))
    n, m = win.time_idx.size, win.ii.size
    st0 = od_pipe.initial_guess(win, seed=11)
    iters, inits = [9, 10, 11, 12, 13], [True, False, False, False, False]
    for pivot in (False, True):
        outs = []
        for mask in (9, 25):
            e = BAEngine(n, m)
            e.set_solver(chunk, -1)
            e.set_pivoting(pivot)
            e.set_fusion(mask)
            e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
            e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
            e.set_states(st0, 1e-4)
            e.run_schedule(iters, inits)
            outs.append((e.get_states(), e.debug("dpose")))
            e.close()
        a, b = outs
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0][0], b[0][0]) and a[0][1] == b[0][1] and a[0][3] == b[0][3], (chunk, pivot)
        assert np.isfinite(a[1]).all() and np.abs(a[1]).max() > 0


@pytest.mark.parametrize("chunk", [4, 6, 8])
def test_resident_solve_gives_the_bits_of_three_launches(chunk):
    """VBA_OPT_FUSION bits 5 / 6 (k_solve_resident): chunk elimination, the two split-off cyclic-reduction levels and -- bit 6
    -- the one-workgroup tail as ONE grid whose consumer blocks wait for their producers on flags.  The same bodies run, so
    the bits are those of the three launches, unpivoted and pivoted; and every block publishes its flag whatever it did, so a
    handle whose windows differ in length (one of them below the size from which the levels are split off at all, one
    that finishes its schedule early... none may leave a consumer waiting) comes back with the same bits too.  Not the
    default: a hop over a flag measured slower than a kernel boundary."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    cfg = synth.WindowConfig("res", 300, 20, 5)
    win = od_pipe.prepare_window(*synth.make_sequence(cfg, seed=11))
    n, m = win.time_idx.size, win.ii.size
    st0 = od_pipe.initial_guess(win, seed=11)
    iters, inits = [9, 10, 11, 12, 13], [True, False, False, False, False]
    for pivot in (False, True):
        outs = []
        for mask in (15, 15 + 32, 15 + 64):
            e = BAEngine(n, m)
            e.set_solver(chunk, -1)
            e.set_pivoting(pivot)
            e.set_fusion(mask)
            e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
            e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
            e.set_states(st0, 1e-4)
            e.run_schedule(iters, inits)
            outs.append((e.get_states(), e.debug("dpose")))
            e.close()
        for b in outs[1:]:
            a = outs[0]
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[0][0], b[0][0]) and a[0][1] == b[0][1] and a[0][3] == b[0][3], (chunk, pivot)
        assert np.isfinite(outs[0][1]).all() and np.abs(outs[0][1]).max() > 0
    if chunk != 8:
        return
    # windows of 300, 120 (14 separators: the one-workgroup variant) and 260 poses in one handle
    cfgs = [synth.WindowConfig("r0", 300, 20, 5), synth.WindowConfig("r1", 120, 20, 5), synth.WindowConfig("r2", 260, 20, 5)]
    wins = [od_pipe.prepare_window(*synth.make_sequence(c, seed=20 + k)) for k, c in enumerate(cfgs)]
    n_max, m_max = max(w.time_idx.size for w in wins), max(w.ii.size for w in wins)
    iters, inits = list(range(6, 16)), [k < 10 for k in range(6, 16)]
    got = []
    for mask in (15, 15 + 32, 15 + 64):
        e = BAEngine(n_max, m_max, windows=3)
        e.set_fusion(mask)
        for k, w in enumerate(wins):
            e.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, w.time_idx.size, window=k)
            e.upload_window(w.intrinsics, w.cumrot_last, w.time_idx, window=k)
            e.set_states(od_pipe.initial_guess(w, seed=k), 1e-4, window=k)
        e.run_schedule(iters, inits)
        got.append([e.get_states(window=k) for k in range(3)])
        e.close()
    for b in got[1:]:
        for k in range(3):
            assert np.array_equal(got[0][k][0], b[k][0]) and got[0][k][1] == b[k][1] and got[0][k][3] == b[k][3], k


@pytest.mark.parametrize("solver", [-2], ids=["one-per-wave"])
@pytest.mark.parametrize("reg", [False, True])
def test_one_window_per_wave_walk_forming_its_own_blocks_gives_the_bits_of_the_assembled_path(c2, reg, solver):
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


@pytest.mark.parametrize("chunk", [4, 5, 6, 7, 8, 9, 11, 13])
def test_three_cyclic_reduction_levels_in_front_give_the_bits_of_two(chunk, monkeypatch):
    """Round 4, the one more structural try at the single-window solve: THREE cyclic-reduction levels on their own CUs
    (k_cr_level012: groups of eight separators, fifteen blocks, eight waves; VBA_CR_LEVELS=3) in front of a one-workgroup kernel
    that starts from n / 8 blocks.  The same eliminations and folds in another place: the bits of two levels in front -- for
    separator counts of every residue mod 8 (a 300-pose window cut into chunks of 4 .. 13: 74, 59, 49, 42, 37, 33, 27, 23
    separators, the last one below the size from which levels are split off at all), unpivoted and pivoted.  Measured SLOWER
    (C3: 11.67 + 12.95 us against 8.45 + 14.99 for the two kernels, +0.45 us on the average call): comparison build only."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    cfg = synth.WindowConfig("cr3", 300, 20, 5)
    win = od_pipe.prepare_window(*synth.make_sequence(cfg, seed=11))
    n, m = win.time_idx.size, win.ii.size
    st0 = od_pipe.initial_guess(win, seed=11)
    iters, inits = [9, 10, 11, 12, 13], [True, False, False, False, False]
    for pivot in (False, True):
        outs = []
        for levels in ("2", "3"):
            monkeypatch.setenv("VBA_CR_LEVELS", levels)
            e = BAEngine(n, m)
            e.set_solver(chunk, -1)
            e.set_pivoting(pivot)
            e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
            e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
            e.set_states(st0, 1e-4)
            e.run_schedule(iters, inits)
            outs.append((e.get_states(), e.debug("dpose")))
            e.close()
        a, b = outs
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0][0], b[0][0]) and a[0][1] == b[0][1] and a[0][3] == b[0][3], (chunk, pivot)
        assert np.isfinite(a[1]).all() and np.abs(a[1]).max() > 0


def test_resident_solve_run_twice_on_one_handle_is_never_replayed_as_a_graph():
    """k_solve_resident takes the epoch of its launch as a kernel argument; a graph replay would freeze it and the consumers'
    flags of the previous run would read as already set.  Handles with VBA_OPT_FUSION bits 5 / 6 therefore launch kernel by
    kernel: the same schedule three times on ONE handle gives the bits of a default handle every time, and captures nothing."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    cfg = synth.WindowConfig("res2", 300, 20, 5)
    win = od_pipe.prepare_window(*synth.make_sequence(cfg, seed=12))
    n, m = win.time_idx.size, win.ii.size
    st0 = od_pipe.initial_guess(win, seed=12)
    iters, inits = [9, 10, 11, 12, 13], [True, False, False, False, False]

    def run(mask):
        e = BAEngine(n, m)
        e.set_solver(8, -1)
        e.set_fusion(mask)
        e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
        e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
        outs = []
        for rep in range(3):
            e.set_states(st0, 1e-4)
            e.run_schedule(iters, inits)
            outs.append(e.get_states())
        stats = e.schedule_graph_stats()
        e.close()
        return outs, stats
    ref, stats_ref = run(15)
    assert stats_ref[0] >= 1 and sum(stats_ref) == 3       # (five calls: the state-buffer parity alternates between runs, two graphs)
    for mask in (15 + 32, 15 + 64):
        got, stats = run(mask)
        assert stats == (0, 0), (mask, stats)
        for a, b in zip(ref, got):
            assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[3] == b[3], mask
