"""The two code paths bench.py times, pinned at bench size (C3 500/50k, C4 500/200k, C5S 500/125k).

* headline: ``vba_run_schedule`` on a default one-window handle (latency mode: bin buckets, accept test folded into the
  next call's accumulation, ungated accumulation) -- against the reference's states (tests/golden/c3|c4|c5s.npz, made by
  the reference's driver, BA_filtering.py:4-98 called as od_pipe.py:1036-1040) and bit for bit against the call-by-call
  ``vba_iterate`` chain;
* batched: a 16-window handle (bandwidth-mode kernels: two digit passes + k_select_finish, k_obs_accumulate<8,true>,
  k_dynamics_pair on the second stream, k_assemble / k_init_step, k_solve, k_trial<1,0>, k_decide) stepped call by call
  against the same fixtures, with a window that rejects trials inside it;
* the miss paths of the warm select (forced misses, overflowing bin buckets) at 100 000 and 400 000 keys;
* the two degenerate random windows the free-running 1e-5 bar does not hold for: conditioning, shown against the oracle's
  own dense / banded solves.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, rel_err  # noqa: F401  (c2 fixture comes from conftest)
from oracle import ba_oracle as O

pytestmark = pytest.mark.gpu

CHECKPOINTS = (0, 9, 10, 14, 19)


def _window(name):
    from vinsat_amd import od_pipe, synth
    if name == "c5s":
        det, orb = synth.make_subwindow("C5", 500)
    else:
        det, orb = synth.make_sequence(name.upper())
    return od_pipe.prepare_window(det, orb)


def _engine(win, windows=1, conf=None, mode=-1):
    from vinsat_amd.engine import BAEngine
    n, m = win.time_idx.size, win.ii.size
    e = BAEngine(n, m, windows=windows, mode=mode)
    for w in range(windows):
        e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences if conf is None else conf, win.ii, n, window=w)
        e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx, window=w)
    return e


def _close_to_reference(st, ref, k):
    assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6, k
    q, qr = st[:, 3:7], ref[:, 3:7]
    assert (2 * np.arccos(np.clip(np.abs((q * qr).sum(-1)), 0, 1))).max() < 1e-6, k
    assert rel_err(st, ref) < 1e-6, k


@pytest.fixture(scope="module", params=["c3", "c4", "c5s"])
def sized(request):
    name = request.param
    if not os.path.exists(os.path.join(GOLDEN, f"{name}.npz")):
        pytest.skip(f"{name} fixture not generated")
    g = load_golden(name)
    win = _window(name)
    assert np.array_equal(win.time_idx, g["in_time_idx"])
    assert np.array_equal(np.array([win.ii.size, win.ii.sum(), win.ii[0], win.ii[-1]]), g["in_ii_digest"])
    return name, g, win


def test_headline_path_run_schedule_vs_reference_states(sized):
    """What bench.py's `value` is timed on: set_states + ONE vba_run_schedule of the 20 calls on a default handle."""
    name, g, win = sized
    iters, inits = [int(x) for x in g["iters"]], [bool(x) for x in g["initialize"]]
    eng = _engine(win)
    # (1) the 20 calls as one chained schedule
    eng.set_states(g["states0"][0], 1e-4)
    trials = eng.run_schedule(iters, inits)
    st20, lam20, hess20, ntr20, flags20 = eng.get_states()
    assert trials >= int(g["n_trials"].sum())
    assert lam20 == g["lamda_out"][19] and ntr20 == g["n_trials"][19] and flags20 == 0
    _close_to_reference(st20, g["states_out_19"][0], 19)
    assert eng.warm_select_misses() == 0 and eng.solver_fallbacks() == 0
    # (2) the same schedule cut at the calls the fixture holds: chained segments, every checkpoint against the reference
    eng.set_states(g["states0"][0], 1e-4)
    prev = 0
    for k in CHECKPOINTS:
        eng.run_schedule(iters[prev:k + 1], inits[prev:k + 1])
        st, lam, _, ntr, flags = eng.get_states()
        assert lam == g["lamda_out"][k] and ntr == g["n_trials"][k] and flags == 0, k
        _close_to_reference(st, g[f"states_out_{k}"][0], k)
        prev = k + 1
    assert np.array_equal(st, st20)
    # (3) call by call with the states crossing the host every time (vba_iterate: no carried keys, own decide launch)
    st, lam = g["states0"][0], 1e-4
    for k in range(20):
        st, lam, hess, ntr, flags = eng.iterate(iters[k], inits[k], lam, st)
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k] and flags == 0, k
    assert np.array_equal(st, st20) and np.array_equal(hess, hess20)
    eng.close()


def test_batched_path_sixteen_c3_windows_stepped_vs_reference_states():
    """What bench.py's `batched` series is timed on, at C3 size: 16 windows per launch.  Windows 0..13 are the reference's
    C3 run; window 14 starts from another initial guess (against a one-window handle); window 15 has confidences of 3, so
    that its LM loop rejects trials (against the oracle, which tests/test_oracle_golden.py pins to the reference's
    rejections): the other windows must not notice."""
    from vinsat_amd import od_pipe
    g, win = load_golden("c3"), _window("c3")
    iters, inits = [int(x) for x in g["iters"]], [bool(x) for x in g["initialize"]]
    W = 16
    conf3 = np.full_like(win.confidences, 3.0)
    eng = _engine(win, windows=W, mode=0)       # (the bandwidth-mode kernel set, as bench.py's 4096 windows run it)
    n = win.time_idx.size
    eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, conf3, win.ii, n, window=15)
    st_other = od_pipe.initial_guess(win, seed=7)
    for w in range(W):
        eng.set_states(st_other if w == 14 else g["states0"][0], 1e-4, window=w)
    single = _engine(win)
    ref14, lam14 = st_other, 1e-4
    ref15, lam15 = g["states0"][0].copy(), 1e-4
    seen15 = []
    for k in range(20):
        eng.step(iters[k], inits[k])
        ref14, lam14, _, ntr14, _ = single.iterate(iters[k], inits[k], lam14, ref14)
        lam15_in = lam15
        ref15, lam15, _, ntr15 = O.ba_iteration(iters[k], ref15, win.cumrot_last, win.landmarks_uv, win.landmarks_xyz, win.ii,
                                                win.time_idx, win.intrinsics, conf3, lam15, initialize=inits[k])
        seen15.append(ntr15)
        for w in (0, 5, 13):
            st, lam, _, ntr, flags = eng.get_states(window=w)
            assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k] and flags == 0, (k, w)
            if k in CHECKPOINTS:
                _close_to_reference(st, g[f"states_out_{k}"][0], k)
            if w == 0:
                st0w = st
            else:
                assert np.array_equal(st, st0w), (k, w)         # equal windows: equal bits
        st, lam, _, ntr, flags = eng.get_states(window=14)
        assert ntr == ntr14 and lam == lam14 and flags == 0, k
        assert rel_err(st, ref14) < 1e-6, k          # another reduction tree (8 lanes per pose, one wave per chain)
        st, lam, _, ntr, flags = eng.get_states(window=15)
        assert ntr == ntr15 and lam == lam15, (k, ntr, ntr15)
        # "lamda too large" (BA_filtering.py:75-77) only when the damping really ran out
        assert flags in (0, 1) and (flags == 0 or lam15_in * 10.0 ** ntr15 > 1e4), (k, flags)
        assert rel_err(st, ref15) < 1e-6, k
    assert max(seen15) >= 3, seen15                 # the window did reject trials
    stepped = [eng.get_states(window=w) for w in range(W)]
    # the same 20 calls chained on the device (what bench.py issues): bit for bit
    for w in range(W):
        eng.set_states(st_other if w == 14 else g["states0"][0], 1e-4, window=w)
    eng.run_schedule(iters, inits)
    for w in range(W):
        got = eng.get_states(window=w)
        assert np.array_equal(got[0], stepped[w][0]) and got[1] == stepped[w][1] and got[3] == stepped[w][3], w
    single.close()
    eng.close()


def test_first_windows_of_a_large_handle_keep_what_was_uploaded_right_after_creation():
    """A handle of 4096 C3 windows (bench.py's batched series) clears a ~50 GB arena when it is created.  The clear is a device-side fill that the host
    does not wait for by itself, and the handle's streams are non-blocking: before round 4 the fill was still running when
    the first windows were uploaded and zeroed their observations again (at 4096 windows the first 3..11 windows of
    bench.py's batched series rejected trials and ended elsewhere).  Every window is uploaded at once after creation and
    must end the chained 20-call schedule on the bits of a 4-window handle with the same settings."""
    g, win = load_golden("c3"), _window("c3")
    iters, inits = [int(x) for x in g["iters"]], [bool(x) for x in g["initialize"]]
    n = win.time_idx.size
    out = {}
    for W in (4, 4096):
        eng = _engine(win, windows=W, mode=0)
        eng.set_solver(0)
        eng.set_accumulate_lanes(8)
        eng.set_states(g["states0"][0], 1e-4, window=-1)
        eng.run_schedule(iters, inits)
        st, lam, _, ntr, flags = eng.get_states_all()
        eng.close()
        assert np.all(flags == 0) and np.all(ntr == ntr[-1]) and np.all(lam == lam[-1]), W
        assert all(np.array_equal(st[w, :n], st[-1, :n]) for w in range(W)), [w for w in range(W) if not np.array_equal(st[w, :n], st[-1, :n])][:8]
        out[W] = st[0, :n].copy()
    assert np.array_equal(out[4], out[4096])
    _close_to_reference(out[4096], g["states_out_19"][0], 19)


def test_chained_schedule_replayed_as_a_graph_gives_the_bits_of_kernel_by_kernel_launches():
    """vba_run_schedule captures the launches of a latency-mode handle's first pass as a hipGraph and replays it while nothing that
    goes into them has changed (VBA_OPT_SCHEDULE_GRAPH).  Same kernels, same arguments: every window of every schedule must end
    on the bits of a handle that launches kernel by kernel -- on the replay, after a setting that changes kernel arguments (the
    graph is captured again), after another window was uploaded (only device data changed: the graph stays), with a window that
    rejects trials (the host finishes its calls behind the replay) and after the schedule itself changed."""
    from vinsat_amd import od_pipe
    g, win = load_golden("c3"), _window("c3")
    iters, inits = [int(x) for x in g["iters"]], [bool(x) for x in g["initialize"]]
    n = win.time_idx.size
    conf3 = np.full_like(win.confidences, 3.0)
    st_other = od_pipe.initial_guess(win, seed=7)

    def run(graph):
        eng = _engine(win, windows=3, mode=1)
        eng.set_schedule_graph(graph)
        eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, conf3, win.ii, n, window=1)
        out = []

        def schedule(its, ins):
            for w in range(3):
                eng.set_states(st_other if w == 2 else g["states0"][0], 1e-4, window=w)
            eng.run_schedule(its, ins)
            out.append([eng.get_states(window=w) for w in range(3)])
        schedule(iters, inits)                  # captured
        schedule(iters, inits)                  # replayed
        eng.set_trial_tiles(1)                  # other kernel arguments (and another kernel): captured again
        eng.set_fusion(14)
        schedule(iters, inits)
        schedule(iters, inits)
        eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n, window=1)     # device data only
        schedule(iters, inits)
        schedule(iters[:12], inits[:12])        # another schedule
        schedule(iters[:12], inits[:12])
        stats = eng.schedule_graph_stats()
        eng.close()
        return out, stats
    ref, stats0 = run(False)
    got, stats1 = run(True)
    assert stats0 == (0, 0) and stats1[0] == 3 and stats1[1] == 4, (stats0, stats1)
    assert max(x[1][3] for x in ref[:4]) >= 3                                     # window 1 did reject trials
    for k, (a, b) in enumerate(zip(ref, got)):
        for w in range(3):
            assert np.array_equal(a[w][0], b[w][0]) and a[w][1] == b[w][1] and np.array_equal(a[w][2], b[w][2]) and a[w][3:] == b[w][3:], (k, w)
    _close_to_reference(got[4][0][0], g["states_out_19"][0], 19)


def test_long_gap_window_chained_and_replayed_has_the_bits_of_call_by_call_steps():
    """The second batch of the two-pass sequence (25 poses, gaps of 935 and 510 s: the long edges are propagated parallel in time by
    kernels of their own, vba_long.hip) through vba_run_schedule -- first pass captured, second replayed as a graph -- against a
    handle that steps call by call, and against the reference's own run of that batch (gap.npz, calls 20 .. 39)."""
    from vinsat_amd import od_pipe, synth
    from vinsat_amd.engine import BAEngine
    g = load_golden("gap")
    win = od_pipe.prepare_window(*synth.make_two_pass_sequence())
    n, m = win.time_idx.size, win.ii.size
    # the driver's input of the second batch: the first batch's result + dead reckoning across the gap
    from vinsat_amd import ba as ba_mod
    run = od_pipe.SequenceRun(*synth.make_two_pass_sequence())
    p = run.next_patch()
    st, vel, lam, _ = ba_mod.BA_window(range(20), [k < 10 for k in range(20)], p["states"], p["velocities"], p["imu"], p["uv"], p["xyz"],
                                       p["ii"], p["time_idx"], p["intr"], p["conf"], p["lam"])
    run.finish_patch(st, vel)
    st0 = run.next_patch()["states"][0].numpy().copy()
    ba_mod.release()
    assert st0.shape == (n, 10)
    iters, inits = [int(x) for x in g["iters"][20:]], [bool(x) for x in g["initialize"][20:]]
    assert not any(inits) and len(iters) == 20

    def engine():
        e = BAEngine(n, m)
        e.upload_observations(win.landmarks_xyz, win.landmarks_uv, win.confidences, win.ii, n)
        e.upload_window(win.intrinsics, win.cumrot_last, win.time_idx)
        return e
    e = engine()
    outs = []
    for rep in range(3):
        e.set_states(st0, float(g["lamda_in"][20]))
        e.run_schedule(iters, inits)
        outs.append(e.get_states())
    assert e.schedule_graph_stats() == (1, 2)
    e.close()
    e = engine()
    e.set_states(st0, float(g["lamda_in"][20]))
    for it, init in zip(iters, inits):
        e.step(it, init)
    ref = e.get_states()
    e.close()
    for o in outs:
        assert np.array_equal(o[0], ref[0]) and o[1] == ref[1] and np.array_equal(o[2], ref[2])
    assert rel_err(ref[0], g["states_out_39"][0]) < 1e-6 and ref[1] == g["lamda_out"][39]
    # the chain through a long gap is carried from the trial kernel to the next call's factor when it started from the very bits of
    # that call's state; a handle that has never seen the states finds the chain itself -- same function, same bits: every call
    # of a short chained run against a FRESH handle fed the states and damping the call in front left
    e = engine()
    e.set_states(st0, float(g["lamda_in"][20]))
    st, lam = st0, float(g["lamda_in"][20])
    for it in iters[:4]:
        e.step(it, False)
        want = e.get_states()
        f = engine()
        got = f.iterate(it, False, lam, st)
        f.close()
        assert np.array_equal(got[0], want[0]) and got[1] == want[1] and np.array_equal(got[2], want[2]), it
        st, lam = want[0], want[1]
    e.close()


@pytest.mark.parametrize("step", [1, 2, 3], ids=["end-capture", "instantiate", "first-launch"])
def test_a_graph_that_cannot_be_made_falls_back_to_kernel_by_kernel_launches(step):
    """A capture that cannot be ended, instantiated or launched has executed nothing: the handle gives up on graphs and enqueues
    the pass again for real (VBA_GRAPH_FAIL_INJECT pretends the failure; read once per process, hence the child process)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "from conftest import load_golden, golden_inputs\n"
        "from vinsat_amd.engine import BAEngine\n"
        "g = load_golden('c2'); inp = golden_inputs(g)\n"
        "n, m = inp['K'].shape[0], inp['xyz'].shape[0]\n"
        "e = BAEngine(n, m)\n"
        "e.upload_observations(inp['xyz'], inp['uv'], inp['conf'], inp['ii'], n); e.upload_window(inp['K'], inp['cumrot'], inp['time_idx'])\n"
        "its, ins = [int(x) for x in g['iters']], [bool(x) for x in g['initialize']]\n"
        "outs = []\n"
        "for rep in range(2):\n"
        "    e.set_states(g['states0'][0], 1e-4); e.run_schedule(its, ins); outs.append(e.get_states()[0])\n"
        "assert e.schedule_graph_stats() == (0, 0), e.schedule_graph_stats()\n"
        "ref = g['states_out_19'][0]\n"
        "assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0] - ref).max() / np.abs(ref).max() < 1e-6\n"
        "print('fallback ok')\n" % (ROOT, ROOT))
    env = dict(os.environ, VBA_GRAPH_FAIL_INJECT=str(step))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "fallback ok" in p.stdout, p.stderr[-2000:]


@pytest.mark.parametrize("fusion", [14, 12])
def test_trial_kernel_tiles_per_block_do_not_change_a_bit(fusion):
    """VBA_OPT_TRIAL_TILES: an observation block of the plain latency-mode trial kernel takes 1, 2, 4 or 8 tiles of 256 rows and
    reserves its share of the bin buckets once.  Block sums stay per tile, the histogram is integers, a bucket is a set -- so
    nothing may depend on the value: three C3-sized windows on one handle (window 1 with confidences of 3: its LM loop
    rejects trials; window 2 from another initial guess), call by call with a tight bucket capacity (overflowing buckets:
    the miss path), then the chained schedule with the allocated one."""
    from vinsat_amd import od_pipe
    g, win = load_golden("c3"), _window("c3")
    iters, inits = [int(x) for x in g["iters"]], [bool(x) for x in g["initialize"]]
    n = win.time_idx.size
    conf3 = np.full_like(win.confidences, 3.0)
    st_other = od_pipe.initial_guess(win, seed=7)
    ref = None
    for tiles in (1, 2, 4, 8):
        eng = _engine(win, windows=3, mode=1)
        eng.set_fusion(fusion)
        eng.set_trial_tiles(tiles)
        eng.upload_observations(win.landmarks_xyz, win.landmarks_uv, conf3, win.ii, n, window=1)
        got = []
        for chained in (False, True):
            eng.set_bucket_cap(0 if chained else 64)        # (0: the allocated capacity; 64: the bin of the median overflows)
            for w in range(3):
                eng.set_states(st_other if w == 2 else g["states0"][0], 1e-4, window=w)
            if chained:
                eng.run_schedule(iters, inits)
                got.append([eng.get_states(window=w) for w in range(3)])
            else:
                for k in range(20):
                    eng.step(iters[k], inits[k])
                    got.append([eng.get_states(window=w) for w in range(3)])
        misses = eng.warm_select_misses()
        eng.close()
        assert misses >= 2, misses
        if ref is None:
            ref = got
            assert max(x[1][3] for x in got) >= 3                            # window 1 did reject trials
            _close_to_reference(got[19][0][0], g["states_out_19"][0], 19)
            assert np.array_equal(got[19][0][0], got[20][0][0])              # chained = stepped (no tight buckets at the end)
            continue
        for a, b in zip(ref, got):
            for w in range(3):
                assert np.array_equal(a[w][0], b[w][0]) and a[w][1] == b[w][1] and np.array_equal(a[w][2], b[w][2]) and a[w][3:] == b[w][3:], (tiles, w)


@pytest.mark.parametrize("name", ["c3", "c4"])
def test_warm_select_miss_paths_at_bench_size(name):
    """100 000 (C3) and 400 000 (C4: 1/512-binade bins) carried keys: a warm select forced to miss on every call, and bin
    buckets too short for the bin of the median (a real overflow: the key is dropped on the writing side and the reader
    reports a miss), both end in the bits of the default path and of the exact digit passes."""
    g, win = load_golden(name), _window(name)
    iters, inits = [int(x) for x in g["iters"]], [bool(x) for x in g["initialize"]]
    outs = {}
    for mode in ("default", "exact", "forced-miss", "overflow-16", "overflow-64"):
        eng = _engine(win)
        if mode == "exact":
            eng.set_warm_select(0)
        elif mode == "forced-miss":
            eng.set_warm_select(2)
        elif mode.startswith("overflow"):
            eng.set_bucket_cap(int(mode.split("-")[1]))
        eng.set_states(g["states0"][0], 1e-4)
        eng.run_schedule(iters, inits)
        outs[mode] = eng.get_states()
        misses = eng.warm_select_misses()
        if mode in ("default", "exact"):
            assert misses == 0, (mode, misses)
        elif mode == "forced-miss":
            assert misses >= 19, misses
        else:       # the bin of the median holds ~130 (C3) / ~260 (C4) keys
            assert misses >= 15, (mode, misses)
        eng.close()
    ref = outs["default"]
    _close_to_reference(ref[0], g["states_out_19"][0], 19)
    for mode, o in outs.items():
        assert np.array_equal(o[0], ref[0]) and o[1] == ref[1] and o[3] == ref[3] and o[4] == ref[4], mode


@pytest.mark.parametrize("seed", [276, 294])
def test_degenerate_random_windows_differ_by_conditioning_not_by_code_path(seed):
    """Seeds 276 (37 poses / 116 rows) and 294 (13 poses / 16 rows) of the randomised test exceed its FREE-RUNNING 1e-5 bar.
    Each call started from the oracle's state agrees as everywhere else; and the oracle's own two solvers (LAPACK banded LU
    against dense LU on the same matrices) drift apart by as much over the six free-running calls
    (tests/test_oracle_golden.py::test_conditioning_of_the_degenerate_random_windows) -- the GPU chain stays within a small
    multiple of that."""
    import random_windows
    from vinsat_amd.engine import BAEngine
    win, xyz, uv, ii, conf, t, st0 = random_windows.make(seed)
    n = win.time_idx.size
    args = (win.cumrot_last, uv, xyz, ii, t, win.intrinsics, conf)
    eng = BAEngine(n, ii.size)
    eng.upload_observations(xyz, uv, conf, ii, n)
    eng.upload_window(win.intrinsics, win.cumrot_last, t)
    chains = {}
    for solver in ("banded", "dense"):
        st, lam = st0.copy(), 1e-4
        for it, init in random_windows.SCHEDULE:
            if solver == "banded":      # like for like: this call from the oracle's state
                out, lam_g, hess, ntr, flags = eng.iterate(it, init, lam, st)
            st, lam, hess_ref, ntr_ref = O.ba_iteration(it, st, *args, lam, initialize=init, solver=solver)
            if solver == "banded":
                assert ntr == ntr_ref and lam_g == lam
                assert rel_err(out, st) < 1e-6, it
        chains[solver] = st
    drift = rel_err(chains["banded"], chains["dense"])
    assert drift > 1e-6                         # the oracle disagrees with itself at this level
    eng.set_states(st0, 1e-4)
    eng.run_schedule([c[0] for c in random_windows.SCHEDULE], [c[1] for c in random_windows.SCHEDULE])
    free = eng.get_states()[0]
    assert min(rel_err(free, chains["banded"]), rel_err(free, chains["dense"])) < 10 * drift
    eng.close()


def test_batched_warm_select_gives_the_bits_of_the_exact_digit_passes(c2):
    """Handles of 16 windows and more select warm as well since round 3: the trial kernel bins its keys in 1/64-binade bins
    around its own median, ONE pass (k_select_warm, a block reserves its share of the list with one atomic) compacts the bin
    of the wanted rank and k_select_finish ranks that list (also the long-list path: 1/8-binade bins hold a few hundred
    keys here).  The median must be the exact lower median of the device's own residuals -- the carried keys have the bits
    the next call's reprojection produces, whichever kernel made them (vba_math.h: vba_mul / vba_add) -- and the states
    after 20 calls the bits of the exact two-pass select."""
    from conftest import golden_inputs
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    W = 16
    res = {}
    for mode in ("exact", "warm-default", "warm-49", "warm-44"):
        e = BAEngine(n, m, windows=W, mode=0)
        if mode == "exact":
            e.set_warm_select(0)
        elif mode != "warm-default":
            e.set_warm_shift(int(mode.split("-")[1]))
        for w in range(W):
            e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n, window=w)
            e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"], window=w)
            e.set_states(g["states0"][0], 1e-4, window=w)
        for k in range(20):
            e.step(k, k < 10)
            sc = e.debug("scalars", window=3)
            a = np.abs(inp["uv"] - e.debug("est", window=3)).reshape(-1)
            assert sc[0] == np.sort(a)[(a.size - 1) // 2], (mode, k)
            assert sc[7] == g["n_trials"][k]
        assert e.warm_select_misses() == 0
        res[mode] = e.get_states(window=3)
        e.close()
    for mode, o in res.items():
        assert np.array_equal(o[0], res["exact"][0]) and o[1] == res["exact"][1], mode
    assert rel_err(res["exact"][0], g["states_out_19"][0]) < 1e-7


@pytest.mark.parametrize("pivot", [False, True], ids=["unpivoted", "pivoted"])
def test_four_windows_per_wave_solve_gives_the_bits_of_one_window_per_wave(pivot):
    """The sequential driver of handles with several windows packs FOUR chains into a wavefront, one per row of 16 lanes,
    with DPP row broadcasts for the pivot columns (k_solve_quad; the default of the 4096-window bench series).  Six windows
    of DIFFERENT pose counts (a full group of four and a partial one; 23 .. 61 poses) against the same windows solved one
    per wavefront (vba_set_solver(h, -2)) on the same handle geometry: identical bits, also with row pivoting."""
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import od_pipe, synth
    lens = [48, 23, 61, 37, 52, 30]
    wins = [od_pipe.prepare_window(*synth.make_sequence(synth.WindowConfig("q", n_, 20, 5), seed=s)) for s, n_ in enumerate(lens)]
    n_max = max(w.time_idx.size for w in wins)
    m_max = max(w.ii.size for w in wins)
    sched = [(0, True), (1, True), (10, False), (11, False), (12, False), (19, False)]

    def run(solver):
        e = BAEngine(n_max, m_max, windows=len(wins))
        e.set_solver(solver)
        e.set_pivoting(pivot)
        for k, w in enumerate(wins):
            e.upload_observations(w.landmarks_xyz, w.landmarks_uv, w.confidences, w.ii, w.time_idx.size, window=k)
            e.upload_window(w.intrinsics, w.cumrot_last, w.time_idx, window=k)
            e.set_states(od_pipe.initial_guess(w, seed=k), 1e-4, window=k)
        outs = []
        for it, init in sched:
            e.step(it, init)
            outs.append([e.get_states(window=k) for k in range(len(wins))])
            outs.append([e.debug("dpose", window=k) for k in range(len(wins))])
        e.close()
        return outs

    quad, single = run(0), run(-2)
    for a, b in zip(quad, single):
        for k in range(len(wins)):
            if isinstance(a[k], tuple):
                assert np.array_equal(a[k][0], b[k][0]) and a[k][1] == b[k][1] and a[k][3] == b[k][3] and a[k][4] == b[k][4], k
            else:
                assert np.array_equal(a[k], b[k]), k
    # and against the oracle: the last window's last call from the states the device held before it
    assert all(np.isfinite(x[0]).all() for x in quad[-2])


def test_hop_integrator_chained_runs_of_the_reference_on_gpu():
    """vba_set_integrator(h, 1) / vinsat_amd.ba.configure(integrator="hop"): the integrator the reference itself takes when it
    sees a GPU (predict_gpu, BA_filtering.py:16-17).  Against the reference's own chained runs with that integrator
    (tools/gen_golden.py HOPC2 / HOPGAP): the C2 window -- every call from the reference's input states with the systems of
    calls 10 and 19, then the 20 calls chained on the device -- and the two-pass sequence through the drop-in driver (40
    calls, a ~950 s gap: nine 100 s hops and a remainder)."""
    from conftest import golden_inputs
    from vinsat_amd.engine import BAEngine
    from vinsat_amd import ba as ba_mod
    from vinsat_amd import od_pipe, synth
    g = load_golden("hopc2")
    inp = golden_inputs(g)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    eng = BAEngine(n, m)
    eng.set_integrator(True)
    eng.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    eng.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    for k in range(20):
        st_in = g[f"states_in_{k}"][0] if f"states_in_{k}" in g else (g["states0"][0] if k == 0 else g[f"states_out_{k-1}"][0])
        out, lam, hess, ntr, flags = eng.iterate(int(g["iters"][k]), bool(g["initialize"][k]), float(g["lamda_in"][k]), st_in)
        assert ntr == g["n_trials"][k] and lam == g["lamda_out"][k] and flags == 0, k
        assert rel_err(out, g[f"states_out_{k}"][0]) < 1e-8, k
        if f"A_bands_{k}" in g:
            A = eng.debug("bands")
            A[:, 1] += eng.debug("scalars")[4] * np.eye(9)
            assert rel_err(A, g[f"A_bands_{k}"][0]) < 1e-11, k
            assert rel_err(eng.debug("rhs"), g[f"JTr_{k}"][0].reshape(n, 9)) < 1e-9, k
            assert np.abs(eng.debug("r_pred") - g[f"r_pred_{k}"][0]).max() < 1e-9, k
            D = np.array([1, 1, 1, 100.0, 100, 100])
            Jf = g[f"Jf_blocks_{k}"][:, 0]
            Phi = eng.debug("Phi")
            assert rel_err((D[None, :, None] * Phi[:-1])[:, :, :3], Jf[:, :, :3]) < 1e-12
            assert rel_err((D[None, :, None] * Phi[:-1])[:, :, 3:], Jf[:, :, 6:]) < 1e-12
    eng.set_states(g["states0"][0], 1e-4)
    eng.run_schedule([int(x) for x in g["iters"]], [bool(x) for x in g["initialize"]])
    st, lam, _, ntr, flags = eng.get_states()
    assert lam == g["lamda_out"][19] and flags == 0
    assert rel_err(st, g["states_out_19"][0]) < 1e-7
    eng.close()
    # the two-pass sequence through the drop-in driver
    gg = load_golden("hopgap")
    det, orb = synth.make_two_pass_sequence()
    ba_mod.release()
    ba_mod.configure(integrator="hop")
    try:
        rec = []
        errors, first_det, times = od_pipe.streaming_version(detections=det, orbit_np=orb, record=rec)
    finally:
        ba_mod.configure(integrator="rk4")
        ba_mod.release()
    assert [r["states"].shape[1] for r in rec] == list(gg["n_poses_per_call"])
    for k in range(40):
        ref = gg[f"states_out_{k}"][0]
        st = rec[k]["states"][0].numpy()
        assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6, k
        assert rel_err(st, ref) < 1e-6, k
        assert rec[k]["lamda"] == gg["lamda_out"][k]
    assert rel_err(errors.numpy(), gg["errors"]) < 1e-5


def _ba_args(inp, n):
    import torch
    imu = torch.zeros((1, n, 1, 10), dtype=torch.float64)
    imu[0, :, 0, 6:10] = torch.from_numpy(inp["cumrot"])
    return dict(imu=imu, uv=torch.from_numpy(inp["uv"])[None], xyz=torch.from_numpy(inp["xyz"])[None], K=torch.from_numpy(inp["K"])[None],
                conf=torch.from_numpy(inp["conf"]), ii=inp["ii"].copy(), t=inp["time_idx"].copy())


@pytest.mark.parametrize("fusion", [None, 14], ids=["mask-auto", "mask-14"])
@pytest.mark.parametrize("fixture", ["c2", "rej"])
def test_pipelined_driver_loop_gives_the_bits_of_call_by_call_steps(fixture, fusion):
    """vinsat_amd.ba.BA in the reference's loop shape (od_pipe.py:1036-1040).  Behind every call that feeds back the previous
    result the library enqueues the next call speculatively (iter + 1; the phase change at iter 10 is a wrong guess the first
    time and learnt afterwards); the results must be the bits of vba_step-by-step runs and of the reference's fixtures -- also
    when calls reject trials and exhaust lamda (rej.npz: 1 .. 9 trials per call), where the speculated call finds the accept
    test of the call in front not clean and skips itself."""
    import torch
    from conftest import golden_inputs
    from vinsat_amd import ba as ba_mod
    from vinsat_amd.engine import BAEngine
    g = load_golden(fixture)
    inp = golden_inputs(g)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    a = _ba_args(inp, n)
    ba_mod.release()
    # mask 14 (what big single windows get: the step is formed by the fused landmark-only assembly / the recovery of the
    # partitioned solve, which then also write the trial states to mapped host memory): the loop must pipeline there too
    ba_mod.configure(fusion=fusion if fusion is not None else "auto")
    outs = []
    for rep in range(3):        # the second and third window profit from what the first one taught
        st, lam = torch.from_numpy(g["states0"][0].copy())[None], 1e-4
        seq = []
        for k in range(20):
            st, _, lam, hess = ba_mod.BA(int(g["iters"][k]), st, None, a["imu"], a["uv"], a["xyz"], a["ii"], a["t"], a["K"], a["conf"], 1e-3, 1e-3,
                                         lam, None, initialize=bool(g["initialize"][k]))
            assert ba_mod.BA.last["n_trials"] == g["n_trials"][k] and lam == g["lamda_out"][k], (rep, k)
            seq.append((st.numpy()[0].copy(), lam, hess.numpy()[0].copy()))
        outs.append(seq)
    hits, discards = ba_mod._cache["eng"].pipeline_stats()
    # most calls were found already enqueued (a call that rejects trials closes the chain: the next one starts afresh)
    assert hits >= (45 if fixture == "c2" else 10), (hits, discards)
    ba_mod.configure(fusion="auto")
    ba_mod.release()
    # the same schedule step by step on a plain engine
    e = BAEngine(n, m)
    e.set_pipeline(False)
    if fusion is not None:
        e.set_fusion(fusion)
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    e.set_states(g["states0"][0], 1e-4)
    for k in range(20):
        e.step(int(g["iters"][k]), bool(g["initialize"][k]))
        st, lam, hess, ntr, flags = e.get_states()
        for rep in range(3):
            assert np.array_equal(outs[rep][k][0], st) and outs[rep][k][1] == lam and np.array_equal(outs[rep][k][2], hess), (rep, k)
        assert rel_err(st, g[f"states_out_{k}"][0]) < 1e-6, k
    e.close()


def test_BA_sees_in_place_edits_of_its_numpy_arguments(c2):
    """The reference passes ii and time_idx as ndarrays; a drop-in must not serve stale device data when the caller edits one
    of them in place between calls (no identity check can see that): the content is compared with a private copy."""
    import torch
    from conftest import golden_inputs
    from vinsat_amd import ba as ba_mod
    g, inp = c2, golden_inputs(c2)
    n = inp["K"].shape[0]
    a = _ba_args(inp, n)
    ba_mod.release()
    st0 = torch.from_numpy(g["states0"][0].copy())[None]

    def call():
        return ba_mod.BA(0, st0, None, a["imu"], a["uv"], a["xyz"], a["ii"], a["t"], a["K"], a["conf"], 1e-3, 1e-3, 1e-4, None, initialize=True)[0].numpy().copy()

    base = call()
    assert np.array_equal(call(), base)
    k = 1234                                    # one row moves to the neighbouring pose: nothing a strided sample would see
    assert a["ii"][k] + 1 < n
    a["ii"][k] += 1
    a["ii"].sort()
    edited = call()
    assert not np.array_equal(edited, base)
    fresh = dict(a, ii=a["ii"].copy())
    ba_mod.release()
    ref = ba_mod.BA(0, st0, None, fresh["imu"], fresh["uv"], fresh["xyz"], fresh["ii"], fresh["t"], fresh["K"], fresh["conf"], 1e-3, 1e-3, 1e-4, None,
                    initialize=True)[0].numpy()
    assert np.array_equal(edited, ref)
    ba_mod.release()


@pytest.mark.parametrize("which", ["time_idx", "intrinsics", "confidences", "confidences-strided"])
def test_BA_sees_in_place_edits_of_any_of_seven_numpy_arguments_between_resident_calls(c2, which):
    """An all-NumPy caller: imu_meas, landmarks, landmarks_xyz, ii, time_idx, intrinsics, confidences are seven ndarrays.  Every
    one of them is watched by the library during the RESIDENT calls of the loop ``states = BA(iter, states, ...)`` (eight watch
    slots; round 3 had four, so the 5th .. 7th array could be edited unseen); an array that cannot be watched (a strided view) is
    compared on the Python side before every call.  The call after an in-place edit must run on the edited window."""
    from conftest import golden_inputs
    from vinsat_amd import ba as ba_mod
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    imu = np.zeros((1, n, 1, 10))
    imu[0, :, 0, 6:] = inp["cumrot"]
    conf_store = np.zeros((m, 2))
    conf_store[:, 0] = inp["conf"]
    a = dict(imu=imu, uv=inp["uv"][None].copy(), xyz=inp["xyz"][None].copy(), ii=inp["ii"].copy(), t=inp["time_idx"].copy(),
             K=inp["K"][None].copy(), conf=conf_store[:, 0] if which.endswith("strided") else inp["conf"].copy())
    assert all(isinstance(v, np.ndarray) for v in a.values()) and a["conf"].flags.c_contiguous != which.endswith("strided")

    sched = [(0, True), (1, True), (10, False), (11, False)]       # (the edit comes in front of a full-phase call: time_idx matters there)

    def loop(args, edit_before=None):
        ba_mod.release()
        st, lam = g["states0"].copy(), 1e-4
        outs = []
        for k, (it, init) in enumerate(sched):
            if k == edit_before:
                edit(args)
            st, _, lam, _ = ba_mod.BA(it, st, None, args["imu"], args["uv"], args["xyz"], args["ii"], args["t"], args["K"], args["conf"],
                                      1e-3, 1e-3, lam, None, initialize=init)
            outs.append(st.numpy().copy())
        return outs

    def edit(args):
        if which == "time_idx":
            args["t"][-1] += 3                      # the last gap grows: another dynamics factor
        elif which == "intrinsics":
            args["K"][0, :, 0] *= 1.001
        else:
            args["conf"][::2] *= 0.5

    orig = {k: np.array(v) for k, v in a.items()}
    base = loop(a)
    edited = loop(a, edit_before=2)                 # calls 0, 1 resident on the old window, the edit, calls 2, 3
    assert np.array_equal(edited[1], base[1]) and not np.array_equal(edited[2], base[2])
    # what a caller gets who never edits in place: calls 0, 1 on the original arrays, call 2 on fresh copies of the edited ones
    fresh = {k: np.array(v) for k, v in a.items()}
    ba_mod.release()
    st, lam = g["states0"].copy(), 1e-4
    for k, (it, init) in enumerate(sched[:3]):
        w = orig if k < 2 else fresh
        st, _, lam, _ = ba_mod.BA(it, st, None, w["imu"], w["uv"], w["xyz"], w["ii"], w["t"], w["K"], w["conf"], 1e-3, 1e-3, lam, None,
                                  initialize=init)
    assert rel_err(edited[2], st.numpy()) < 1e-12
    ba_mod.release()


@pytest.mark.parametrize("which", ["imu", "uv", "xyz", "ii", "t", "K", "conf"])
def test_BA_window_sees_in_place_edits_of_its_numpy_arguments(c2, which):
    """``BA_window`` goes set_states -> vba_run_schedule, which does not evaluate the library's host watch: an ndarray argument
    edited in place between two calls (same Python object) is compared with the uploaded copy in front of the device call.  All
    seven array arguments as ndarrays, one of them edited."""
    from conftest import golden_inputs
    from vinsat_amd import ba as ba_mod
    g, inp = c2, golden_inputs(c2)
    n = inp["K"].shape[0]
    imu = np.zeros((1, n, 1, 10))
    imu[0, :, 0, 6:] = inp["cumrot"]
    a = dict(imu=imu, uv=inp["uv"][None].copy(), xyz=inp["xyz"][None].copy(), ii=inp["ii"].copy(), t=inp["time_idx"].copy(),
             K=inp["K"][None].copy(), conf=inp["conf"].copy())
    iters, inits = [0, 1, 10, 11], [True, True, False, False]

    def call(w):
        return ba_mod.BA_window(iters, inits, g["states0"].copy(), None, w["imu"], w["uv"], w["xyz"], w["ii"], w["t"], w["K"], w["conf"], 1e-4)[0].numpy().copy()

    ba_mod.release()
    base = call(a)
    assert np.array_equal(call(a), base)
    if which == "imu":
        a["imu"][0, 3:9, 0, 6:] = a["imu"][0, 4:10, 0, 6:].copy()
    elif which == "uv":
        a["uv"][0, 100:200] += 3.0
    elif which == "xyz":
        a["xyz"][0, 777] += 0.5
    elif which == "ii":
        a["ii"][1234] += 1
        a["ii"].sort()
    elif which == "t":
        a["t"][-1] += 3
    elif which == "K":
        a["K"][0, :, 0] *= 1.001
    else:
        a["conf"][::2] *= 0.5
    edited = call(a)
    assert not np.array_equal(edited, base)
    fresh = {k: np.array(v) for k, v in a.items()}
    ba_mod.release()
    assert np.array_equal(call(fresh), edited)
    ba_mod.release()


def test_strict_mode_sees_a_torch_argument_edited_through_its_numpy_alias(c2):
    """``tensor.numpy()[...] = x`` does not bump ``_version``: by default the drop-in trusts address / shape / version of torch
    arguments (``invalidate()`` is the documented way out); ``configure(strict=True)`` content-checks torch CPU arguments like
    ndarrays -- between resident calls of the driver's loop (library-side comparison) and in front of ``BA_window``."""
    import torch
    from conftest import golden_inputs
    from vinsat_amd import ba as ba_mod
    g, inp = c2, golden_inputs(c2)
    n = inp["K"].shape[0]
    a = _ba_args(inp, n)
    assert isinstance(a["conf"], torch.Tensor) and isinstance(a["uv"], torch.Tensor)
    sched = [(0, True), (1, True), (2, True)]

    def loop(args, edit_before=None, window=False):
        ba_mod.release()
        st, lam = torch.from_numpy(g["states0"].copy()), 1e-4
        outs = []
        for k, (it, init) in enumerate(sched):
            if k == edit_before:
                args["conf"].numpy()[::2] *= 0.5
            if window:
                st, _, lam, _ = ba_mod.BA_window([it], [init], st, None, args["imu"], args["uv"], args["xyz"], args["ii"], args["t"], args["K"],
                                                 args["conf"], lam)
            else:
                st, _, lam, _ = ba_mod.BA(it, st, None, args["imu"], args["uv"], args["xyz"], args["ii"], args["t"], args["K"], args["conf"],
                                          1e-3, 1e-3, lam, None, initialize=init)
            outs.append(st.numpy().copy())
        return outs

    try:
        ba_mod.configure(strict=True)
        for window in (False, True):
            b = dict(a, conf=a["conf"].clone())
            base = loop(b, window=window)
            ver = b["conf"]._version
            edited = loop(b, edit_before=2, window=window)
            assert b["conf"]._version == ver                                # the edit was invisible to the version counter ...
            assert np.array_equal(edited[1], base[1]) and not np.array_equal(edited[2], base[2]), window   # ... and seen all the same
            # what a caller gets who never edits in place: calls 0, 1 on the original tensor, call 2 on a fresh copy of the edited one
            fresh = dict(b, conf=b["conf"].clone())
            ba_mod.release()
            st_in, lam_in = torch.from_numpy(g["states0"].copy()), 1e-4
            for k, (it, init) in enumerate(sched):
                w = a if k < 2 else fresh
                st_in, _, lam_in, _ = ba_mod.BA(it, st_in, None, w["imu"], w["uv"], w["xyz"], w["ii"], w["t"], w["K"], w["conf"], 1e-3, 1e-3, lam_in,
                                                None, initialize=init)
            assert rel_err(edited[2], st_in.numpy()) < 1e-12, window
    finally:
        ba_mod.configure(strict=False)
        ba_mod.release()


def test_pipeline_predictor_does_not_learn_across_window_boundaries(c2):
    """Six windows in a row through the driver's loop (call 0 with uploaded states, calls 1 .. 19 resident): the speculated calls
    that are dropped stay at one per boundary at most (the call guessed behind iter 19) plus the one wrong guess at the first
    phase switch -- round 3 learnt "iter 1 follows iter 19" from the boundary and wasted a call every second window."""
    from conftest import golden_inputs
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    e = BAEngine(n, m)
    e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
    e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
    windows = 6
    for w in range(windows):
        st, lam, _, _, _ = e.iterate(0, True, 1e-4, g["states0"][0])
        for k in range(1, 20):
            st, lam, _, _, _ = e.iterate_resident(k, k < 10)
        assert rel_err(st, g["states_out_19"][0]) < 1e-7
    hits, discards = e.pipeline_stats()
    assert hits >= windows * 17 and discards <= windows + 1, (hits, discards)
    e.close()


def test_pipelined_calls_survive_whatever_happens_between_them(c2):
    """The speculated call is an implementation detail: reading the states, fetching intermediates, toggling a switch,
    uploading the window again or running a plain step between two resident calls must neither change a result nor leave the
    handle in a state from which the next call differs from the unpipelined sequence -- BA and BA_reg, warm-select misses
    forced on every call included."""
    from conftest import golden_inputs
    from vinsat_amd.engine import BAEngine
    g, inp = c2, golden_inputs(c2)
    gr = load_golden("reg_c2")
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]

    def make(pipe, reg, warm):
        e = BAEngine(n, m)
        e.set_pipeline(pipe)
        e.set_warm_select(warm)
        e.upload_observations(inp["xyz"], inp["uv"], inp["conf"], inp["ii"], n)
        e.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])
        if reg:
            e.upload_prior(gr["states_prior"][0], gr["hessian_state_t"][0])
            e.set_prior(True)
        return e

    for reg, warm in ((False, 1), (True, 1), (False, 2)):
        plain, piped = make(False, reg, warm), make(True, reg, warm)
        rp = plain.iterate(0, True, 1e-4, g["states0"][0])
        rq = piped.iterate(0, True, 1e-4, g["states0"][0])
        assert np.array_equal(rp[0], rq[0])
        for k in range(1, 20):
            init = k < 10
            rp = plain.iterate_resident(k, init)
            rq = piped.iterate_resident(k, init)
            assert np.array_equal(rp[0], rq[0]) and rp[1] == rq[1] and np.array_equal(rp[2], rq[2]) and rp[3] == rq[3] and rp[4] == rq[4], (reg, warm, k)
            # something else happens to the pipelined handle between the calls
            if k % 5 == 1:
                st = piped.get_states()
                assert np.array_equal(st[0], rq[0]) and st[1] == rq[1]
            elif k % 5 == 2:
                from vinsat_amd._lib import VbaError
                with pytest.raises(VbaError):               # intermediates of a pipelined call are not kept: refused, loudly
                    piped.debug("rhs")
            elif k % 5 == 3:
                piped.set_chunk_waves(2)                    # (a no-op value: the speculated call is dropped all the same)
            elif k == 9:
                piped.upload_window(inp["K"], inp["cumrot"], inp["time_idx"])        # the same constants again: "a new window"
                piped.set_states(rq[0], rq[1])
                plain.set_states(rp[0], rp[1])
                rp = plain.iterate(10, False, rp[1], rp[0])
                rq = piped.iterate(10, False, rq[1], rq[0])
                assert np.array_equal(rp[0], rq[0])
        if warm == 2:
            assert piped.warm_select_misses() > 10
        hits, drops = piped.pipeline_stats()
        assert hits > 0 and drops > 0
        plain.close()
        piped.close()
