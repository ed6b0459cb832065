"""The batch dimension of ``BA`` behind the reference's own call surface (``-m gpu``).

The reference's ``BA`` carries ``bsz = states.shape[0]`` (``BA_filtering.py:14``) and hard-codes batch index 0 (``:24, :37``); its
real batch axis is the loop over sequences (``od_pipe.py:1069-1077``).  Here ``vinsat_amd.ba.BA`` / ``BA_window`` take a batch
(dense ``[B, n, 10]`` or a list of windows of different sizes) and ``errors_eval.run_folder(batched=True)`` drives every
sequence's current batch as a window of ONE ragged handle.  Checked: the reference's states (fixtures C1, C2, GAP, REJ made by
the reference's own ``streaming_version``) to <= 1e-6, trial counts and dampings exactly, and -- at equal handle settings -- the
bits of the four sequential runs.
"""
import os

import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

# settings that a handle otherwise derives from its own geometry (window count, largest window): pinned, a window computes the
# same bits alone and in a batch -- 8 lanes per pose, chunks of 8 poses + cyclic reduction, step and blocks formed by their own launches
PINS = dict(lanes=8, fusion=12, solver=(8, -1), mode="lat")


def _sequences():
    from vinsat_amd import synth
    return {"c1": synth.make_sequence("C1"), "c2": synth.make_sequence("C2", seed=0), "gap": synth.make_two_pass_sequence(),
            "rej": synth.make_sequence("C2", seed=3, conf=3.0)}


@pytest.fixture
def pinned():
    from vinsat_amd import ba
    ba.configure(**PINS)
    yield ba
    ba.configure(lanes="auto", fusion="auto", solver="auto", mode="auto")
    ba.release()


def _first_patches(seqs):
    from vinsat_amd.od_pipe import SequenceRun
    return [SequenceRun(det.copy(), orb.copy()).next_patch() for det, orb in seqs]


def _check_against_fixture(name, g, k, st, lam, ntr):
    ref = g[f"states_out_{k}"][0]
    assert st.shape == ref.shape, (name, k)
    assert np.abs(st[:, :3] - ref[:, :3]).max() / np.abs(ref[:, :3]).max() < 1e-6, (name, k)
    assert (2 * np.arccos(np.clip(np.abs((st[:, 3:7] * ref[:, 3:7]).sum(-1)), 0, 1))).max() < 1e-6, (name, k)
    assert rel_err(st, ref) < 1e-6, (name, k)
    assert lam == g["lamda_out"][k] and ntr == g["n_trials"][k], (name, k, lam, ntr)


@pytest.mark.parametrize("pins", [False, True], ids=["default-settings", "pinned-settings"])
def test_BA_with_a_ragged_batch_of_four_windows_vs_reference_and_sequential_runs(pins):
    """``for iter in range(20): states, ... = BA(iter, states, ...)`` (od_pipe.py:1036-1040) with FOUR windows per call -- C1 (10
    poses / 200 rows), C2 (100 / 5 000), the first batch of the two-pass sequence (12 / 72) and the rejecting window (100 /
    5 000, 1 .. 9 LM trials per call, lamda exhaustion) -- as lists of the reference's single-window arguments.  Every call of
    every window against the reference's own run; with pinned settings also bit for bit against four sequential loops."""
    from vinsat_amd import ba
    names = ["c1", "c2", "gap", "rej"]
    seqs = _sequences()
    gold = {k: load_golden(k) for k in names}
    if pins:
        ba.configure(**PINS)
    try:
        ps = _first_patches([seqs[k] for k in names])
        lst = lambda key: [p[key] for p in ps]
        states, lams = lst("states"), [1e-4] * 4
        per_call = []
        for k in range(20):
            states, vel, lams, hess = ba.BA(k, states, lst("velocities"), lst("imu"), lst("uv"), lst("xyz"), lst("ii"), lst("time_idx"),
                                            lst("intr"), lst("conf"), 1e-3, 1e-3, lams, None, initialize=k < 10)
            assert isinstance(states, list) and len(states) == 4 and len(lams) == 4 and len(hess) == 4
            assert all(h.shape == (1, 9, 9) for h in hess) and vel[1] is ps[1]["velocities"]
            for b, name in enumerate(names):
                assert states[b].shape == ps[b]["states"].shape
                _check_against_fixture(name, gold[name], k, states[b][0].numpy(), lams[b], ba.BA.last["n_trials"][b])
            per_call.append(([s.numpy().copy() for s in states], list(lams), [h.numpy().copy() for h in hess]))
        assert max(gold["rej"]["n_trials"]) == 9           # (the batch really contained stalling windows)
        if not pins:
            return
        for b, name in enumerate(names):            # the same loop, one window at a time, same handle settings
            p = ps[b]
            st, lam = p["states"], 1e-4
            for k in range(20):
                st, _, lam, hs = ba.BA(k, st, p["velocities"], p["imu"], p["uv"], p["xyz"], p["ii"], p["time_idx"], p["intr"], p["conf"],
                                       1e-3, 1e-3, lam, None, initialize=k < 10)
                assert np.array_equal(st.numpy(), per_call[k][0][b]), (name, k)
                assert lam == per_call[k][1][b] and np.array_equal(hs.numpy(), per_call[k][2][b]), (name, k)
    finally:
        ba.configure(lanes="auto", fusion="auto", solver="auto", mode="auto")
        ba.release()


def test_BA_with_a_dense_batch_has_the_reference_shapes(pinned):
    """The reference's own tensor shapes with bsz = 3: states [3, n, 10], landmarks [3, m, 2], ... ; ``ii`` and ``time_idx``
    shared, confidences per window ([3, m]: the golden ones, confidences of 3 -- rejections --, and a scaled copy).  Window 0 is
    the reference's C2 run; every window has the bits of its own single-window loop."""
    import torch
    from conftest import golden_inputs
    ba = pinned
    g = load_golden("c2")
    inp = golden_inputs(g)
    n, m = inp["K"].shape[0], inp["xyz"].shape[0]
    B = 3
    conf = np.stack([inp["conf"], np.full(m, 3.0), inp["conf"] * 0.9])
    imu = torch.zeros((B, n, 2, 10), dtype=torch.float64)
    imu[:, :, -1, 6:] = torch.from_numpy(inp["cumrot"])
    rep = lambda a: torch.from_numpy(np.repeat(a[None], B, axis=0))
    uv, xyz, K = rep(inp["uv"]), rep(inp["xyz"]), rep(inp["K"])
    vel = torch.from_numpy(np.repeat(g["in_velocities"], B, axis=0))
    states, lams = rep(g["states0"][0]), 1e-4
    outs = []
    for k in range(20):
        states, v_out, lams, hess = ba.BA(k, states, vel, imu, uv, xyz, inp["ii"], inp["time_idx"], K, torch.from_numpy(conf), 1e-3, 1e-3,
                                          lams, None, initialize=k < 10)
        assert v_out is vel and states.shape == (B, n, 10) and hess.shape == (B, 9, 9) and len(lams) == B
        _check_against_fixture("c2", g, k, states[0].numpy(), lams[0], ba.BA.last["n_trials"][0])
        outs.append((states.numpy().copy(), list(lams)))
    assert max(ba.BA.last["n_trials"]) >= 1
    for b in range(B):
        st, lam = torch.from_numpy(g["states0"]), 1e-4
        for k in range(20):
            st, _, lam, _ = ba.BA(k, st, vel[b:b + 1], imu[b:b + 1], uv[b:b + 1], xyz[b:b + 1], inp["ii"], inp["time_idx"], K[b:b + 1],
                                  torch.from_numpy(conf[b]), 1e-3, 1e-3, lam, None, initialize=k < 10)
            assert np.array_equal(st[0].numpy(), outs[k][0][b]) and lam == outs[k][1][b], (b, k)


def test_BA_batch_sees_new_windows_and_in_place_edits(pinned):
    """The batch engine keeps its windows on the device between calls: another list of windows, or an ndarray argument edited in
    place, must be seen (content of ndarrays is compared with the uploaded copy before every call)."""
    ba = pinned
    seqs = _sequences()
    ps = _first_patches([seqs["c1"], seqs["gap"]])
    lst = lambda key: [p[key] for p in ps]

    def call(states, lams, confs, iis):
        return ba.BA(0, states, lst("velocities"), lst("imu"), lst("uv"), lst("xyz"), iis, lst("time_idx"), lst("intr"), confs,
                     1e-3, 1e-3, lams, None, initialize=True)

    iis = [np.array(p["ii"]) for p in ps]
    a, _, la, _ = call(lst("states"), [1e-4, 1e-4], lst("conf"), iis)
    b, _, lb, _ = call(lst("states"), [1e-4, 1e-4], lst("conf"), iis)
    assert all(np.array_equal(x.numpy(), y.numpy()) for x, y in zip(a, b)) and la == lb
    conf2 = [ps[0]["conf"] * 0.5, ps[1]["conf"]]
    c, _, _, _ = call(lst("states"), [1e-4, 1e-4], conf2, iis)
    assert not np.array_equal(c[0].numpy(), a[0].numpy()) and np.array_equal(c[1].numpy(), a[1].numpy())
    # an in-place edit of ii (rows of window 0 handed to another pose)
    iis[0][:5] = iis[0][-1]
    d, _, _, _ = call(lst("states"), [1e-4, 1e-4], conf2, iis)
    assert not np.array_equal(d[0].numpy(), c[0].numpy()) and np.array_equal(d[1].numpy(), c[1].numpy())


def test_run_folder_batched_on_one_ragged_handle(tmp_path, pinned):
    """``errors_eval.run_folder(batched=True)``: the four sequences (C1, C2, the two-pass GAP sequence with its second batch of 25
    poses and a ~945 s gap, REJ) as windows of ONE handle -- round 0 four windows, round 1 the one sequence that has a second
    batch.  Against the reference's own results (errors / times of the fixtures; states after every round) and, at equal handle
    settings, bit for bit against the sequential ``run_folder``."""
    from vinsat_amd import errors_eval, od_pipe
    names = ["c1", "c2", "gap", "rej"]
    seqs = _sequences()
    gold = {k: load_golden(k) for k in names}
    for sub in ("tmp_dets", "tmp_pose"):
        os.makedirs(tmp_path / sub)
    for k, name in enumerate(names):
        det, orb = seqs[name]
        np.save(tmp_path / "tmp_dets" / f"{k:02d}_all_detections.npy", det)
        np.save(tmp_path / "tmp_pose" / f"{k:02d}_orbit_eci_zyxvecs.npy", orb)
    e_b, t_b = errors_eval.run_folder(str(tmp_path), batched=True)
    saved = np.load(tmp_path / "errors.npy", allow_pickle=True)
    assert len(saved) == 4 and all(np.array_equal(np.asarray(a, dtype=np.float64), b) for a, b in zip(saved, e_b))
    e_s, t_s = errors_eval.run_folder(str(tmp_path))
    for k, name in enumerate(names):
        g = gold[name]
        assert np.array_equal(e_b[k], e_s[k]), name                     # bits of the sequential run
        assert np.array_equal(t_b[k], t_s[k]) and np.array_equal(t_b[k], g["times"]), name
        assert rel_err(e_b[k], g["errors"]) < 1e-5, name
    # states after every round against the reference's own run
    rec = []
    res = od_pipe.streaming_batched([(d.copy(), o.copy()) for d, o in (seqs[k] for k in names)], record=rec)
    assert [(r["round"], r["sequence"]) for r in rec] == [(0, 0), (0, 1), (0, 2), (0, 3), (1, 2)]
    for r in rec:
        g = gold[names[r["sequence"]]]
        call = 19 + 20 * r["round"]
        ref = g[f"states_out_{call}"][0]
        assert rel_err(r["states"][0].numpy(), ref) < 1e-6 and r["lamda"] == g["lamda_out"][call], (r["round"], r["sequence"])
    assert int(res[2][1]) == int(gold["gap"]["first_detection"])
