"""Replica driver: ``errors_eval.run_folder(folder, gpus=N)`` deals the sequences of a folder longest-first to N fresh worker
processes (the reference's outer loop over sequence files, od_pipe.py:1063-1086, spread over the GPUs of a node) and merges their
results into the same errors.npy / times.npy.  On the CPU the oracle stands in for the GPU BA (as tests/test_od_pipe_host.py does);
``-m gpu``: two workers share the one device and must reproduce the single-process files bit for bit at pinned handle settings."""
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err

TESTS = os.path.join(ROOT, "tests")


def _folder(tmp_path, seqs):
    for sub in ("tmp_dets", "tmp_pose"):
        os.makedirs(tmp_path / sub)
    for k, (det, orb) in enumerate(seqs):
        np.save(tmp_path / "tmp_dets" / f"{k:03d}_all_detections.npy", det)
        np.save(tmp_path / "tmp_pose" / f"{k:03d}_orbit_eci_zyxvecs.npy", orb)
    return str(tmp_path)


def test_longest_first_split():
    from vinsat_amd.errors_eval import split_longest_first
    assert split_longest_first([10, 200, 30, 200, 5], 2) == [[1, 2], [0, 3, 4]]       # 200 | 200, 30 -> w0 (230), 10 -> w1 (210), 5 -> w1
    assert split_longest_first([7, 7, 7], 5) == [[0], [1], [2], [], []]
    assert split_longest_first([], 3) == [[], [], []]


def test_two_worker_processes_reproduce_the_single_process_files(tmp_path, monkeypatch):
    from test_od_pipe_host import _oracle_ba
    from vinsat_amd import errors_eval, synth
    seqs = [synth.make_sequence("C1", seed=0), synth.make_two_pass_sequence(), synth.make_sequence("C1", seed=1),
            synth.make_sequence("C1", seed=2), synth.make_sequence("C1", seed=3)]
    folder = _folder(tmp_path, seqs)
    e1, t1 = errors_eval.run_folder(folder, ba=_oracle_ba)
    one = [np.load(os.path.join(folder, f), allow_pickle=True) for f in ("errors.npy", "times.npy")]
    monkeypatch.setenv("PYTHONPATH", TESTS)
    stats = []
    e2, t2 = errors_eval.run_folder(folder, ba="test_od_pipe_host:_oracle_ba", gpus=2, stats=stats, timeout=600)
    two = [np.load(os.path.join(folder, f), allow_pickle=True) for f in ("errors.npy", "times.npy")]
    assert len(e2) == 5
    for k in range(5):
        assert np.array_equal(e1[k], e2[k]) and np.array_equal(t1[k], t2[k]), k
        assert np.array_equal(np.asarray(one[0][k], dtype=np.float64), np.asarray(two[0][k], dtype=np.float64))
        assert np.array_equal(np.asarray(one[1][k]), np.asarray(two[1][k]))
    assert rel_err(e2[0], load_golden("c1")["errors"]) < 1e-6 and rel_err(e2[1], load_golden("gap")["errors"]) < 1e-6
    # the two-pass sequence is the shortest in rows (144 against 200): dealt last, to the lighter worker
    assert sorted(s["sequences"] for s in stats) == [2, 3] and sum(s["rows"] for s in stats) == 144 + 4 * 200
    assert all(s["ba_calls"] == 20 * s["sequences"] + (20 if s["rows"] % 200 else 0) for s in stats)
    assert all(s["wall"] > 0 and s["prep"] > 0 for s in stats)


def test_a_failing_worker_fails_the_run(tmp_path, monkeypatch):
    from vinsat_amd import errors_eval, synth
    folder = _folder(tmp_path, [synth.make_sequence("C1", seed=0), synth.make_sequence("C1", seed=1)])
    monkeypatch.setenv("PYTHONPATH", TESTS)
    with pytest.raises(RuntimeError, match="replica worker"):
        errors_eval.run_folder(folder, ba="test_od_pipe_host:no_such_function", gpus=2, timeout=300)
    with pytest.raises(ValueError):
        errors_eval.run_folder(folder, ba=lambda *a, **k: None, gpus=2)
    assert not os.path.exists(os.path.join(folder, "errors.npy"))


@pytest.mark.gpu
def test_two_workers_on_one_device_have_the_bits_of_the_single_process_run(tmp_path):
    """HIP BA, batched driver, pinned handle settings: two worker processes sharing device 0 against this process."""
    from test_gpu_batch_surface import PINS, _sequences
    from vinsat_amd import ba, errors_eval
    seqs = _sequences()
    names = ["c1", "c2", "gap", "rej"]
    folder = _folder(tmp_path, [seqs[k] for k in names])
    try:
        e1, t1 = errors_eval.run_folder(folder, batched=True, configure=PINS)
    finally:
        ba.configure(lanes="auto", fusion="auto", solver="auto", mode="auto")
        ba.release()
    stats = []
    e2, t2 = errors_eval.run_folder(folder, batched=True, gpus=[0, 0], configure=dict(PINS, solver=list(PINS["solver"])), stats=stats, timeout=900)
    for k, name in enumerate(names):
        assert np.array_equal(e1[k], e2[k]) and np.array_equal(t1[k], t2[k]), name
        assert rel_err(e2[k], load_golden(name)["errors"]) < 1e-5, name
    assert len(stats) == 2 and all(s["device"] == 0 for s in stats)
    saved = np.load(os.path.join(folder, "errors.npy"), allow_pickle=True)
    assert all(np.array_equal(np.asarray(a, dtype=np.float64), b) for a, b in zip(saved, e2))
