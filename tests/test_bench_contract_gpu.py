"""bench.py contract on the GPU box: exactly one JSON line on stdout with the keys the driver reads."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "20",
                          "--windows", "8", "--batched-steps", "20", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 20 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 100 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["accuracy"]["max_rel_pos_err_vs_ref"] < 1e-6
    assert d["batched"]["windows"] == 8 and d["batched"]["value"] > d["value"]
    assert d["python_BA_call"]["value"] > 100 and d["roofline"]["whole_call"]["bytes"] == 208.0 * 50000 + 5000.0 * 500


def test_gpus_2_runs_two_replica_ranks_on_this_box():
    """The driver's `python bench.py --gpus N` must measure N ranks.  On a one-GPU box both ranks share the device (the
    control plane is gloo, replicas need no RCCL); the RCCL leg says that it was skipped."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "20",
                          "--windows", "0", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "replicas2"
    assert d["value"] > 100 and abs(d["value"] * d["ms_per_step"] / 1e3 - 2.0) < 1e-6       # both windows' iterations count
    import torch
    if torch.cuda.device_count() < 2:
        assert "skipped" in d["sharded"]
    else:
        assert d["sharded"]["rccl_ranks"] == 2


def test_gpus_2_rehearses_the_library_issued_sharded_leg_on_the_test_double_of_rccl():
    """`bench.py --gpus 2` with VBA_BENCH_FAKE_RCCL=1 on the one-GPU box: both rank processes run the library-issued sharded
    leg (vba_sh_run_schedule, carried-keys protocol) end to end on tests/fake_rccl -- the bench's own orchestration (joint
    decision to run, barriers, max-over-ranks, statistics) is what is rehearsed; the rate is not a measurement."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices: the real RCCL leg runs instead")
    fake_dir = os.path.join(ROOT, "tests", "fake_rccl")
    subprocess.check_call(["make", "-C", fake_dir], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, VBA_BENCH_FAKE_RCCL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "20", "--windows", "0",
                          "--cpu-seconds", "0", "--sweep", "", "--no-configs", "--no-schur"], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    sh = d["sharded"]
    assert "rehearsal" in sh and sh["value_library_issued"] > 10 and "native_error" not in sh, sh
    # the windows BASELINE.json names as sharded, FIXED totals split over the two ranks, beside their one-GPU rates; and the
    # weak-scaled headline window
    for key, poses, rows in (("C4", 500, 200000), ("C5", 2004, 500000), ("weak", 500, 100000)):
        e = sh[key]
        assert e["poses"] == poses and e["observations_total"] == rows and e["observations_per_rank"] == rows // 2, (key, e)
        assert e["value_library_issued"] > 10 and e["one_gpu_unsharded"] > 10 and "native_error" not in e, (key, e)
        assert e["first_exchange_bytes_per_rank"] <= 32 * 1024 and e["calls_repeated_after_a_missed_select"] >= 0, (key, e)
