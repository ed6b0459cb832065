"""The per-row part of the driver's data preparation on the device (``vba_prepare_rows``, ``od_pipe.prepare_window(device=...)``)
against the host path, which is bit-identical to the reference's own arrays (``tests/test_od_pipe_host.py``)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "two-pass"])
def test_device_rows_agree_with_the_host_preparation(name):
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_two_pass_sequence() if name == "two-pass" else synth.make_sequence(name)
    host = od_pipe.prepare_window(det.copy(), orb.copy())
    dev = od_pipe.prepare_window(det.copy(), orb.copy(), device=0)
    # integer outputs and everything that does not go through the per-row kernel: exact
    assert np.array_equal(dev.mask, host.mask) and np.array_equal(dev.ii, host.ii) and np.array_equal(dev.time_idx, host.time_idx)
    for key in ("landmarks_uv", "confidences", "intrinsics", "poses_gt", "cumrot_last", "vel_gt_full", "omega_gt"):
        assert np.array_equal(getattr(dev, key), getattr(host, key)), key
    # positions of ~6400 km: the device library's sin / cos against the host's
    assert np.abs(dev.landmarks_xyz - host.landmarks_xyz).max() < 1e-11
    assert rel_err(dev.landmarks_xyz, host.landmarks_xyz) < 1e-14
    assert np.abs(dev.extras["proj_gt"] - host.extras["proj_gt"]).max() < 1e-6


def test_outlier_mask_on_the_device_matches_the_host_rule():
    """Rows pushed out of the image, beyond 1000 px of their measurement and below the confidence bar (od_pipe.py:930)."""
    from vinsat_amd import od_pipe, synth
    det, orb = synth.make_sequence("C2")
    det = det.copy()
    rng = np.random.default_rng(4)
    k = rng.choice(det.shape[0], size=600, replace=False)
    det[k[:200], 5] = 0.5                                # low confidence
    det[k[200:400], 3] += 1500.0                         # measurement far from the reprojection
    det[k[400:], 2] += 4.0                               # latitude moved: the landmark leaves the image
    host = od_pipe.prepare_window(det.copy(), orb.copy())
    dev = od_pipe.prepare_window(det.copy(), orb.copy(), device=0)
    assert host.mask.sum() < det.shape[0] - 500
    assert np.array_equal(dev.mask, host.mask) and np.array_equal(dev.ii, host.ii) and np.array_equal(dev.time_idx, host.time_idx)


def test_driver_with_device_rows_reproduces_the_reference_run():
    """streaming_version with the HIP BA prepares its rows on the device by default: the reference's own results for C1."""
    from conftest import load_golden
    from vinsat_amd import od_pipe, synth
    g = load_golden("c1")
    det, orb = synth.make_sequence("C1")
    errors, first_det, times = od_pipe.streaming_version(detections=det, orbit_np=orb)
    assert rel_err(errors.numpy(), g["errors"]) < 1e-5 and int(first_det) == int(g["first_detection"])
