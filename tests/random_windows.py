"""Randomised window shapes shared by the GPU parity test, tools/stress_random.py and tools/conditioning_evidence.py."""
import numpy as np

SCHEDULE = [(0, True), (1, True), (3, True), (10, False), (12, False), (19, False)]


def make(seed, long_gaps=False):
    """Random window (2..70 poses, 0..60 rows per pose, gaps 1..60 s, confidences 0.3..1.2, shuffled rows, one pose
    without rows).  ``long_gaps``: up to four of the gaps become 65 .. 1300 s (the edges that are propagated parallel in time,
    vba_long.hip).  Returns (win, xyz, uv, ii, conf, time_idx, states0)."""
    from vinsat_amd import od_pipe, synth
    rng = np.random.default_rng(1000 + seed)
    n_target = int(rng.integers(2, 71))
    det, orb = synth.make_sequence(synth.WindowConfig("rnd", n_target, int(rng.integers(3, 61)), 5), seed=seed)
    win = od_pipe.prepare_window(det, orb)
    n = win.time_idx.size
    keep = rng.random(win.ii.size) < rng.uniform(0.3, 1.0)
    if n > 3:
        keep[win.ii == int(rng.integers(0, n))] = False            # one pose without rows
    if keep.sum() < 2:
        keep[:2] = True
    order = rng.permutation(np.nonzero(keep)[0])
    xyz, uv, ii = win.landmarks_xyz[order], win.landmarks_uv[order], win.ii[order]
    conf = rng.uniform(0.3, 1.2, size=ii.size)
    gaps = rng.integers(1, 61, size=n - 1)
    if long_gaps:
        lrng = np.random.default_rng(5000 + seed)
        for k in lrng.choice(n - 1, size=min(n - 1, int(lrng.integers(1, 5))), replace=False):
            gaps[k] = int(lrng.integers(65, 1301))
    t = np.cumsum(np.concatenate([[10], gaps])).astype(np.int64)
    return win, xyz, uv, ii, conf, t, od_pipe.initial_guess(win, seed=seed)
