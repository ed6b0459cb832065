#!/usr/bin/env python3
"""cProfile of the driver's data preparation of a C3 sequence with the per-row part on the device (diagnostic)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vinsat_amd import od_pipe, synth
det, orb = synth.make_sequence("C3")
for _ in range(3):
    od_pipe.SequenceRun(det.copy(), orb.copy(), device=0)
t0 = time.perf_counter()
for _ in range(10):
    od_pipe.SequenceRun(det.copy(), orb.copy(), device=0)
print("SequenceRun", round(1e2 * (time.perf_counter() - t0), 3), "ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    r = od_pipe.SequenceRun(det.copy(), orb.copy(), device=0); r.next_patch()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
