#!/usr/bin/env python3
"""One-off stress: the randomised GPU parity test (tests/test_gpu_parity.py::test_randomised_windows_vs_oracle) over a
range of seeds.  usage: tools/stress_random.py first last   (STRESS_LONG=1: with long gaps among the random ones)"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    try:
        T.test_randomised_windows_vs_oracle(seed, long_gaps=bool(os.environ.get("STRESS_LONG")))
    except Exception as ex:
        bad.append(seed)
        print(f"seed {seed}: {type(ex).__name__}: {str(ex)[:200]}", flush=True)
        if os.environ.get("STRESS_TRACE"):
            traceback.print_exc()
    if seed % 20 == 0:
        print(f"... seed {seed}, failures so far {bad}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
