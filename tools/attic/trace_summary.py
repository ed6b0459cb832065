#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --kernel-trace --stats directories: python tools/trace_summary.py <dir> ..."""
import csv, glob, sys
for d in sys.argv[1:]:
    import os
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]      # (the newest run of that directory)
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("==", d)
    for r in rows[:16]:
        print(f'{r["Name"][:78]:78s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"]) / 1e3:9.2f} us  {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
    print(f"sum of kernel time per BA call (80 calls traced): {tot / 1e3 / 80:.1f} us")
