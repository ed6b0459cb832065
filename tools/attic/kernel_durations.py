#!/usr/bin/env python3
"""List the durations (us) of every launch of kernels whose name contains a substring, from a kernel_trace CSV."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), " ".join(f"{x:.0f}" for x in d[-40:]))
