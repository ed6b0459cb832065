#!/usr/bin/env python3
"""Kernel time against the gaps between consecutive kernels in a rocprofv3 --kernel-trace CSV (last N kernels): what a
dependent kernel boundary costs, launched kernel by kernel or replayed as a graph.  usage: gap_summary.py trace.csv [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 700
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
small = [g for g in gaps if g < 20000]          # (between schedules the host synchronises: not a kernel boundary)
print(f"{len(rows)} kernels: {dur / len(rows) / 1e3:.2f} us mean duration, {sum(small) / max(len(small), 1) / 1e3:.2f} us mean gap over {len(small)} boundaries "
      f"(median {sorted(small)[len(small) // 2] / 1e3:.2f}), {len(gaps) - len(small)} longer pauses")
