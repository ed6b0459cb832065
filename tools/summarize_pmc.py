#!/usr/bin/env python3
"""Summarise rocprofv3 output collected by tools/profile_pmc.sh into profiles/<tag>_*.{csv,json}.

HBM traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports half of the bytes of
wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Both figures are given.
"""
import csv, glob, json, os, sys
from collections import defaultdict

def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("vba::", "").strip()

def main():
    tag, windows = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
    root = os.path.join("gpurun_out", tag)
    out = {}
    newest = lambda files: sorted(files, key=os.path.getmtime)[-1:]     # gpurun_out accumulates older runs
    ks = newest(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True))
    stats = {}
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        os.makedirs("profiles", exist_ok=True)
        with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
            f.write(open(ks[0]).read())
        for r in rows:
            stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"]))
    pmc = {}
    for what in ("fetch", "write"):
        files = newest(glob.glob(os.path.join(root, what, "**", "*counter_collection.csv"), recursive=True))
        acc = defaultdict(list)
        for fn in files:
            for r in csv.DictReader(open(fn)):
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        pmc[what] = {k: sum(v) / len(v) for k, v in acc.items()}
    for k in sorted(set(stats) | set(pmc.get("fetch", {}))):
        f = pmc.get("fetch", {}).get(k)
        w = pmc.get("write", {}).get(k)
        e = dict(stats.get(k, {}))
        if f is not None and w is not None:
            e.update(FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, hbm_bytes_per_launch=(2 * f + w) * 1024, hbm_bytes_per_launch_uncorrected=(f + w) * 1024)
        out[k] = e
    json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
    for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("pct", 0)):
        print(k, e)

if __name__ == "__main__":
    main()
