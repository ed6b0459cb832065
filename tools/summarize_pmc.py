#!/usr/bin/env python3
"""Summarise rocprofv3 output collected by tools/profile_pmc.sh into profiles/<tag>_*.{csv,json}.

  <tag>_kernel_stats.csv   the --kernel-trace --stats summary, verbatim
  <tag>_traffic.json       per kernel: calls, average duration, HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes):
                           on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM
                           section); WRITE_SIZE is exact.  Both figures are given.
  <tag>_sq_counters.json   per kernel, per launch: waves, wave-cycles, VALU instructions, share of the wave-cycles spent
                           issuing VALU / issuing anything / parked in s_waitcnt / stalled at issue
  <tag>_mfma.json          per kernel with matrix-core instructions: MFMA instructions and flops per launch, matrix-pipe busy
                           cycles, and (with the kernel durations) flop/s against the fp64 matrix peak
usage: tools/summarize_pmc.py <tag> [n_observation_rows_per_launch for the per-observation figures]
"""
import csv, glob, json, os, sys
from collections import defaultdict

MFMA_F64_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix (vendor figure; the guide's MFMA table has no f64 row)


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    return name.replace("void ", "").replace("vba::", "").strip()


def newest(files):
    return sorted(files, key=os.path.getmtime)[-1:]     # gpurun_out accumulates older runs


def counters(root, what):
    """{kernel: {counter: mean per launch}}"""
    acc = defaultdict(lambda: defaultdict(list))
    for fn in newest(glob.glob(os.path.join(root, what, "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = defaultdict(dict)
        for r in csv.DictReader(open(fn)):
            per_dispatch[(r["Dispatch_Id"], short(r["Kernel_Name"]))][r["Counter_Name"]] = float(r["Counter_Value"])
        for (_, k), cs in per_dispatch.items():
            for c, v in cs.items():
                acc[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    tag = sys.argv[1]
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    root = os.path.join("gpurun_out", tag)
    os.makedirs("profiles", exist_ok=True)
    stats = {}
    ks = newest(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True))
    if ks:
        with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
            f.write(open(ks[0]).read())
        for r in csv.DictReader(open(ks[0])):
            stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"]))
    fetch, write = counters(root, "fetch"), counters(root, "write")
    if fetch or write:
        out = {}
        for k in sorted(set(stats) | set(fetch)):
            e = dict(stats.get(k, {}))
            f, w = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
            if f is not None and w is not None:
                e.update(FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, hbm_bytes_per_launch=(2 * f + w) * 1024, hbm_bytes_per_launch_uncorrected=(f + w) * 1024)
            out[k] = e
        json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
    sq = counters(root, "sq")
    if sq:
        out = {}
        for k, c in sq.items():
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            e = dict(c)
            if wc > 0:      # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* all count quad-cycles: the ratios are unit free
                e.update(valu_issue_share=c.get("SQ_ACTIVE_INST_VALU", 0) / wc, any_issue_share=c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                         waitcnt_share=c.get("SQ_WAIT_ANY", 0) / wc, issue_stall_share=c.get("SQ_WAIT_INST_ANY", 0) / wc)
            if c.get("SQ_WAVES"):
                e["valu_insts_per_wave"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"]
            if rows and c.get("SQ_INSTS_VALU"):
                e["valu_wave_insts_per_64_rows"] = c["SQ_INSTS_VALU"] / (rows / 64.0)
            if k in stats:
                e["avg_us"] = stats[k]["avg_us"]
            out[k] = e
        json.dump(out, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
    mf = counters(root, "mfma")
    if mf:
        out = {}
        for k, c in mf.items():
            if not c.get("SQ_INSTS_VALU_MFMA_F64") and not c.get("SQ_INSTS_MFMA"):
                continue
            e = dict(c)
            # v_mfma_f64_16x16x4: 16 * 16 * 4 multiply-adds = 2048 flops per wave instruction; MOPS counts 512-flop units
            insts = c.get("SQ_INSTS_VALU_MFMA_F64", 0.0)
            e["mfma_flops_per_launch"] = insts * 2048.0
            e["mfma_flops_from_mops"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
            e["mfma_share_of_valu_insts"] = insts / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None
            if k in stats:
                t = stats[k]["avg_us"] * 1e-6
                e["avg_us"] = stats[k]["avg_us"]
                e["mfma_TFLOPs"] = e["mfma_flops_per_launch"] / t / 1e12
                e["frac_of_f64_matrix_peak"] = e["mfma_TFLOPs"] / MFMA_F64_PEAK_TFLOPS
            if c.get("SQ_BUSY_CU_CYCLES"):
                e["mfma_busy_share_of_cu_busy_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CU_CYCLES"]
            out[k] = e
        out["_peak_TFLOPs"] = MFMA_F64_PEAK_TFLOPS
        json.dump(out, open(f"profiles/{tag}_mfma.json", "w"), indent=1)
    for k, e in sorted(stats.items(), key=lambda kv: -kv[1].get("pct", 0))[:14]:
        print(k, e)


if __name__ == "__main__":
    main()
