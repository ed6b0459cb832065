#!/bin/bash
# rocprofv3 kernel-trace averages of the single-window chain for two builds of the library (A = the tree's, B = $1): per-kernel
# times resolve differences that the wall clock of a whole schedule does not (box noise is ~0.5 us per call).
# usage (on the GPU box): bash tools/kernel_ab.sh /root/repo/ab/libold.so
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/kab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/ab_fusion.py 15 > $OUT/a.out 2>&1
export VBA_LIB=$1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/ab_fusion.py 15 > $OUT/b.out 2>&1
python3 - <<PY
import csv, glob
def load(d):
    f = glob.glob("$OUT/" + d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1000) for r in csv.DictReader(open(f))}
a, b = load("a"), load("b")
for k in a:
    if k in b and a[k][0] > 500:
        print(f"{k[:72]:72s} calls {a[k][0]:6d}  A {a[k][1]:7.3f}  B {b[k][1]:7.3f}  A-B {a[k][1] - b[k][1]:+.3f} us")
PY
