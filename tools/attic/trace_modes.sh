#!/bin/bash
# rocprofv3 kernel traces of the chained batched schedule at given "W mode solver" triples (diagnostic; on the GPU box):
#   bash tools/trace_modes.sh "22 lat part" "64 bw part" ...
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  export VBA_MODE=$2 VBA_SOLVER=$3      # ("auto auto": the handle's own choices)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4t_w$1_$2_$3 -- python3 $R/tools/batched_chain.py $1 3 > $R/gpurun_out/r4t_w$1_$2_$3.out 2>&1
  echo "done $cfg"
done
