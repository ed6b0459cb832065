#!/bin/bash
# rocprofv3 kernel traces of the chained one-window schedule of other BASELINE configs (diagnostic; on the GPU box):
#   bash tools/trace_configs.sh C4 C5
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  export VBA_CONFIG=$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4c_$cfg -- python3 $R/tools/batched_chain.py 1 10 > $R/gpurun_out/r4c_$cfg.out 2>&1
  echo "done $cfg"
done
