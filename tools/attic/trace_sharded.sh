#!/bin/bash
# rocprofv3 kernel trace of the one-rank sharded schedule (library-issued exchanges): profiles/<tag>_sharded1_kernel_stats.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/sharded_native_rate.py 2>&1 | grep "us per"
VBA_SH_STEPPED=1 python3 $R/tools/sharded_native_rate.py 2>&1 | grep "us per"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_sharded1/trace -- python3 $R/tools/sharded_native_rate.py > $R/gpurun_out/r04_sharded1/trace.out 2>&1
