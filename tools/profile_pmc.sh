#!/bin/bash
# Collect rocprofv3 evidence on the GPU box: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes
# (TCC slots do not fit both, see MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output under gpurun_out/<tag>/.
# usage: tools/profile_pmc.sh <tag> <python script + args...>
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.out 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 "$@" > $OUT/fetch.out 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 "$@" > $OUT/write.out 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
