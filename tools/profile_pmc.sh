#!/bin/bash
# Collect rocprofv3 evidence on the GPU box, one pass per counter group (the blocks' slots do not fit more: TCC holds either
# FETCH_SIZE or WRITE_SIZE, SQ eight counters; MI355X_MICROARCH.md "rocprofv3 PMC slots"); --kernel-trace --stats in a run
# of its own, never combined with --pmc.  Output under gpurun_out/<tag>/<pass>/.
# usage: tools/profile_pmc.sh <tag> <passes: comma list of trace,fetch,write,sq,mfma> <python script + args...>
set -e
TAG=$1; shift
PASSES=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in ${PASSES//,/ }; do
  case $P in
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.out 2> $OUT/trace.err ;;
    fetch) rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 "$@" > $OUT/fetch.out 2> $OUT/fetch.err ;;
    write) rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 "$@" > $OUT/write.out 2> $OUT/write.err ;;
    sq)    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python3 "$@" > $OUT/sq.out 2> $OUT/sq.err ;;
    mfma)  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/mfma -- python3 "$@" > $OUT/mfma.out 2> $OUT/mfma.err ;;
    *) echo "unknown pass $P"; exit 2 ;;
  esac
  echo "pass $P done"
done
find $OUT -name "*.csv" | wc -l
