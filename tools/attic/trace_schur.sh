#!/bin/bash
# rocprofv3 kernel trace (+ MFMA counters) of the free-landmark Schur add-on at 2000 poses / 12000 x 12000 reduced system
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04_schur
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_schur/trace -- python3 $R/tools/schur_bench.py 2000 60000 3 > $R/gpurun_out/r04_schur/trace.out 2>&1
if [ "$1" = "mfma" ]; then
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/r04_schur/mfma -- python3 $R/tools/schur_bench.py 2000 60000 3 > $R/gpurun_out/r04_schur/mfma.out 2>&1
fi
