# A/B of two against three cyclic-reduction levels in front of the one-workgroup kernel (VBA_CR_LEVELS, comparison build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export VBA_LIB=$R/vinsat_amd/libvinsat_ba_variants.so
for l in 2 3; do
  export VBA_CR_LEVELS=$l
  python3 $R/tools/dump_states.py /tmp/s$l.npz > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4t_cr$l -- python3 $R/tools/single_chain.py 20 > $R/gpurun_out/r4t_cr$l.out 2>&1
done
python3 -c "
import numpy as np
a=np.load('/tmp/s2.npz'); b=np.load('/tmp/s3.npz')
print('equal bits:', all(np.array_equal(a[k], b[k]) for k in a.files))"
