cd $GRAFT_REPO_ROOT
export VBA_SWEEP_CONFIG=C2
echo "## default"; VBA_SWEEP_COMBOS=auto:auto python tools/mode_sweep.py 256 512 1000
for c in 2 4 8 12 16; do echo "## chunk $c"; VBA_SWEEP_CHUNK=$c VBA_SWEEP_COMBOS=bw:part python tools/mode_sweep.py 256 512 1000; done
export VBA_SWEEP_CONFIG=C3
for c in 12 14; do echo "## C3 chunk $c"; VBA_SWEEP_CHUNK=$c VBA_SWEEP_COMBOS=bw:part python tools/mode_sweep.py 32 128 1000; done
echo "## C3 chunk 8"; VBA_SWEEP_CHUNK=8 VBA_SWEEP_COMBOS=bw:part python tools/mode_sweep.py 32 128 1000
