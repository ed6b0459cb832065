cd $GRAFT_REPO_ROOT
export VBA_SWEEP_CONFIG=C4
for f in 15 14 12; do echo "## lat f$f"; VBA_SWEEP_FUSION=$f VBA_SWEEP_COMBOS=lat:part python tools/mode_sweep.py 1 2 4 6 8 12 16; done
echo "## bw"; VBA_SWEEP_COMBOS=bw:part python tools/mode_sweep.py 2 4 6 8 12 16
echo "## auto"; VBA_SWEEP_COMBOS=auto:auto python tools/mode_sweep.py 1 2 4 6 8 12 16
