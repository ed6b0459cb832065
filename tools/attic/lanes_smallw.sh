cd $GRAFT_REPO_ROOT
echo "## auto"; VBA_SWEEP_COMBOS=auto:auto python tools/mode_sweep.py 2 3 4 6 8 12
for l in 64 32 16; do echo "## lanes $l"; VBA_SWEEP_LANES=$l VBA_SWEEP_COMBOS=auto:auto python tools/mode_sweep.py 2 3 4 6 8 12; done
