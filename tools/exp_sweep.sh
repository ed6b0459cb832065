cd $GRAFT_REPO_ROOT
export VBA_SWEEP_CONFIG=C2
for f in 15 14 12; do
echo "## lat:part fusion $f"; VBA_SWEEP_FUSION=$f VBA_SWEEP_COMBOS=lat:part python tools/mode_sweep.py 1 4 16 32 64 128 256 512
done
echo "## bw:part"; VBA_SWEEP_COMBOS=bw:part python tools/mode_sweep.py 4 16 32 64 128 256 512 1024 4096 16384
echo "## bw:seq"; VBA_SWEEP_COMBOS=bw:seq python tools/mode_sweep.py 64 256 1024 4096 16384
export VBA_SWEEP_CONFIG=C3
echo "## C3 auto"; VBA_SWEEP_COMBOS=auto:auto python tools/mode_sweep.py 1 2 3 4 8 11 12 16 22 28 30 31 32 64 128 256 512 1024 1025 2048 4096
