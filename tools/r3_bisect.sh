R=$GRAFT_REPO_ROOT
cd $R
T="tests/test_hip_parity.py -k sigma_omega_zero_sum_cache_matches_the_oracle -q -x"
echo "== occ3"; RECOGYM_WALK_OCC=3 python -m pytest $T 2>&1 | tail -3
echo "== occ4 omreg0"; RECOGYM_WALK_OMREG=0 python -m pytest $T 2>&1 | tail -3
echo "== occ4 omreg1 solo0"; RECOGYM_WALK_SOLO=0 python -m pytest $T 2>&1 | tail -3
echo "== occ3 solo0"; RECOGYM_WALK_OCC=3 RECOGYM_WALK_SOLO=0 python -m pytest $T 2>&1 | tail -3
