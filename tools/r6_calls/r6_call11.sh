# Round 6, GPU call 11: C3 at 21.3 M users on one GPU (raw log past 2^31 rows), 2 000 sampled users replayed by the oracle.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout 1500 python tools/beyond_2_31_run.py 21300000 --out $O/oracle_spot_check_21M_users.jsonl > $O/beyond_2_31_run.txt 2>&1
tail -5 $O/beyond_2_31_run.txt | cut -c1-600
