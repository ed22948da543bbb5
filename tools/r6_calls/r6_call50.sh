# Round 6, GPU call 50: the oracle replays 2 000 sampled users of every benched workload AT FULL SIZE on the final tree
# (tests/oracle_spot_check.py): C3, c3drift (k_draw_tp + k_pick), C5 (the rebuilt LogReg act at 10^4 classes), a C4 shard (k_draw_tpw, no k_tail).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/oracle_spot_check_call50.jsonl
timeout 2400 python tests/oracle_spot_check.py c3 c3drift c5 c4shard --out $O/oracle_spot_check_call50.jsonl > $O/oracle_spot_check_call50.txt 2>&1
tail -12 $O/oracle_spot_check_call50.txt | cut -c1-250
