# Round 5, GPU call 14: k_sweep_xh with 8 waves per block (one block per CU: 256 users share one stream of table tiles) against 4.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
RECOGYM_XH_WAVES=8 timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pipelined_walk or walk_certificate or every_K_class or product_counts_around or error_free_sweep" 2>&1 | tail -4 > $O/gpu_tests_call14.txt
rm -f $O/ab_call14_xh_waves.jsonl
for w in 4 8 4 8; do RECOGYM_XH_WAVES=$w timeout 90 python tools/xh_probe.py 2000000 waves$w 2>>$O/ab14.err | tail -1 >> $O/ab_call14_xh_waves.jsonl; done
