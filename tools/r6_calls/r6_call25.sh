# Round 6, GPU call 25: what the waves of k_draw_tp / k_draw_tpw wait for — SQ counter passes (own runs, --kernel-trace only) on
# one unsliced step (tools/tp_probe.py 2 M users of c3drift; tools/wide_step0.py 500 k users of C4), and k_draw_tp forced to ONE
# block per CU (-DRG_TP_SMEM_PAD=8192 host build): what the second block is worth.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
rm -f $O/ab_call25_occ.jsonl
timeout 300 python tools/tp_probe.py 2000000 default 2>/dev/null | tail -1 >> $O/ab_call25_occ.jsonl
RECOGYM_HIP_LIB=$R/recogym_amd/csrc/librecogym_hip_pad8k.so timeout 300 python tools/tp_probe.py 2000000 one_block_per_cu 2>/dev/null | tail -1 >> $O/ab_call25_occ.jsonl
cat $O/ab_call25_occ.jsonl
cd /tmp && export TMPDIR=/tmp
pmc() { # name, counters, cmd...
  name=$1; cnt=$2; shift; shift
  timeout 600 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/$name -o run -- "$@" > $O/$name.out 2> $O/$name.err
  f=$(find $O/$name -name '*counter_collection.csv' | head -1)
  python - "$f" "$O/${name}_counters.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'k_draw_tp' not in k and 'k_pick' not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id'])); n[k] += 1; dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
with open(sys.argv[2], 'w') as f:
    f.write('kernel,dispatches,total_ns,counter,value\n')
    for k in sorted(agg, key=lambda k: -dur[k]):
        for c, v in sorted(agg[k].items()):
            f.write(f'"{k[:60]}",{n[k]},{dur[k]},{c},{v:.6g}\n')
PY
  rm -rf $O/$name $O/$name.out
}
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU"
C="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL"
pmc pmc_call25_tp_a "$A" python $R/tools/tp_probe.py 2000000
pmc pmc_call25_tp_b "$B" python $R/tools/tp_probe.py 2000000
pmc pmc_call25_tp_c "$C" python $R/tools/tp_probe.py 2000000
pmc pmc_call25_tpw_a "$A" python $R/tools/wide_step0.py 500000
pmc pmc_call25_tpw_b "$B" python $R/tools/wide_step0.py 500000
pmc pmc_call25_tpw_c "$C" python $R/tools/wide_step0.py 500000
head -40 $O/pmc_call25_tp_a_counters.csv
