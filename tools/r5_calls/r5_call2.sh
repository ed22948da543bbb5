# Round 5, GPU call 2: where k_sweep_xh's time goes — timing builds (-DRG_XH_ABL bits, results wrong by design), 2 M users; the
# tightened delta on C3; SQ counters of the sweep.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd $R
rm -f $O/ab_call2_xh_ablation.jsonl
for v in 0 32 1 2 3 4 8 16 64 127; do
  lib=$R/recogym_amd/csrc/librecogym_hip_xhabl$v.so
  [ $v = 0 ] && lib=$R/recogym_amd/csrc/librecogym_hip.so
  RECOGYM_HIP_LIB=$lib timeout 90 python tools/xh_probe.py 2000000 abl$v 2>>$O/ab2.err | tail -1 >> $O/ab_call2_xh_ablation.jsonl
done
RECOGYM_XH=0 timeout 90 python tools/xh_probe.py 2000000 old_kernel 2>>$O/ab2.err | tail -1 >> $O/ab_call2_xh_ablation.jsonl
timeout 120 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-drift-line --no-materialise --no-other-workloads 2>>$O/ab2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(name='c3_xh_delta2', ms_per_step=round(d['ms_per_step'],2), kernels={k:[v['ms'], v.get('units')] for k,v in d['kernels'].items()})))" > $O/ab_call2_c3.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_xh_sq1 -- python $R/tools/xh_probe.py 2000000 pmc1 > $O/pmc_xh_sq1.out 2>>$O/ab2.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc_xh_sq2 -- python $R/tools/xh_probe.py 2000000 pmc2 > $O/pmc_xh_sq2.out 2>>$O/ab2.err
python - <<'P'
import csv,glob,collections,json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r5'
for d in ('pmc_xh_sq1','pmc_xh_sq2'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    with open(f'{O}/{d}_summary.json','w') as fo:
        json.dump({k:dict(v) for k,v in agg.items() if 'sweep' in k or 'draw' in k or 'walk2' in k}, fo, indent=1)
P
find $O -name "*.csv" -size +2M -delete
