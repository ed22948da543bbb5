mkdir -p gpurun_out/r2_pmc_c4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/r2_pmc_c4/$name -- python $R/bench.py --workload c4shard --users 200000 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r2_pmc_c4/$name.out 2> $R/gpurun_out/r2_pmc_c4/$name.err
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC
cd $R/gpurun_out/r2_pmc_c4
python - <<'PY'
import csv, glob, collections
for name in ('sq1', 'sq2'):
    fs = glob.glob(f'{name}/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(name, 'no counter file', open(name + '.err').read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'][:52]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in agg:
        if 'k_draw_f16w' in k or 'k_exact_sums' in k:
            print(name, k, {c: f'{v:.4g}' for c, v in agg[k].items()})
PY
rm -rf sq1 sq2
