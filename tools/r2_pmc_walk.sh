# PMC counters of the walk kernel on a 2 M-user C3 run (rocprofv3 --pmc in its own passes, --kernel-trace only)
mkdir -p gpurun_out/r2_pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/r2_pmc/$name -- python $R/bench.py --workload c3 --users 2000000 --steps 1 --warmup 0 --no-cpu-baseline --no-drift-line > $R/gpurun_out/r2_pmc/$name.out 2> $R/gpurun_out/r2_pmc/$name.err
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp2 TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
cd $R/gpurun_out/r2_pmc
python - <<'PY'
import csv, glob, collections
for name in ('sq1', 'sq2', 'tcp', 'tcp2', 'tcc', 'ta'):
    fs = glob.glob(f'{name}/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(name, 'no counter file', open(name + '.err').read()[-300:].replace('\n', ' | ')); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in agg:
        if 'k_walk<' in k:
            print(name, k, {c: f'{v:.4g}' for c, v in agg[k].items()})
PY
rm -rf sq1 sq2 tcp tcp2 tcc ta
