#!/usr/bin/env python
"""profiles/<round>/pmc_traffic.json from the counter summaries tools/r3_profiles.sh leaves under gpurun_out/<round>_prof/
(and copies of those summaries + kernel stats + bench lines into profiles/<round>/).

Every PMC pass runs `bench.py --single-run`: exactly ONE simulation per arm inside the profiled process, so the counter
value summed over a kernel's dispatches belongs to one run and is divided by THAT run's units (round 2 divided the sum
over two runs by one run's events: 2x too much)."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else 'r3'
O = os.path.join(ROOT, 'gpurun_out', f'{RND}_prof')
P = os.path.join(ROOT, 'profiles', RND)


def val(name, kern, counter):
    tot, disp = 0.0, 0
    for r in csv.DictReader(open(f'{O}/{name}_counters.csv')):
        if kern in r['kernel'] and r['counter'] == counter:
            tot += float(r['value'])
            disp += int(r['dispatches'])
    if not disp:
        raise KeyError((name, kern, counter))
    return tot, disp


def bench_line(name):
    return json.loads(open(f'{O}/{name}_bench_line.json').read().strip().splitlines()[-1])


def entry(workload, fetch_run, write_run, kern, units, unit_name, what):
    (f, nf), (w, nw) = val(fetch_run, kern, 'FETCH_SIZE'), val(write_run, kern, 'WRITE_SIZE')
    return dict(workload=workload, unit_name=unit_name, units_in_profiled_run=units, dispatches_in_profiled_run=nf,
                kernels_matched=kern + '*',
                fetch_size_kb=f, write_size_kb=w,
                hbm_bytes_per_unit=(2 * f + w) * 1024 / units, fetch_bytes_per_unit_uncorrected=f * 1024 / units,
                write_bytes_per_unit=w * 1024 / units,
                source=f'profiles/{RND}/{fetch_run}_counters.csv + {write_run}_counters.csv ({what}; rocprofv3 --pmc on bench.py '
                       f'--single-run: one run per pass)')


def main():
    os.makedirs(P, exist_ok=True)
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), one counter per pass, --kernel-trace only, on bench.py --single-run '
                   '(ONE simulation per pass).  hbm_bytes_per_unit = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / units of that run: FETCH_SIZE '
                   'doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 B; an upper bound for narrower '
                   'accesses); the counters sit at the L2 <-> fabric boundary, Infinity-Cache hits included.  Measured on reduced runs '
                   '(users stated per entry) as per-unit figures; bench.py scales them to the benched run.',
           'kernels': {}}
    def try_add(key, fn):
        try:
            out['kernels'][key] = fn()
        except Exception as e:      # a pass that was not taken this round
            print('skipped', key, repr(e))
    d = bench_line('pmc_c3_fetch')
    try_add('k_walk', lambda: entry('c3', 'pmc_c3_fetch', 'pmc_c3_write', 'k_walk', d['config']['events_per_step'], 'events',
                                    f"C3, {d['config']['users_per_gpu']} users; k_walk2 rounds + k_walk_solo"))
    # the sweep of the same run (k_sweep_xh since round 5) and what the walk kernels ISSUE (bench.py's issue_roofline)
    try_add('k_sweep_xh', lambda: entry('c3', 'pmc_c3_fetch', 'pmc_c3_write', 'k_sweep_xh', d['config']['users_per_gpu'], 'swept draws',
                                        f"C3, {d['config']['users_per_gpu']} users"))

    def issue():
        ds = bench_line('pmc_c3_sq')
        ev = ds['config']['events_per_step']
        (valu, nd), (salu, _) = val('pmc_c3_sq', 'k_walk', 'SQ_INSTS_VALU'), val('pmc_c3_sq', 'k_walk', 'SQ_INSTS_SALU')
        (wc, _), (wa, _) = val('pmc_c3_sq', 'k_walk', 'SQ_WAVE_CYCLES'), val('pmc_c3_sq', 'k_walk', 'SQ_WAIT_ANY')
        return dict(workload='c3', policy='ouc', unit_name='events', units_in_profiled_run=ev, dispatches_in_profiled_run=nd,
                    valu_wave_instr_per_unit=valu / ev, salu_wave_instr_per_unit=salu / ev, wait_any_over_wave_cycles=wa / wc,
                    source=f'profiles/{RND}/pmc_c3_sq_counters.csv (C3, {ds["config"]["users_per_gpu"]} users; SQ_INSTS_VALU / SQ_INSTS_SALU of '
                           f'k_walk2 + k_walk_solo over the run\'s events; rocprofv3 --pmc on bench.py --single-run)')
    try_add('k_walk_issue', issue)
    try:
        dd = bench_line('pmc_c3drift_fetch')
        try_add('k_draw_bf16p', lambda: entry('c3drift', 'pmc_c3drift_fetch', 'pmc_c3drift_write', 'k_draw_bf16p',
                                              dd['kernels']['draw_sweep']['units'], 'swept draws', f"c3drift, {dd['config']['users_per_gpu']} users"))
    except Exception as e:
        print('skipped c3drift', repr(e))
    try:     # round 6: the sweep without a scratch and its search on the matrix cores
        dd = bench_line('pmc_c3drift_fetch')
        for kern in ('k_draw_tp', 'k_pick', 'k_advance_run', 'k_exact_sums'):
            try_add(kern, lambda kern=kern: entry('c3drift', 'pmc_c3drift_fetch', 'pmc_c3drift_write', kern,
                                                  dd['kernels']['draw_sweep']['units'] if kern != 'k_advance_run' else dd['config']['events_per_step'],
                                                  'swept draws' if kern != 'k_advance_run' else 'events', f"c3drift, {dd['config']['users_per_gpu']} users"))
    except Exception as e:
        print('skipped c3drift (r6 kernels)', repr(e))
    try:
        d5 = bench_line('pmc_c5_fetch')
        k5 = d5['kernels']['logreg_ips_frozen.logreg_acts']
        try_add('k_logreg_select', lambda: entry('c5', 'pmc_c5_fetch', 'pmc_c5_write', 'k_logreg_', k5['units'], 'acts',
                                                 f"c5, {d5['config']['users_per_gpu']} users per arm; k_logreg_select + screen + decide"))
    except Exception as e:
        print('skipped c5', repr(e))
    json.dump(out, open(os.path.join(P, 'pmc_traffic.json'), 'w'), indent=1)
    for k, v in out['kernels'].items():
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a != 'source'})
    for f in sorted(os.listdir(O)):
        if f.endswith('_kernel_stats.csv') or f.endswith('_counters.csv') or f.endswith('_bench_line.json'):
            shutil.copy(os.path.join(O, f), os.path.join(P, f))


if __name__ == '__main__':
    main()
