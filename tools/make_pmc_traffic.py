#!/usr/bin/env python
"""profiles/r2/pmc_traffic.json from the counter summaries tools/r2_profiles.sh leaves under gpurun_out/r2_prof/
(and copies of those summaries + kernel stats into profiles/r2/)."""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, 'gpurun_out', 'r2_prof')
P = os.path.join(ROOT, 'profiles', 'r2')


def val(name, kern, counter):
    for r in csv.DictReader(open(f'{O}/{name}_counters.csv')):
        if kern in r['kernel'] and r['counter'] == counter:
            return float(r['value'])
    raise KeyError((name, kern, counter))


def bench_line(name):
    return json.loads(open(f'{O}/{name}.out').read().strip().splitlines()[-1])


def entry(fetch_run, write_run, kern, units, unit_name, what):
    f, w = val(fetch_run, kern, 'FETCH_SIZE'), val(write_run, kern, 'WRITE_SIZE')
    return dict(unit_name=unit_name, units_in_profiled_run=units, fetch_size_kb=f, write_size_kb=w,
                hbm_bytes_per_unit=(2 * f + w) * 1024 / units, fetch_bytes_per_unit_uncorrected=f * 1024 / units,
                write_bytes_per_unit=w * 1024 / units,
                source=f'profiles/r2/{fetch_run}_counters.csv + {write_run}_counters.csv ({what}; offline rocprofv3 --pmc, scaled to the run)')


def main():
    os.makedirs(P, exist_ok=True)
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), one counter per pass, --kernel-trace only (tools/r2_profiles.sh). '
                   'hbm_bytes_per_unit = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / units: FETCH_SIZE doubled as MI355X_MICROARCH.md '
                   'prescribes for gfx950 (128-byte requests tallied at 64 B; an upper bound for narrower accesses); the counters sit '
                   'at the L2 <-> fabric boundary, Infinity-Cache hits included.  Measured on reduced runs (C3 / c3drift: 2 M users, '
                   'C4: 300 k users) as per-unit figures; bench.py scales them to the benched run.',
           'kernels': {}}
    d = bench_line('pmc_c3_fetch')
    out['kernels']['k_walk'] = entry('pmc_c3_fetch', 'pmc_c3_write', 'k_walk', d['config']['events_per_step'], 'events', 'C3, 2 M users')
    d = bench_line('pmc_c3drift_fetch')
    out['kernels']['k_draw_bf16p'] = entry('pmc_c3drift_fetch', 'pmc_c3drift_write', 'k_draw_bf16p', d['kernels']['draw_sweep']['units'],
                                           'swept draws', 'c3drift, 2 M users')
    d = bench_line('pmc_c4_fetch')
    out['kernels']['k_draw_*'] = entry('pmc_c4_fetch', 'pmc_c4_write', 'k_draw_f16w', d['kernels']['draw_sweep']['units'],
                                       'swept draws', 'c4shard, 300 k users')
    json.dump(out, open(os.path.join(P, 'pmc_traffic.json'), 'w'), indent=1)
    for k, v in out['kernels'].items():
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a != 'source'})
    for f in sorted(os.listdir(O)):
        if f.endswith('_kernel_stats.csv') or f.endswith('_counters.csv') or f.endswith('_bench_line.json'):
            shutil.copy(os.path.join(O, f), os.path.join(P, f))


if __name__ == '__main__':
    main()
