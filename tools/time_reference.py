#!/usr/bin/env python
"""Times the UNMODIFIED NumPy reference (/root/reference, via tests/ref_harness.py's gym/numba shims) on bounded
samples of BASELINE.json's configurations, on the cores of the box this script runs on (SURVEY.md §8d "CPU
baseline timing").  /root/reference exists only in the build container, not on the GPU box, so the result is
committed as profiles/r4/numpy_reference_cpu.json and bench.py quotes it (with this provenance) beside the
C-port baseline it times live.  Sample sizes are SURVEY.md §8(d)'s: C1 all 1 000 users, C2 2 000, C3 200, C4's shape with P
capped at the reference's np.int16 ceiling on 50 users; one process, and 8 processes on disjoint seeds.  The C port
(oracle/recogym_oracle.c, float64) is timed IN THE SAME CONTAINER on the same samples (1 and 8 threads) by
`python tests/port_timing.py` right after this script (the oracle is test infrastructure: only code under tests/ runs it), which
adds `port_same_box` and the port / NumPy ratios to the same JSON: bench.py's live port timing on the GPU box's host cores divided
by that ratio is what the NumPy reference would do there.

    python tools/time_reference.py && python tests/port_timing.py            # ~5 minutes
"""
import json
import multiprocessing as mp
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)

CASES = {
    # name: (env overrides, agent kind, users in the sample)
    'c1': (dict(num_products=10, K=5, sigma_omega=0.0), None, 1000),
    'c2': (dict(num_products=1000, K=20, sigma_omega=0.0), 'random', 2000),
    'c3': (dict(num_products=10000, K=20, sigma_omega=0.0), 'ouc', 200),
    'c4_capped': (dict(num_products=32767, K=64, sigma_omega=0.1), None, 50),   # reference ceiling: np.int16 ids
}


def run_case(name, seed=42):
    import ref_harness as rh
    rh.import_reference()
    import make_golden as mg
    over, kind, users = CASES[name]
    args = {**mg.BASE, 'random_seed': seed, **over}
    env = rh.make_reference_env(args)
    agent = mg.make_agent(kind, dict(num_products=args['num_products'], random_seed=seed)) if kind else None
    t0 = time.perf_counter()
    df = env.generate_logs(users, agent)
    dt = time.perf_counter() - t0
    # real events = rows minus the trailing phantom row of every user
    events = len(df) - users
    return events, dt


def _worker(a):
    return run_case(*a)


def main():
    out = dict(host=platform.node(), cpu_count=os.cpu_count(), python=platform.python_version(),
               numba='absent (sig/ff run as NumPy; affects reco_env_v1.py:32-41 only)',
               what='unmodified /root/reference recogym via tests/ref_shims (gym, numba stubs), env.generate_logs',
               cases={})
    import numpy
    out['numpy'] = numpy.__version__
    for name in CASES:
        ev, dt = run_case(name)
        n = min(os.cpu_count() or 1, 8)
        t0 = time.perf_counter()
        with mp.Pool(n) as pool:
            res = pool.map(_worker, [(name, 1000 + i) for i in range(n)])
        wall = time.perf_counter() - t0
        out['cases'][name] = dict(env=CASES[name][0], agent=CASES[name][1], users=CASES[name][2],
                                  one_core_events_per_s=ev / dt, one_core_events=ev, one_core_seconds=dt,
                                  processes=n, all_core_events_per_s=sum(r[0] for r in res) / wall,
                                  all_core_wall_seconds=wall)
        print(name, json.dumps(out['cases'][name]))
    path = os.path.join(ROOT, 'profiles', 'r4', 'numpy_reference_cpu.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path, '- now run: python tests/port_timing.py')


if __name__ == '__main__':
    main()
