"""Times the C port (oracle/recogym_oracle.c) on the samples tools/time_reference.py timed the NumPy reference on — TEST
INFRASTRUCTURE (imports the oracle) — and adds `port_same_box` + the port / NumPy ratios to profiles/r4/numpy_reference_cpu.json, so
that both were timed on the same box.

    python tools/time_reference.py && python tests/port_timing.py"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def port_case(env_over, kind, users, threads):
    """`threads` oracle instances on disjoint id ranges (ctypes releases the GIL) -> (events, wall seconds)."""
    from oracle import oracle as orc
    from recogym_amd import _abi
    from recogym_amd.envs.configuration import Configuration
    from recogym_amd.envs.reco_env_v1 import env_1_args
    cfg = Configuration({**env_1_args, 'random_seed': 42, **env_over})
    kw = {}
    if kind == 'random':
        kw = dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=42)
    elif kind == 'ouc':
        kw = dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=42,
                  ouc=dict(select_randomly=True, epsilon=0.0, exploit_explore=True, reverse_pop=False))
    orc.lib()
    res = [0] * threads

    def work(k):
        env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **kw)
        ev = 0
        for lo in range(0, users, 25):
            n = min(25, users - lo)
            rows = env.generate_logs(n, first_user_id=1_000_000 * k + lo, capacity=n * 2000 + 10000)
            ev += int((rows['phantom'] == 0).sum())
        res[k] = ev
    th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return sum(res), time.perf_counter() - t0


if __name__ == '__main__':
    path = os.path.join(ROOT, 'profiles', 'r4', 'numpy_reference_cpu.json')
    out = json.load(open(path))
    for name, c in out['cases'].items():
        n = c['processes']
        pe1, pw1 = port_case(c['env'], c['agent'], c['users'], 1)
        pe8, pw8 = port_case(c['env'], c['agent'], c['users'], n)
        c['port_same_box'] = dict(what='oracle/recogym_oracle.c (float64 C port, Philox draws) on the same sample in this container',
                                  one_thread_events_per_s=pe1 / pw1, one_thread_events=pe1, one_thread_seconds=pw1,
                                  threads=n, all_thread_events_per_s=pe8 / pw8, all_thread_wall_seconds=pw8,
                                  port_over_numpy_one_core=(pe1 / pw1) / c['one_core_events_per_s'],
                                  port_over_numpy_all_core=(pe8 / pw8) / c['all_core_events_per_s'])
        print(name, json.dumps(c['port_same_box']))
    json.dump(out, open(path, 'w'), indent=1)
