"""Times the C port (oracle/recogym_oracle.c) on a bounded sample of a configuration — TEST INFRASTRUCTURE (imports the oracle);
tools/time_reference.py calls it so that the NumPy reference and the port are timed on the same box and the same samples."""
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
