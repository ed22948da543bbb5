"""A/B of the walked run's execution forms on one workload: every configuration (a set of RECOGYM_* switches read at
rg_sim_create) runs `--reps` times on the same users; one JSON line each with the run times, the per-kernel HIP-event
times of one more (profiled) run, the counters and — with --digest — the log checksum (must be equal across forms).

    python tools/pipe_probe.py [--workload c3] [--users N] [--reps 3] [--digest] CONFIG ...
    CONFIG = name:K=V,K=V   e.g.  serial:RECOGYM_PIPE=0  pipe4:RECOGYM_PIPE=4,RECOGYM_PIPE_MODE=1
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recogym_amd.sim import Simulator, default_log_capacity

argv = sys.argv[1:]
def opt(name, default):
    if name in argv:
        i = argv.index(name); v = argv[i + 1]; del argv[i:i + 2]; return v
    return default
workload = opt('--workload', 'c3')
users = int(opt('--users', bench.WORKLOADS[workload][1]))
reps = int(opt('--reps', 3))
digest = '--digest' in argv
if digest:
    argv.remove('--digest')
cfg = bench.make_config(workload)
kw = bench.arms_of(workload, cfg)[0][1]
first = None
for spec in argv:
    name, _, kv = spec.partition(':')
    env = dict(x.split('=') for x in kv.split(',') if x)
    os.environ.update(env)
    sim = Simulator(cfg, users, device='cuda:0', log_capacity=default_log_capacity(cfg, users), **kw)
    ms = []
    for r in range(reps + 1):
        sim.reset_users(0, users)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.run()
        torch.cuda.synchronize(); ms.append(round(1e3 * (time.perf_counter() - t0), 2))
    sim.set_profiling(True)
    sim.reset_users(0, users)
    sim.run()
    prof = {k: round(v, 2) if isinstance(v, float) else v for k, v in sim.profile().items()}
    sim.set_profiling(False)
    c = sim.counters()
    out = dict(config=name, env=env, workload=workload, users=users, run_ms=ms[1:], first_run_ms=ms[0], best_ms=min(ms[1:]), profile=prof,
               counters={k: c[k] for k in ('organic', 'bandit', 'clicks', 'phantom', 'exact_draws', 'exact_sweeps', 'anchored',
                                           'memo_hits', 'log_rows', 'log_dropped', 'live', 'step')})
    if digest:
        out['digest'] = sim.log_digest()
        first = first or out['digest']
        out['digest_equal_to_first'] = out['digest'] == first
    print(json.dumps(out), flush=True)
    sim.close(); del sim; torch.cuda.empty_cache()
    for k in env:
        os.environ.pop(k, None)
