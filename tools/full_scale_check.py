"""Row-level parity at full size of what the headline actually runs.

For each workload the DEFAULT path (sigma_omega = 0: sweep -> k_walk with the fp32-decided clicks, the integer-decided
OrganicUserEventCounter act and the float64 sums of the parked users on both float64 pipes; sigma_omega > 0: certified
fp16-split MFMA sweeps, the run in rounds: k_advance_run) is compared with the float64-only lock-step path — RECOGYM_DRAW=f64
RECOGYM_WALK=0 RECOGYM_RUN_AHEAD=0 with the
click-probability export on, which forces every click through float64 — on the counters and on an order-independent
checksum of EVERY log row (Simulator.log_digest).  The float64-only path is the oracle's arithmetic on the device
(tests/test_hip_parity.py pins it on the oracle and on the reference's logs at small sizes).

c5 (both arms of BASELINE config 5): the frozen-LogReg arm's default act (fp16 screen + float64 refinement of the candidates)
is compared with RECOGYM_LOGREG=fp32 (round 2's certified fp32 scores + float64 fallback) on top of the float64 draws.

    python tools/full_scale_check.py [c3 c2 c4shard c5] [--users N]      -> one JSON line per run + a verdict line
"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recogym_amd.sim import Simulator, default_log_capacity

users_override = int(sys.argv[sys.argv.index('--users') + 1]) if '--users' in sys.argv else 0
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith('--') and sys.argv[i - 1] != '--users']
workloads = args or ['c3', 'c2', 'c4shard']
FAST = {}                                                           # the default path
# float64 draws, lock-step with an event per launch (no run-ahead rounds), float64 clicks
EXACT = {'RECOGYM_DRAW': 'f64', 'RECOGYM_WALK': '0', 'RECOGYM_LOGREG': 'fp32', 'RECOGYM_RUN_AHEAD': '0'}
ok_all = True
for wl, arm, kw in [(w, a, k) for w in workloads for a, k in bench.arms_of(w, bench.make_config(w))]:
    per_gpu = bench.WORKLOADS[wl][1]
    n = users_override or per_gpu
    cfg = bench.make_config(wl)
    out = {}
    for name, env, p_click in (('default', FAST, False), ('float64', EXACT, True)):
        for k in EXACT:
            os.environ.pop(k, None)
        os.environ.update(env)
        sim = Simulator(cfg, n, device='cuda:0', log_capacity=default_log_capacity(cfg, n), p_click=p_click, **kw)
        sim.reset_users(0, n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.run()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        c = sim.counters()
        out[name] = dict(workload=wl, arm=arm, path=name, env=env, p_click_export=p_click, users=n, seconds=round(dt, 3),
                         counters={k: c[k] for k in ('organic', 'bandit', 'clicks', 'phantom', 'exact_draws', 'exact_sweeps', 'anchored', 'memo_hits',
                                                     'lr_acts', 'lr_exact', 'exact_overflow', 'hist_overflow', 'log_dropped', 'live')},
                         digest=sim.log_digest())
        print(json.dumps(out[name]), flush=True)
        sim.close(); del sim; torch.cuda.empty_cache()
    same = out['default']['digest'] == out['float64']['digest'] and all(
        out['default']['counters'][k] == out['float64']['counters'][k] for k in ('organic', 'bandit', 'clicks', 'phantom')) and all(
        out[p]['counters'][k] == 0 for p in out for k in ('exact_overflow', 'hist_overflow', 'log_dropped', 'live'))
    ok_all = ok_all and same
    print(json.dumps(dict(workload=wl, arm=arm, users=n, rows=out['default']['counters']['organic'] + out['default']['counters']['bandit'],
                          verdict='IDENTICAL LOGS' if same else 'MISMATCH')), flush=True)
sys.exit(0 if ok_all else 1)
