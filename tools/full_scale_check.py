"""BASELINE config 3 at full size, twice: certified fp16-split MFMA draws vs float64-only draws
(the oracle's arithmetic).  Prints counters and an order-independent checksum of every log row."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recogym_amd import _abi
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.reco_env_v1 import env_1_args
from recogym_amd.sim import Simulator, default_log_capacity
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': 10000, 'K': 20, 'sigma_omega': 0.0})
out = {}
for mode in ('f16', 'f64'):
    os.environ['RECOGYM_DRAW'] = mode
    sim = Simulator(cfg, n, device='cuda:0', log_capacity=default_log_capacity(cfg, n),
                    policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=42,
                    ouc=dict(select_randomly=True, epsilon=0.0, exploit_explore=True, reverse_pop=False))
    sim.reset_users(0, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sim.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = sim.counters()
    chk = [0, 0, 0, 0]
    step = 1 << 26
    for lo in range(0, c['log_rows'], step):
        rows = sim.log[lo:min(lo + step, c['log_rows'])].to(torch.int64)
        for i in range(4):
            chk[i] = (chk[i] + int((rows[:, i] * (rows[:, 1] + 7) * (rows[:, 0] + 13)).sum().item())) % (1 << 64)
    out[mode] = dict(seconds=round(dt, 2), counters={k: c[k] for k in ('organic', 'bandit', 'clicks', 'phantom', 'log_rows', 'exact_draws')}, checksum=chk)
    print(mode, json.dumps(out[mode]), flush=True)
    sim.close(); del sim; torch.cuda.empty_cache()
same = out['f16']['checksum'] == out['f64']['checksum'] and all(
    out['f16']['counters'][k] == out['f64']['counters'][k] for k in ('organic', 'bandit', 'clicks', 'phantom', 'log_rows'))
print('IDENTICAL LOGS' if same else 'MISMATCH')
