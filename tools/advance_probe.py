"""Per-kernel time of the first N lock-step steps for different device policies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recogym_amd import _abi
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.reco_env_v1 import env_1_args
from recogym_amd.sim import Simulator
P, K, n, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': P, 'K': K, 'sigma_omega': float(sys.argv[5]) if len(sys.argv) > 5 else 0.0})
pols = {'none': {}, 'random': dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=1),
        'ouc': dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=1,
                    ouc=dict(select_randomly=True, epsilon=0.0, exploit_explore=True, reverse_pop=False))}
for name, kw in pols.items():
    sim = Simulator(cfg, n, device='cuda:0', log_capacity=n * (steps + 2), **kw)
    for rep in range(2):
        sim.reset_users(0, n)
        sim.set_profiling(True)
        for _ in range(steps):
            sim.step()
        pr = sim.profile()
        sim.set_profiling(False)
    c = sim.counters()
    ev = c['organic'] + c['bandit']
    print(f'{name:7s}: {ev/1e6:.1f} M events in {steps} steps: mfma {pr["draw_mfma_ms"]:.1f} ms exact {pr["draw_exact_ms"]:.1f} ms '
          f'advance {pr["advance_ms"]:.1f} ms ({pr["advance_ms"]*1e6/max(c["bandit"],1):.3f} ns/bandit event; bandit={c["bandit"]/1e6:.1f}M)', flush=True)
    sim.close()
