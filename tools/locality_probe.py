"""Per-kernel cost per event in successive windows of lock-step steps (C3-like workload):
shows how the cost of the gather-bound kernels moves as the live lists thin out and scramble."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recogym_amd import _abi
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.reco_env_v1 import env_1_args
from recogym_amd.sim import Simulator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
win = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nwin = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': 10000, 'K': 20, 'sigma_omega': 0.0})
sim = Simulator(cfg, n, device='cuda:0', log_capacity=0, policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=1,
                ouc=dict(select_randomly=True, epsilon=0.0, exploit_explore=True, reverse_pop=False))
for rep in range(2):
    sim.reset_users(0, n)
    prev = dict(organic=0, bandit=0)
    for w in range(nwin):
        sim.set_profiling(True)
        for _ in range(win):
            sim.step()
        pr = sim.profile()
        sim.set_profiling(False)
        c = sim.counters()
        o, b = c['organic'] - prev['organic'], c['bandit'] - prev['bandit']
        prev = c
        if rep == 1:
            print(f'steps {w*win:4d}-{(w+1)*win:4d}: live {c["live"]/1e6:6.2f}M organic {o/1e6:6.1f}M bandit {b/1e6:6.1f}M | '
                  f'mfma {pr["draw_mfma_ms"]*1e6/max(o,1):.3f} search {pr["draw_search_ms"]*1e6/max(o,1):.3f} '
                  f'exact {pr["draw_exact_ms"]*1e6/max(o,1):.3f} ns/organic | advance {pr["advance_ms"]*1e6/max(o+b,1):.3f} ns/event', flush=True)
