"""Timing ablations of the organic draw kernels (results are wrong when RECOGYM_ABLATE != 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.reco_env_v1 import env_1_args
from recogym_amd.sim import Simulator
P, K, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for ab in [int(x) for x in sys.argv[4].split(',')]:
    os.environ['RECOGYM_ABLATE'] = str(ab)
    cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': P, 'K': K, 'sigma_omega': 0.0})
    sim = Simulator(cfg, n, device='cuda:0', log_capacity=0)
    for rep in range(2):
        sim.reset_users(0, n)
        sim.set_profiling(True)
        sim.step()
        pr = sim.profile()
        sim.set_profiling(False)
    c = sim.counters()
    tf = 2.0 * P * K * n / (pr['draw_mfma_ms'] * 1e-3) / 1e12
    print(f'ablate={ab}: mfma {pr["draw_mfma_ms"]:.2f} ms ({tf:.1f} TF)  exact {pr["draw_exact_ms"]:.2f} ms  advance {pr["advance_ms"]:.2f} ms', flush=True)
    sim.close()
