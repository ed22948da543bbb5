import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.reco_env_v1 import env_1_args
from recogym_amd.sim import Simulator
for P, K, n, sig in [(10, 5, 20000, 0.1), (1000, 20, 20000, 0.0), (10000, 20, 20000, 0.0), (10000, 20, 20000, 0.1), (2000, 64, 5000, 0.1)]:
    cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': P, 'K': K, 'sigma_omega': sig})
    sim = Simulator(cfg, n, device='cuda:0', log_capacity=0)
    sim.reset_users(0, n); torch.cuda.synchronize(); t0 = time.time(); sim.run(); torch.cuda.synchronize(); dt = time.time() - t0
    c = sim.counters()
    ev = c['organic'] + c['bandit']
    print(f'P={P} K={K} sig={sig} n={n}: events={ev} organic={c["organic"]} exact={c["exact_draws"]} '
          f'amb_frac={c["exact_draws"]/max(c["organic"],1):.4f} time={dt:.3f}s  {ev/dt/1e6:.2f} M ev/s steps={c["step"]}')
    sim.close()
