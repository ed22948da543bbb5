"""Fraction of organic draws the MFMA kernel could not certify, per shape and kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.reco_env_v1 import env_1_args
from recogym_amd.sim import Simulator
for P, K, n in [(10, 5, 600), (1000, 20, 600), (4100, 8, 600), (10000, 20, 20000)]:
    out = []
    for draw, variant in (('bf16', 'lean'), ('bf16', 'pipe'), ('f16', 'pipe')):
        os.environ['RECOGYM_DRAW'] = draw
        os.environ['RECOGYM_BF16'] = variant
        cfg = Configuration({**env_1_args, 'random_seed': 100 + P, 'num_products': P, 'K': K})
        sim = Simulator(cfg, n, device='cuda:0')
        sim.reset_users(0, n)
        sim.run()
        c = sim.counters()
        out.append(f'{draw}/{variant}: exact {c["exact_draws"]}/{c["organic"]} = {c["exact_draws"]/c["organic"]:.4f} clicks {c["clicks"]}')
        sim.close()
    print(P, K, ' | '.join(out), flush=True)
