"""BASELINE config 5's shape on one GPU: verify_agents(env, N, {BanditMFSquare lookup, frozen LogregMulticlassIps}) at
P = 10 000, K = 20 — the full evaluate_agent loop of recogym_amd (both frozen policies inside the device loop, counters
reduced, Beta quantiles of the CTR).  The models are random (training is the reference's host code, out of scope):
a random product -> action table with logit propensities, and random coefficients for a 10^4-class LogReg."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import recogym_amd as recogym
from recogym_amd import Configuration, env_1_args
from recogym_amd.agents import LastViewTableAgent, LogregFrozenAgent

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
env = recogym.make('reco-gym-v1')
env.init_gym({**env_1_args, 'random_seed': 42, 'num_products': P, 'K': 20})
rng = np.random.RandomState(0)
cfg = Configuration({'num_products': P})
agents = {
    'bandit-mf (frozen table)': LastViewTableAgent(cfg, rng.randint(0, P, size=P), ps=rng.rand(P).astype(np.float64)),
    'logreg-ips (frozen, %d classes)' % P: LogregFrozenAgent(cfg, rng.randn(P, P) * 0.1, rng.randn(P) * 0.1, np.arange(P, dtype=np.int32)),
}
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    df = recogym.verify_agents(env, n, agents)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'verify_agents: P={P}, {n} users per agent, 2 agents: {dt:.2f} s', flush=True)
print(df.to_string())
