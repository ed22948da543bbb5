"""Events/s of `env.generate_logs(n, agent)` with an ARBITRARY Python agent (not Random / OUC / a frozen model), at the reference's
small shapes, on both host-driven paths:
  batched   (the default since round 5: _generate_logs_batched) B users per rg_sim_step launch, agent.act on the host per user;
  per_user  (_generate_logs_per_user: reset / step / step_offline, one user at a time, one launch + one pinned read-back per event).
The NumPy reference's own figures for the same shapes are in profiles/r4/numpy_reference_cpu.json (c1: P = 10, c2: P = 1 000).
    python tools/per_user_path.py [--users 4096] [--per-user-users 40]      -> one JSON line per shape and path"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import recogym_amd as recogym
from recogym_amd.agents.abstract import Agent


class PythonEpsilonAgent(Agent):
    """Not a device policy: most-viewed product with probability 0.8, else uniform (its own numpy stream)."""

    def __init__(self, config):
        super().__init__(config)
        self.rng = np.random.RandomState(7)
        self.reset()

    def reset(self):
        self.views = np.zeros(self.config.num_products, dtype=np.int64)

    def act(self, observation, reward, done):
        for s in observation.sessions():
            self.views[int(s['v'])] += 1
        P = self.config.num_products
        a = int(self.views.argmax()) if self.rng.rand() < 0.8 else int(self.rng.randint(P))
        return {**super().act(observation, reward, done), 'a': a, 'ps': 1.0, 'ps-a': ()}


users = int(sys.argv[sys.argv.index('--users') + 1]) if '--users' in sys.argv else 4096
pu_users = int(sys.argv[sys.argv.index('--per-user-users') + 1]) if '--per-user-users' in sys.argv else 40
ref = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r4', 'numpy_reference_cpu.json')))
for name, case, over in (('P=10 K=5', 'c1', dict(num_products=10, K=5, sigma_omega=0.0)),
                         ('P=1000 K=20', 'c2', dict(num_products=1000, K=20, sigma_omega=0.0)),
                         ('P=10000 K=20', 'c3', dict(num_products=10000, K=20, sigma_omega=0.0))):
    for path in ('batched', 'per_user'):
        env = recogym.make('reco-gym-v1')
        env.init_gym({**recogym.env_1_args, 'random_seed': 42, **over})
        agent = PythonEpsilonAgent(recogym.Configuration({'num_products': over['num_products'], 'random_seed': 7}))
        gen = env._generate_logs_batched if path == 'batched' else env._generate_logs_per_user
        n = users if path == 'batched' else pu_users
        gen(3, agent, 0)                                # warm-up (library load, first launches)
        t0 = time.perf_counter()
        df = gen(n, agent, 0)
        dt = time.perf_counter() - t0
        events = len(df) - n
        r = ref['cases'][case]
        print(json.dumps(dict(shape=name, path=path, users=n, events=int(events), seconds=round(dt, 2), events_per_s=round(events / dt, 1),
                              us_per_event=round(1e6 * dt / events, 2),
                              numpy_reference_one_core_events_per_s=round(r['one_core_events_per_s'], 1),
                              numpy_reference_agent=r['agent'], numpy_reference_where='build container, profiles/r4/numpy_reference_cpu.json',
                              ratio_to_numpy_one_core=round(events / dt / r['one_core_events_per_s'], 3))), flush=True)
