"""Debug probe for the K = 65 class (fp32 MFMA draw + tile float64 kernel + k_tail): rows vs the oracle, the
mismatching events printed with their step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recogym_amd import Configuration, env_1_args
from recogym_amd.sim import Simulator
from oracle import oracle as orc

for K in [int(a) for a in sys.argv[1:]] or [65]:
  cfg = Configuration({**env_1_args, 'random_seed': 500 + K, 'num_products': 333, 'K': K, 'sigma_omega': 0.07})
  want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(250)
  for rep in range(2):
      sim = Simulator(cfg, 250, device='cuda:0', p_click=True)
      sim.reset_users(0, 250)
      sim.run()
      rows = sim.rows()
      cnt = sim.counters()
      sim.close()
      bad = np.nonzero(rows['v'] != want['v'])[0]
      print('K', K, 'rep', rep, 'rows', len(rows), len(want), 'bad', bad.size, {k: cnt[k] for k in ('exact_draws', 'organic', 'steps') if k in cnt})
      for i in bad[:40]:
          print('   row', i, 'u', rows['u'][i], 't', rows['t'][i], 'z', rows['z'][i], 'got', rows['v'][i], 'want', want['v'][i])
