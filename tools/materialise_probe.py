"""generate_logs end to end (simulate -> sort -> decode on device -> DataFrame) for n users."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recogym_amd
from recogym_amd.envs.reco_env_v1 import env_1_args
from recogym_amd.agents import OrganicUserEventCounterAgent, organic_user_count_args
from recogym_amd.envs.configuration import Configuration
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
env = recogym_amd.make('reco-gym-v1')
env.init_gym({**env_1_args, 'random_seed': 42, 'num_products': 1000, 'K': 20, 'sigma_omega': 0.0})
agent = OrganicUserEventCounterAgent(Configuration({**organic_user_count_args, 'num_products': 1000, 'select_randomly': True}))
for rep in range(2):
    t0 = time.perf_counter()
    cnt, sim = env.simulate(n, agent, 0)
    import torch; torch.cuda.synchronize()
    t1 = time.perf_counter()
    cols = sim.log_columns()
    t2 = time.perf_counter()
    from recogym_amd.envs.reco_env_v1 import columns_to_dataframe
    df = columns_to_dataframe(cols, 1000)
    t3 = time.perf_counter()
    sim.close()
    rows = len(df)
    print(f'{n} users, {rows/1e6:.1f} M rows: simulate {t1-t0:.2f}s | sort+decode on device+copy {t2-t1:.2f}s ({rows/(t2-t1)/1e6:.0f} M rows/s) | '
          f'DataFrame {t3-t2:.2f}s ({rows/(t3-t2)/1e6:.1f} M rows/s) | end to end {rows/(t3-t0)/1e6:.1f} M rows/s', flush=True)
print(df.dtypes.to_dict())

# --- training feed on the device over the same kind of log ---
import torch
from recogym_amd.agents.feature_feed import train_data_from_log_torch
cnt, sim = env.simulate(n, agent, 0)
dev_cols = sim.log_columns_device()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = train_data_from_log_torch(dev_cols, 1000)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'training feed on device: {dev_cols["u"].numel()/1e6:.1f} M log rows -> {t["crow"].numel()-1} feature rows, '
          f'{t["col"].numel()/1e6:.0f} M CSR entries in {t1-t0:.2f} s = {dev_cols["u"].numel()/(t1-t0)/1e6:.1f} M rows/s', flush=True)
sim.close()
