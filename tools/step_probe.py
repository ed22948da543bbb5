"""Where a bench step's wall time goes: reset_users / run / counters timed separately (each followed by a device sync)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
_, users, _, pol = bench.WORKLOADS[wl]
from recogym_amd.sim import default_log_capacity
cfg, sim = bench.make_sim(wl, users, 'cuda:0', default_log_capacity(bench.make_config(wl), users))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sim.reset_users(0, users); torch.cuda.synchronize(); t1 = time.perf_counter()
    sim.run(); torch.cuda.synchronize(); t2 = time.perf_counter()
    c = sim.counters(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'rep {rep}: reset {1e3*(t1-t0):.2f} ms  run {1e3*(t2-t1):.2f} ms  counters {1e3*(t3-t2):.2f} ms  total {1e3*(t3-t0):.2f} ms')
sim.set_profiling(True); sim.reset_users(0, users); sim.run(); print(sim.profile())
