"""rg_sim_sort_log on the whole log of one C3 run (under rocprofv3 --kernel-trace --stats: which of its kernels the time is).
usage: python tools/sort_probe.py [users]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from recogym_amd.sim import Simulator, default_log_capacity  # noqa: E402

users = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfg = bench.make_config('c3')
sim = Simulator(cfg, users, device='cuda:0', log_capacity=default_log_capacity(cfg, users), **bench.arms_of('c3', cfg)[0][1])
sim.reset_users(0, users)
sim.run()
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    out, off = sim.sorted_log()
    torch.cuda.synchronize()
    print(f'sort {i}: {1e3 * (time.perf_counter() - t0):.2f} ms, {out.shape[0]} rows', flush=True)
    del out, off
