"""Time of ONE lock-step step 0 of a c3drift-shaped run (every user organic: an unsliced sweep of all of them): k_draw_tp and
k_pick from the library's HIP-event profile.  Used with the -DRG_TP_ABL / -DRG_PICK_ABL timing builds (RECOGYM_HIP_LIB): their
results are wrong by design, only the times are read.  usage: python tools/tp_probe.py [users] [name]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

users = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
name = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get('RECOGYM_HIP_LIB', 'default'))
from recogym_amd.sim import Simulator, default_log_capacity  # noqa: E402
cfg = bench.make_config('c3drift')
sim = Simulator(cfg, users, device='cuda:0', log_capacity=default_log_capacity(cfg, users), **bench.arms_of('c3drift', cfg)[0][1])
sweep, pick = [], []
for i in range(4):
    sim.set_profiling(True)
    sim.reset_users(0, users)
    sim.step()
    torch.cuda.synchronize()
    p = sim.profile()
    sweep.append(round(p['draw_mfma_ms'], 3)); pick.append(round(p['draw_search_ms'], 3))
    sim.set_profiling(False)
print(json.dumps(dict(name=name, users=users, sweep_ms=sweep[1:], pick_ms=pick[1:])))
