"""Time of the sigma_omega = 0 sweep alone (k_sweep_xh or, RECOGYM_XH=0, k_draw_bf16p) on a C3-shaped run: HIP events of the
library's profile.  Used with the -DRG_XH_ABL timing builds (RECOGYM_HIP_LIB): their results are wrong by design, only the
sweep's time is read.  usage: python tools/xh_probe.py [users] [name]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

users = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
name = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get('RECOGYM_HIP_LIB', 'default'))
from recogym_amd.sim import Simulator, default_log_capacity  # noqa: E402
cfg = bench.make_config('c3')
sim = Simulator(cfg, users, device='cuda:0', log_capacity=default_log_capacity(cfg, users), **bench.arms_of('c3', cfg)[0][1])
out = []
for i in range(3):
    sim.set_profiling(True)
    sim.reset_users(0, users)
    sim.run()
    torch.cuda.synchronize()
    p = sim.profile()
    out.append(round(p['draw_mfma_ms'], 3))
c = sim.counters()
print(json.dumps(dict(name=name, users=users, sweep_ms=out, exact_sweeps=c['exact_sweeps'], organic=c['organic'])))
