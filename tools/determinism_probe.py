"""The same run N times on one simulator: every ordered log against the first one (rows are keyed by (seed, user, event), so any
difference is a bug, not noise).  usage: python tools/determinism_probe.py <workload> [runs] [p_click 0/1] [arm index] [users]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from recogym_amd.sim import Simulator, default_log_capacity  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'c4shard'
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pc = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
cfg = bench.make_config(wl)
users = int(sys.argv[5]) if len(sys.argv) > 5 else bench.WORKLOADS[wl][1]
arm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
name, kw = bench.arms_of(wl, cfg)[arm]
sim = Simulator(cfg, users, device='cuda:0', log_capacity=default_log_capacity(cfg, users), p_click=pc, **kw)
ref = None
out = dict(workload=wl, arm=name, users=users, p_click=pc, runs=runs, env={k: v for k, v in os.environ.items() if k.startswith('RECOGYM_')}, diffs=[])
for i in range(runs):
    sim.reset_users(0, users)
    sim.run()
    log, off = sim.sorted_log()
    torch.cuda.synchronize()
    if ref is None:
        ref = log.clone()
        out['rows'] = int(ref.shape[0])
        continue
    if log.shape != ref.shape:
        out['diffs'].append(dict(run=i, shape=list(log.shape)))
        continue
    ne = (log != ref).any(dim=1)
    n = int(ne.sum().item())
    first = []
    if n:
        idx = torch.nonzero(ne).flatten()[:5]
        for j in idx.tolist():
            a, b = ref[j].tolist(), log[j].tolist()
            first.append(dict(row=j, u=a[0] & 0xFFFFFFFF, t=a[1], ref_code=a[2] & 0xFFFFFFFF, got_code=b[2] & 0xFFFFFFFF))
    out['diffs'].append(dict(run=i, rows_differing=n, first=first))
    del log, off
print(json.dumps(out), flush=True)
