"""k_walk2's iterations by kind (-DRG_WALK_TIMING build, loaded with RECOGYM_HIP_LIB): how many iterations of each kind a run of a
bench workload takes, how many lanes of the wave have an event of that kind in them, the wave cycles spent in them.
    RECOGYM_HIP_LIB=.../librecogym_hip_walktiming.so python tools/walk_kinds.py [workload] [users]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recogym_amd import _abi

workload = sys.argv[1] if len(sys.argv) > 1 else 'c3'
users = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS[workload][1]
lib = _abi.load()
assert hasattr(lib, 'rg_debug_walk_kinds'), 'needs the -DRG_WALK_TIMING build (RECOGYM_HIP_LIB)'
lib.rg_debug_walk_kinds.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.rg_debug_walk_kinds.restype = None
cfg, sim = bench.make_sim(workload, users, 'cuda:0', None)
out = (C.c_ulonglong * 20)()
for rep in range(2):
    sim.reset_users(0, users)
    lib.rg_debug_walk_kinds(out, 1)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); sim.run(); t1.record(); torch.cuda.synchronize()
lib.rg_debug_walk_kinds(out, 0)
c = sim.counters()
ev = c['organic'] + c['bandit']
names = ['memo', 'search', 'bandit', 'click']
cy = sum(out[4 * k + 2] for k in range(4))
res = dict(workload=workload, users=users, run_ms=round(t0.elapsed_time(t1), 2), events=ev,
           helpers=os.environ.get('RECOGYM_WALK_HELPERS'), memo_hits=c['memo_hits'], organic=c['organic'], exact_sweeps=c['exact_sweeps'], helper_events=out[16], helpers_dealt=out[17])
for k, n in enumerate(names):
    it, ln, cyk = out[4 * k], out[4 * k + 1], out[4 * k + 2]
    res[n] = dict(iterations=it, lanes_per_iteration=round(ln / max(it, 1), 1), cycles_per_iteration=round(cyk / max(it, 1)),
                  share_of_wave_cycles=round(cyk / max(cy, 1), 3))
print(json.dumps(res))
