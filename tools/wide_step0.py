"""The wide-K sweep in steady state: step 0 of a K = 64 run (every user organic, the GPU full for ~19 rounds of blocks),
HIP-event time of the sweep kernel alone; RECOGYM_ABLATE selects the timing experiments of the tile loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recogym_amd import Configuration, env_1_args
from recogym_amd.sim import Simulator

users = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': P, 'K': 64, 'sigma_omega': 0.1})
sim = Simulator(cfg, users, device='cuda:0', log_capacity=users * 2)
best = 1e9
for rep in range(3):
    sim.reset_users(0, users)
    sim.set_profiling(True)
    sim.step()
    prof = sim.profile()
    sim.set_profiling(False)
    best = min(best, prof['draw_mfma_ms'])
import ctypes as C
from recogym_amd import _abi
lib = _abi.load()
if hasattr(lib, 'rg_debug_f16w_timing'):
    out = (C.c_ulonglong * 8)()
    lib.rg_debug_f16w_timing(out)
    nt = max(out[6], 1)
    names = ['barrier wait', 'mu / ring preload', 'MFMA + exp slots', 'sums / book', 'loop top', 'dma issue + reference']
    print('  cycles per tile (wave 0 of every block):', {n: round(out[i] / nt, 1) for i, n in enumerate(names)}, 'sum', round(sum(out[i] for i in range(6)) / nt, 1))
tiles = (users + 255) // 256 * (P // 64)
print(f'ablate {os.environ.get("RECOGYM_ABLATE", "0")}: step-0 sweep {best:.2f} ms = {best * 1e3 * 256 / tiles:.3f} us per tile and block '
      f'(256 CUs busy)')
