"""Wide-K sweep probes: (1) the same K = 64 / drift workload at two table sizes (is it bound by streaming its split
table?); (2) with a -DRG_F16W_TIMING build (RECOGYM_HIP_LIB), where a tile's cycles go."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recogym_amd import Configuration, env_1_args, _abi
from recogym_amd.sim import Simulator, default_log_capacity

sizes = ((12800, 1_250_000), (100_000, 1_250_000)) if len(sys.argv) < 2 else ((int(sys.argv[1]), int(sys.argv[2])),)
for P, users in sizes:
    cfg = Configuration({**env_1_args, 'random_seed': 42, 'num_products': P, 'K': 64, 'sigma_omega': 0.1})
    sim = Simulator(cfg, users, device='cuda:0', log_capacity=default_log_capacity(cfg, users))
    for rep in range(2):
        sim.set_profiling(rep == 1)
        sim.reset_users(0, users); sim.run()
    prof = sim.profile(); c = sim.counters()
    draws = c['organic']
    print(f'P={P}: sweep {prof["draw_mfma_ms"]:.1f} ms for {draws/1e6:.1f} M draws -> {prof["draw_mfma_ms"]*1e6/(draws*P/64):.3f} ns per (draw x 64-product tile); '
          f'exact {prof["draw_exact_ms"]:.1f} ms, advance {prof["advance_ms"]:.1f} ms, exact draws {c["exact_draws"]}')
    lib = _abi.load()
    if hasattr(lib, 'rg_debug_f16w_timing'):
        out = (C.c_ulonglong * 8)()
        lib.rg_debug_f16w_timing(out)
        tiles = max(out[6], 1)
        names = ['barrier wait', 'mu / ring preload', 'MFMA + exp slots', 'sums / book', 'loop top', 'dma issue + reference']
        print('  cycles per tile (wave 0 of every block):', {n: round(out[i] / tiles, 1) for i, n in enumerate(names)}, 'tiles', tiles)
    sim.close(); del sim; torch.cuda.empty_cache()
