"""generate_beta's pairing at a size the reference's own algorithm cannot reach (P x P float64 = 80 GB at P = 10^5): time
recogym_amd.envs.static_params.flip_index_blocked on the tables the reference's draw order gives, check the result's shape
(an involution with exactly 2 x number_of_flips moved products, every swapped pair positively correlated).
    python tools/flip_large.py [P] [K] [flips]      -> one JSON line (profiles/r4/flip_index_blocked_p100000.json)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpy.random.mtrand import RandomState
from recogym_amd.envs.static_params import flip_index_blocked

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
F = int(sys.argv[3]) if len(sys.argv) > 3 else 100
G = RandomState(42).normal(size=(P, K))
t0 = time.perf_counter()
idx = flip_index_blocked(G, F)
dt = time.perf_counter() - t0
moved = np.nonzero(idx != np.arange(P))[0]
assert moved.size == 2 * F and np.array_equal(idx[idx], np.arange(P))
corr = np.einsum('ij,ij->i', G[moved], G[idx[moved]])
print(json.dumps(dict(P=P, K=K, flips=F, seconds=round(dt, 1), cpu_count=os.cpu_count(), moved_products=int(moved.size),
                      least_correlated_swapped_pair=float(corr.min()), most_correlated_swapped_pair=float(corr.max()),
                      full_matrix_bytes=8 * P * P, note='row blocks of Gamma Gamma^T (256 MB each), two passes per round')))
