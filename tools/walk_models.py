"""Two numpy models of C3's users (P = 10 000, K = 20; tables as the bench draws them) behind decisions of round 5:
  * `band`: the share of organic draws the correlated certificate cannot certify, as a function of the sweep terms' budget delta and
    of a separate budget delta_c for the recomputed in-chunk prefixes (cert_correlated, rg_common.hpp) — 0.39 % at 1e-5 / 1e-5
    (measured 0.36 %), 0.403 % at 1e-5 / 7.4e-5, 1.66 % at 1.1e-4 (measured 1.5 %);
  * `memo`: hit rates of a first-come memo of m certified products per user against a memo pre-filled with the m heaviest.
    python tools/walk_models.py band|memo"""
import sys
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from recogym_amd.envs.static_params import draw_tables

cfg = bench.make_config('c3')
G, mu, B, mb = draw_tables(cfg)
rng = np.random.default_rng(1)


def users(n):
    om = rng.normal(0, 1.0, size=(n, cfg.K))
    L = om @ G.T + mu
    L -= L.max(1, keepdims=True)
    return np.exp(L)


def band():
    E_ = users(400)
    n, P = E_.shape

    def uncert(delta, delta_c, rho):
        tot = 0.0
        for u in range(n):
            e = E_[u]; C = np.cumsum(e); S = C[-1]
            Cprev = np.concatenate([[0.0], C[:-1]])
            A = Cprev[(np.arange(P) // 32) * 32]
            a = Cprev - A; b = C - A
            T = S - A
            dp = delta * (1 + 2 * delta); dc = delta_c * (1 + 2 * delta_c)
            lo = (A * (1 + dp) + a * (1 + dc) + rho * S) / (T * (1 - dp) + A * (1 + dp))
            hi = (A * (1 - dp) + b * (1 - dc) - rho * S) / (T * (1 + dp) + A * (1 - dp))
            lo[0] = 0.0; hi[-1] = 1.0
            tot += 1.0 - np.maximum(hi - lo, 0).sum()
        return tot / n
    for d, dc in ((1.0e-5, 1.0e-5), (1.0e-5, 3e-5), (1.0e-5, 7.4e-5), (1.1e-4, 1.1e-4)):
        print(f'delta {d:g}, delta_c {dc:g}: uncertified share of the draws {uncert(d, dc, 1.2e-7):.5f}')


def memo():
    Pm = users(3000)
    Pm /= Pm.sum(1, keepdims=True)
    srt = -np.sort(-Pm, axis=1)
    print('mean mass of the heaviest product', srt[:, 0].mean(), '9 heaviest', srt[:, :9].sum(1).mean(), '18', srt[:, :18].sum(1).mean())
    res = {m: [0, 0] for m in (9, 18, 36, 1000)}
    top = {m: [0, 0] for m in (9, 18)}
    for u in range(Pm.shape[0]):
        nd = max(1, int(rng.exponential(26)))            # organic draws of a user (mean 26 on C3)
        d = rng.choice(cfg.num_products, size=nd, p=Pm[u])
        for m in res:
            seen = set(); hits = 0
            for x in d:
                if x in seen: hits += 1
                elif len(seen) < m: seen.add(x)
            res[m][0] += hits; res[m][1] += nd
        order = np.argsort(-Pm[u])
        for m in top:
            pre = set(order[:m].tolist())
            top[m][0] += sum(1 for x in d if x in pre); top[m][1] += nd
    for m in res: print('first-come memo of', m, 'products: hit rate', round(res[m][0] / res[m][1], 4))
    for m in top: print('memo pre-filled with the', m, 'heaviest: hit rate', round(top[m][0] / top[m][1], 4))


if __name__ == '__main__':
    {'band': band, 'memo': memo}[sys.argv[1] if len(sys.argv) > 1 else 'band']()
