"""Time-weighted view features — the `weight_history_function` option of the reference's ViewsFeaturesProvider
(recogym/agents/abstract.py:318-409) and of AbstractFeatureProvider.train_data (abstract.py:190-279).

With `config.weight_history_function = w` a user's feature vector at time T is not the count of views per product but
`sum over the user's earlier views of w(T - t_view)` per product.  The reference builds it from scipy sparse pieces — a
(views x P) int16 one-hot matrix, `.multiply(weights[:, None])`, `.sum(axis=0, dtype=float32)` — with the product ids and the
times cast to np.int16 on the way (wrapping beyond 32 767 like the reference).  The features feed a float division and a
cumulative sum whose last bits decide sampled actions, so this module makes the SAME scipy / numpy calls on the same dtypes
instead of re-deriving the arithmetic: bit-identical by construction (pinned on fixtures of the unmodified reference).

Per-user host path only: a time-weighted feature vector is a float per (user, product, event), not a count the device history
can hold, so agents with a weight function report no device policy and `generate_logs` walks them one user at a time.
"""
import numpy as np
from scipy import sparse


def weighted_views(products, times, now, weight_fn, num_products):
    """1 x P float32 np.matrix: the reference's `weighted_views.sum(axis=0, dtype=np.float32)` for the views (products, times)
    seen at time `now` (all three in the reference's int16)."""
    n = len(products)
    ixs = [np.int16(i) for i in range(n)]
    views = sparse.coo_matrix((np.ones(n, dtype=np.int16), (ixs, list(products))), shape=(n, num_products), dtype=np.int16)
    weights = weight_fn(np.int16(now) - np.array(list(times)))
    return views.multiply(weights[:, np.newaxis]).sum(axis=0, dtype=np.float32)


class ViewsHistory:
    """The with-history state of a ViewsFeaturesProvider: observe() the organic sessions, features(now) -> what the
    reference's `features(observation)` returns (a coo_matrix when is_sparse, else a dense (1, P) array)."""

    def __init__(self, num_products, weight_fn, is_sparse=False):
        self.num_products = int(num_products)
        self.weight_fn = weight_fn
        self.is_sparse = is_sparse
        self.reset()

    def reset(self):
        self.products = []
        self.times = []

    def observe(self, observation):
        for session in observation.sessions():
            self.products.append(np.int16(session['v']))
            self.times.append(np.int16(session['t']))

    def features(self, now):
        views = sparse.coo_matrix(weighted_views(self.products, self.times, now, self.weight_fn, self.num_products), copy=False)
        return views if self.is_sparse else np.array(views.todense())


def train_data_weighted(log_columns, num_products, weight_fn, is_sparse=True):
    """AbstractFeatureProvider.train_data with a weight function: one weighted feature row per bandit row, built from the views
    of the row's user before it (abstract.py:216-263).  `log_columns` = (t, u, is_bandit, v, a, c, ps) plain arrays in the
    reference's row order.  O(bandit rows x views of the user): the exact form, for the sizes a weighted model is trained on."""
    t, u, is_b, v, a, c, ps = log_columns
    feats, actions, deltas, pss = [], [], [], []
    cur = None
    products, times = [], []
    for i in range(len(u)):
        if u[i] != cur:
            cur, products, times = u[i], [], []
        if not is_b[i]:
            products.append(np.int16(v[i]))
            times.append(np.int16(t[i]))
            continue
        feats.append(sparse.coo_matrix(weighted_views(products, times, np.int16(t[i]), weight_fn, num_products), copy=False))
        actions.append(np.int16(a[i]))
        deltas.append(np.int16(c[i]))
        pss.append(ps[i])
    out = sparse.vstack(feats, format='csr')
    return ((out if is_sparse else np.array(out.todense(), dtype=float)), np.array(actions, dtype=np.int16), np.array(deltas),
            np.array(pss))
