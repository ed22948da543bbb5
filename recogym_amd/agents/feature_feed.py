"""Vectorised offline-training feed.

The reference collects every log row through `ModelBuilder.train` (agents/abstract.py:55-83) and
turns them into the training set of its model-based agents with
`AbstractFeatureProvider.train_data` (agents/abstract.py:190-279): per user, a Python loop over
`DataFrame.iterrows()` that keeps the list of products viewed so far and emits, for every bandit
row (the phantom row included), one 1 x P sparse row of cumulative view counts, plus the action,
the click and the propensity.  That is O(users x rows) pandas work (~10^4 rows/s).

`train_data_from_log` builds the same `(features, actions, deltas, pss)` tuple from the log in one
pass of array operations: organic rows are grouped by (user, product); a group contributes one
CSR entry to every bandit row of its user that comes after its first view, and the entry's value
is the number of the group's views before that row (one searchsorted on composite keys).

With a `weight_history_function` the features are time-weighted float sums, not counts: that (rare) form takes the exact
per-row construction of agents/views_history.train_data_weighted instead of the vectorised count path.
"""
import numpy as np
from scipy import sparse


def _columns_of(log):
    """DataFrame of generate_logs or dict of Simulator.log_columns() -> plain arrays."""
    if isinstance(log, dict):
        is_b = np.asarray(log['is_bandit'], dtype=bool)
        return (np.asarray(log['u']).view(np.uint32).astype(np.int64), is_b,
                np.asarray(log['v'], dtype=np.int64), np.asarray(log['a'], dtype=np.int64),
                np.asarray(log['c']), np.asarray(log['ps'], dtype=np.float64))
    is_b = (log['z'] == 'bandit').to_numpy()
    u = log['u'].to_numpy(dtype=np.int64)
    v = log['v'].to_numpy(dtype=np.float64, na_value=np.nan)
    a = log['a'].to_numpy(dtype=np.float64, na_value=np.nan)
    return (u, is_b, np.where(is_b, 0, np.nan_to_num(v)).astype(np.int64),
            np.where(is_b, np.nan_to_num(a), 0).astype(np.int64),
            log['c'].to_numpy(dtype=np.float64, na_value=np.nan),
            log['ps'].to_numpy(dtype=np.float64, na_value=np.nan))


def _times_of(log):
    return np.asarray(log['t']) if isinstance(log, dict) else log['t'].to_numpy()


def train_data_from_log(log, num_products, is_sparse=True, weight_history_function=None):
    """-> (features, actions, deltas, pss) as AbstractFeatureProvider.train_data returns them:
    features  CSR (n_bandit_rows, P) of int16 view counts, sorted indices (dense float64 array
              when is_sparse is False), rows in log order (users ascending, t ascending);
    actions   int16; deltas int16 (the click column); pss float64.
    `log` must be in the reference's row order (every user's rows contiguous, t ascending)."""
    u, is_b, v, a, c, ps = _columns_of(log)
    if weight_history_function is not None:             # time-weighted views (agents/abstract.py:216-263): the exact per-row form
        from .views_history import train_data_weighted
        return train_data_weighted((_times_of(log), u, is_b, v, a, np.nan_to_num(c), ps), int(num_products),
                                   weight_history_function, is_sparse)
    n = len(u)
    P = int(num_products)
    pos = np.arange(n, dtype=np.int64)
    b_pos = pos[is_b]                                   # positions of the bandit rows = output rows
    nb = len(b_pos)
    actions = a[is_b].astype(np.int16)
    deltas = np.nan_to_num(c[is_b]).astype(np.int16)
    pss = ps[is_b].astype(np.float64)
    o_pos = pos[~is_b]
    if nb == 0 or len(o_pos) == 0:
        feats = sparse.csr_matrix((nb, P), dtype=np.int16)
        return (feats if is_sparse else np.asarray(feats.todense(), dtype=np.float64)), actions, deltas, pss
    # bandit rows before each log position, and the end of every user's block of bandit rows
    b_before = np.cumsum(is_b) - is_b                   # exclusive prefix count at every position
    last_of_user = np.r_[u[1:] != u[:-1], True]
    # index of the last row of the user each row belongs to
    ends = pos[last_of_user]
    user_of_row = np.cumsum(np.r_[0, last_of_user[:-1]])          # dense user ordinal per row
    row_end = ends[user_of_row]                                   # last position of the row's user
    b_after_end = np.cumsum(is_b)                                 # inclusive count: bandit rows up to position
    # organic rows grouped by (user ordinal, product), ordered by position inside a group
    ou, ov = user_of_row[o_pos], v[o_pos]
    key = ou * P + ov
    order = np.argsort(key, kind='stable')                         # stable: positions stay ascending
    key_s, pos_s = key[order], o_pos[order]
    new_grp = np.r_[True, key_s[1:] != key_s[:-1]]
    g_start = np.flatnonzero(new_grp)                              # first view of every group (in sorted order)
    g_of_view = np.cumsum(new_grp) - 1
    first_pos = pos_s[g_start]
    k_lo = b_before[first_pos]                                     # first bandit row after the group's first view
    k_hi = b_after_end[row_end[first_pos]]                         # one past the user's last bandit row
    m = (k_hi - k_lo).astype(np.int64)                             # entries the group contributes
    total = int(m.sum())
    grp = np.repeat(np.arange(len(g_start)), m)
    within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(m) - m, m)
    rows = k_lo[grp] + within                                      # output row of every entry
    cols = (key_s[g_start] % P)[grp]
    # value = views of the group before that bandit row: search (group, position) composite keys
    big = np.int64(n + 1)
    view_keys = g_of_view * big + pos_s
    q = grp * big + b_pos[rows]
    vals = np.searchsorted(view_keys, q, side='left') - g_start[grp]
    feats = sparse.csr_matrix((vals.astype(np.int16), (rows, cols)), shape=(nb, P), dtype=np.int16)
    feats.sort_indices()
    if not is_sparse:
        return np.asarray(feats.todense(), dtype=np.float64), actions, deltas, pss
    return feats, actions, deltas, pss


def train_data_from_log_torch(cols, num_products):
    """The same training set built with torch ops on whatever device the log columns live on
    (`Simulator.log_columns_device()` keeps them on the GPU): -> dict(crow, col, val (int16), actions
    (int16), deltas (int16), pss (float64)) of torch tensors; `crow/col/val` are the CSR arrays of
    the (n_bandit_rows, P) feature matrix with sorted column indices.  Same algorithm as
    train_data_from_log (group organic rows by (user, product); one entry per later bandit row of the
    user; count by searchsorted on composite keys); the sort by (row, column) that scipy does inside
    its COO->CSR conversion is one torch.sort here."""
    import torch
    u = cols['u'].to(torch.int64) & 0xFFFFFFFF
    is_b = cols['is_bandit'].to(torch.bool)
    v = cols['v'].to(torch.int64)
    dev = u.device
    n = u.numel()
    P = int(num_products)
    pos = torch.arange(n, device=dev, dtype=torch.int64)
    b_pos = pos[is_b]
    nb = b_pos.numel()
    out = dict(actions=cols['a'][is_b].to(torch.int16),
               deltas=torch.nan_to_num(cols['c'][is_b].to(torch.float32)).to(torch.int16),
               pss=cols['ps'][is_b].to(torch.float64))
    o_pos = pos[~is_b]
    if nb == 0 or o_pos.numel() == 0:
        out.update(crow=torch.zeros(nb + 1, dtype=torch.int64, device=dev),
                   col=torch.zeros(0, dtype=torch.int64, device=dev),
                   val=torch.zeros(0, dtype=torch.int16, device=dev), shape=(nb, P))
        return out
    ib = is_b.to(torch.int64)
    b_incl = torch.cumsum(ib, 0)
    b_before = b_incl - ib
    last_of_user = torch.ones(n, dtype=torch.bool, device=dev)
    last_of_user[:-1] = u[1:] != u[:-1]
    ends = pos[last_of_user]
    user_of_row = torch.cumsum(last_of_user.to(torch.int64), 0) - last_of_user.to(torch.int64)
    row_end = ends[user_of_row]
    key = user_of_row[o_pos] * P + v[o_pos]
    key_s, order = torch.sort(key, stable=True)
    pos_s = o_pos[order]
    new_grp = torch.ones_like(key_s, dtype=torch.bool)
    new_grp[1:] = key_s[1:] != key_s[:-1]
    g_start = torch.nonzero(new_grp).flatten()
    g_of_view = torch.cumsum(new_grp.to(torch.int64), 0) - 1
    first_pos = pos_s[g_start]
    k_lo = b_before[first_pos]
    k_hi = b_incl[row_end[first_pos]]
    m = k_hi - k_lo
    total = int(m.sum().item())
    grp = torch.repeat_interleave(torch.arange(g_start.numel(), device=dev), m)
    within = torch.arange(total, device=dev, dtype=torch.int64) - torch.repeat_interleave(torch.cumsum(m, 0) - m, m)
    rows = k_lo[grp] + within
    ccol = (key_s[g_start] % P)[grp]
    big = n + 1
    view_keys = g_of_view * big + pos_s
    q = grp * big + b_pos[rows]
    vals = torch.searchsorted(view_keys, q, right=False) - g_start[grp]
    rc, perm = torch.sort(rows * P + ccol)
    crow = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    crow[1:] = torch.cumsum(torch.bincount(rows, minlength=nb), 0)
    out.update(crow=crow, col=rc % P, val=vals[perm].to(torch.int16), shape=(nb, P))
    return out
