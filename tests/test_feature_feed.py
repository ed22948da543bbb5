"""the vectorised training feed reproduces AbstractFeatureProvider.train_data
(reference agents/abstract.py:190-279) — against training sets the unmodified reference built from
two fixture logs (tests/make_golden.py train_feed), and against a row-by-row restatement on edge
cases."""
import json
import os

import numpy as np
import pytest
from scipy import sparse

import golden_util as gu
from recogym_amd.agents.feature_feed import train_data_from_log

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def log_columns_of(cols):
    is_b = cols['z'] == 1
    return dict(u=cols['u'].astype(np.int32), is_bandit=is_b,
                v=np.where(is_b, 0, cols['v']).astype(np.int32),
                a=np.where(is_b, cols['a'], 0).astype(np.int32),
                c=np.where(is_b, cols['c'], np.nan).astype(np.float32), ps=cols['ps'])


@pytest.mark.parametrize('fixture', ['philox_ouc', 'mt_random_agent'])
def test_matches_the_reference_training_set(fixture):
    meta, cols = gu.load(fixture)
    P = meta['env_args']['num_products']
    g = np.load(os.path.join(GOLDEN, 'train_feed_' + fixture + '.npz'))
    want = sparse.csr_matrix((g['data'], g['indices'], g['indptr']), shape=tuple(g['shape']))
    feats, actions, deltas, pss = train_data_from_log(log_columns_of(cols), P)
    assert feats.shape == want.shape and feats.dtype == np.dtype(str(g['data_dtype'])) == np.int16
    assert np.array_equal(feats.indptr, want.indptr)
    assert np.array_equal(feats.indices, want.indices) and np.array_equal(feats.data, want.data)
    assert actions.dtype == g['actions'].dtype and np.array_equal(actions, g['actions'])
    assert np.array_equal(deltas, g['deltas'])
    np.testing.assert_array_equal(pss, g['pss'])


def slow_restatement(u, is_b, v, a, c, ps, P):
    feats, acts, dl, pp = [], [], [], []
    cur, counts = None, None
    for i in range(len(u)):
        if u[i] != cur:
            cur, counts = u[i], np.zeros(P, dtype=np.int64)
        if not is_b[i]:
            counts[v[i]] += 1
        else:
            feats.append(counts.copy()); acts.append(a[i]); dl.append(c[i]); pp.append(ps[i])
    return np.array(feats).reshape(-1, P), np.array(acts), np.array(dl), np.array(pp)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_edge_cases_against_the_row_by_row_loop(seed):
    """Users with no organic row before a bandit row, with no bandit row at all, single-row users,
    repeated views of one product, the DataFrame input form."""
    import pandas as pd
    rng = np.random.RandomState(seed)
    P = 7
    u, z, v, a, c, ps = [], [], [], [], [], []
    for user in range(60):
        n = rng.randint(1, 12)
        kinds = rng.rand(n) < (0.0 if user % 7 == 0 else 1.0 if user % 11 == 0 else 0.5)
        for t in range(n):
            u.append(user * 3 + 5); z.append(int(kinds[t]))
            v.append(0 if kinds[t] else rng.randint(0, 3 if user % 2 else P))
            a.append(rng.randint(0, P) if kinds[t] else 0)
            c.append(float(rng.rand() < 0.2) if kinds[t] else np.nan)
            ps.append(rng.rand() if kinds[t] else np.nan)
    u, z, v, a, c, ps = map(np.array, (u, z, v, a, c, ps))
    is_b = z == 1
    want = slow_restatement(u, is_b, v, a, c, ps, P)
    cols = dict(u=u.astype(np.int32), is_bandit=is_b, v=v.astype(np.int32), a=a.astype(np.int32),
                c=c.astype(np.float32), ps=ps)
    df = pd.DataFrame({'t': np.zeros(len(u), np.float32), 'u': pd.array(u, dtype=pd.UInt16Dtype()),
                       'z': np.where(is_b, 'bandit', 'organic'),
                       'v': pd.array(np.where(is_b, 0, v), dtype=pd.UInt16Dtype()),
                       'a': pd.array(np.where(is_b, a, 0), dtype=pd.UInt16Dtype()), 'c': c, 'ps': ps})
    df.loc[is_b, 'v'] = pd.NA
    df.loc[~is_b, 'a'] = pd.NA
    for log in (cols, df):
        feats, actions, deltas, pss = train_data_from_log(log, P)
        assert np.array_equal(feats.toarray(), want[0])
        assert np.array_equal(actions, want[1]) and np.array_equal(deltas, np.nan_to_num(want[2]))
        np.testing.assert_array_equal(pss, want[3])
        dense = train_data_from_log(log, P, is_sparse=False)[0]
        assert dense.dtype == np.float64 and np.array_equal(dense, want[0])


def test_trainable_logreg_agent_row_by_row_equals_whole_log():
    """recogym_amd.agents.LogregMulticlassIpsAgent: feeding a log through train() observation by
    observation (the reference's offline protocol) and handing the same log to train_from_log()
    give the same training set, hence the same sklearn fit, bit for bit."""
    from recogym_amd.agents import LogregMulticlassIpsAgent, logreg_multiclass_ips_args
    from recogym_amd.envs.configuration import Configuration
    from recogym_amd.envs.context import DefaultContext
    from recogym_amd.envs.observation import Observation
    from recogym_amd.envs.session import OrganicSessions
    meta, cols = gu.load('philox_ouc')
    keep = cols['u'] < 60
    cols = {k: v[keep] for k, v in cols.items()}
    P = meta['env_args']['num_products']
    cfg = Configuration({**logreg_multiclass_ips_args, 'num_products': P, 'random_seed': 3, 'max_iter': 300})
    slow, fast = LogregMulticlassIpsAgent(cfg), LogregMulticlassIpsAgent(cfg)
    sessions, cur = OrganicSessions(), None
    for i in range(len(cols['u'])):
        u, t = int(cols['u'][i]), int(cols['t'][i])
        if u != cur:
            cur, sessions = u, OrganicSessions()
        if cols['z'][i] == 0:
            sessions.next(DefaultContext(t, u), int(cols['v'][i]))
        else:
            action = {'t': t, 'u': u, 'a': int(cols['a'][i]), 'ps': float(cols['ps'][i]), 'ps-a': ()}
            slow.train(Observation(DefaultContext(t, u), sessions), action, int(cols['c'][i]), False)
            sessions = OrganicSessions()
    fast.train_from_log(log_columns_of(cols))
    a, b = slow.build(), fast.build()
    assert np.array_equal(a.coef_t, b.coef_t) and np.array_equal(a.intercept, b.intercept)
    assert np.array_equal(a.classes, b.classes) and a.coef_t.shape[1] == len(a.classes) > 2


def _bmf_training_case():
    import os
    meta, cols = gu.load('mt_random_agent')
    g = np.load(os.path.join(GOLDEN, 'bandit_mf_training_mt_random_agent.npz'))
    return meta, cols, g


def test_bandit_mf_training_from_a_log_equals_the_reference_training():
    """recogym_amd.agents.BanditMFSquareAgent.train_from_log on the fixture log, starting from the
    reference agent's initial embeddings, ends at the embeddings the UNMODIFIED reference reached
    through its own train() calls (490 RMSprop steps; float32 torch arithmetic on the CPU: compared to
    1e-6, exact on the machine that made the fixture), and so does the call-by-call train()."""
    from recogym_amd.agents import BanditMFSquareAgent, bandit_mf_square_args
    from recogym_amd.envs.configuration import Configuration
    from recogym_amd.envs.context import DefaultContext
    from recogym_amd.envs.observation import Observation
    from recogym_amd.envs.session import OrganicSessions
    meta, cols, g = _bmf_training_case()
    cfg = Configuration({**bandit_mf_square_args, 'num_products': meta['env_args']['num_products']})
    fast = BanditMFSquareAgent(cfg, g['init_product'], g['init_user'])
    fast.train_from_log(log_columns_of(cols))
    assert fast.curr_step == int(g['steps'])
    np.testing.assert_allclose(fast.product_embedding.weight.detach().numpy(), g['product'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(fast.user_embedding.weight.detach().numpy(), g['user'], rtol=1e-6, atol=1e-7)
    slow = BanditMFSquareAgent(cfg, g['init_product'], g['init_user'])
    sessions, cur = OrganicSessions(), None
    for i in range(len(cols['u'])):
        u, t = int(cols['u'][i]), int(cols['t'][i])
        if u != cur:
            cur, sessions = u, OrganicSessions()
        if cols['z'][i] == 0:
            sessions.next(DefaultContext(t, u), int(cols['v'][i]))
        else:
            slow.train(Observation(DefaultContext(t, u), sessions),
                       {'t': t, 'u': u, 'a': int(cols['a'][i]), 'ps': float(cols['ps'][i]), 'ps-a': ()},
                       int(cols['c'][i]), False)
            sessions = OrganicSessions()
    assert np.array_equal(slow.product_embedding.weight.detach().numpy(), fast.product_embedding.weight.detach().numpy())
    assert np.array_equal(slow.user_embedding.weight.detach().numpy(), fast.user_embedding.weight.detach().numpy())
    # and the frozen policy both produce is the same table
    assert np.array_equal(slow.frozen().table, fast.frozen().table)


def test_bandit_mf_training_with_organic_only_users():
    """The offline protocol calls train once (no action) for every organic-only warm-up user; those calls
    count towards the mini-batch clock.  Log-fed training reproduces that."""
    from recogym_amd.agents import BanditMFSquareAgent, bandit_mf_square_args
    from recogym_amd.envs.configuration import Configuration
    from recogym_amd.envs.context import DefaultContext
    from recogym_amd.envs.observation import Observation
    from recogym_amd.envs.session import OrganicSessions
    meta, cols = gu.load('mt_drift_organic_users')
    P, n_org = meta['env_args']['num_products'], meta['n_organic']
    cfg = Configuration({**bandit_mf_square_args, 'num_products': P})
    rng = np.random.RandomState(0)
    ip, iu = rng.randn(P, 5).astype(np.float32), rng.randn(P, 5).astype(np.float32)
    fast = BanditMFSquareAgent(cfg, ip, iu)
    fast.train_from_log(log_columns_of(cols), n_org)
    slow = BanditMFSquareAgent(cfg, ip, iu)
    for uid in np.unique(cols['u']):
        sessions = OrganicSessions()
        for i in np.flatnonzero(cols['u'] == uid):
            t = int(cols['t'][i])
            if cols['z'][i] == 0:
                sessions.next(DefaultContext(t, int(uid)), int(cols['v'][i]))
            else:
                slow.train(Observation(DefaultContext(t, int(uid)), sessions),
                           {'t': t, 'u': int(uid), 'a': int(cols['a'][i]), 'ps': float(cols['ps'][i]), 'ps-a': ()},
                           int(cols['c'][i]), False)
                sessions = OrganicSessions()
        if uid < n_org:
            slow.train(Observation(DefaultContext(0, int(uid)), sessions), None, None, True)
    assert slow.curr_step == fast.curr_step > 0
    assert np.array_equal(slow.product_embedding.weight.detach().numpy(), fast.product_embedding.weight.detach().numpy())
    assert np.array_equal(slow.user_embedding.weight.detach().numpy(), fast.user_embedding.weight.detach().numpy())
