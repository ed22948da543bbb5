"""The reference's class surface over the HIP path (needs a GPU): gym.make / init_gym /
reset / step / step_offline / generate_logs / test_agent / verify_agents."""
import json
import os
from copy import deepcopy

import numpy as np
import pandas as pd
import pytest
from scipy.stats.distributions import beta

import golden_util as gu
import recogym_amd as recogym
from recogym_amd import _abi
from recogym_amd.agents import Agent, OrganicUserEventCounterAgent, RandomAgent
from recogym_amd.envs.configuration import Configuration

pytestmark = pytest.mark.gpu


def make_env(over):
    env = recogym.make('reco-gym-v1')
    env.init_gym({**recogym.env_1_args, **over})
    return env


def make_agent(meta):
    aa = {'num_products': meta['env_args']['num_products'], 'with_ps_all': False,
          **meta['agent_args']}
    if meta['agent'] == 'random':
        return RandomAgent(Configuration(aa))
    if meta['agent'] == 'ouc':
        return OrganicUserEventCounterAgent(Configuration(
            {**gu.OUC_DEFAULTS, 'weight_history_function': None, **aa}))
    return None


def test_frozen_bandit_mf_agent_device_and_host_forms():
    """LastViewTableAgent built from the reference BanditMFSquare's (untrained) embeddings: the
    device policy reproduces the reference's log, and the per-user path with the Python act
    logs the same rows."""
    from recogym_amd.agents import LastViewTableAgent
    meta, want = gu.load('philox_bandit_mf')
    cfg = Configuration({'num_products': meta['env_args']['num_products'], 'with_ps_all': False})
    agent = LastViewTableAgent.from_bandit_mf(cfg, want['bmf_product_embedding'],
                                              want['bmf_user_embedding'])
    env = make_env(meta['env_args'])
    df = env.generate_logs(meta['n_users'], agent)
    assert_frames_match(frame_to_cols(df), want, 1e-5)
    df_seq = make_env(meta['env_args'])._generate_logs_per_user(20, agent, 0)
    keep = want['u'] < 20
    assert_frames_match(frame_to_cols(df_seq), {k: v[keep] for k, v in want.items()
                                                if k in ('t', 'u', 'z', 'v', 'a', 'c', 'ps')}, 1e-5)


def test_frozen_logreg_agent_device_and_host_forms():
    """LogregFrozenAgent carrying the model the reference's LogregMulticlassIpsAgent fitted: the
    device policy reproduces the reference's log row for row (every action is an argmax over 30
    class scores: the accumulation order is scipy's), and so does the per-user path with the
    Python act."""
    from recogym_amd.agents import LogregFrozenAgent
    meta, want = gu.load('philox_logreg')
    cfg = Configuration({'num_products': meta['env_args']['num_products'], 'with_ps_all': False})
    agent = LogregFrozenAgent(cfg, want['logreg_coef'], want['logreg_intercept'], want['logreg_classes'])
    env = make_env(meta['env_args'])
    df = env.generate_logs(meta['n_users'], agent)
    assert_frames_match(frame_to_cols(df), want, 1e-12)
    df_seq = make_env(meta['env_args'])._generate_logs_per_user(15, agent, 0)
    keep = want['u'] < 15
    assert_frames_match(frame_to_cols(df_seq), {k: v[keep] for k, v in want.items()
                                                if k in ('t', 'u', 'z', 'v', 'a', 'c', 'ps')}, 1e-12)


def test_log_to_training_feed_to_sklearn_to_frozen_device_policy():
    """The §8f chain end to end: a uniform-policy log from the device -> train_data_from_log ->
    sklearn's LogisticRegression fitted like the reference's build() (weights = deltas / pss,
    logreg_ips.py:89-99) -> LogregFrozenAgent in the device loop.  The new log equals the oracle's
    with the same frozen model, and every logged action IS sklearn's predict() on the feature row
    the feed rebuilds for it."""
    import warnings
    from sklearn.linear_model import LogisticRegression
    from oracle import oracle as orc
    from recogym_amd.agents import LogregFrozenAgent, train_data_from_log
    over = {'random_seed': 3, 'num_products': 25, 'K': 6}
    env = make_env(over)
    log = env.generate_logs(800)
    feats, actions, deltas, pss = train_data_from_log(log, 25)
    assert deltas.sum() > 5
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        lr = LogisticRegression(solver='lbfgs', max_iter=5000, random_state=1).fit(feats, actions, deltas / pss)
    cfg = Configuration({'num_products': 25, 'with_ps_all': False})
    agent = LogregFrozenAgent.from_sklearn(cfg, lr)
    df = make_env(over).generate_logs(300, agent)
    pol = agent.device_policy()
    want = orc.OracleEnv(Configuration({**recogym.env_1_args, **over}), rng_mode=orc.RNG_PHILOX,
                         policy=pol['policy'], policy_seed=0, logreg=pol['logreg']).generate_logs(300)
    assert_frames_match(frame_to_cols(df), {k: want[k] for k in ('t', 'u', 'z', 'v', 'a', 'c', 'ps')}, 1e-12)
    f2, a2, _, _ = train_data_from_log(df, 25)
    assert np.array_equal(lr.predict(f2), a2)


def frame_to_cols(df):
    return {
        't': df['t'].values.astype(np.int64),
        'u': df['u'].astype('Int64').fillna(-1).values.astype(np.int64),
        'z': (df['z'].values == 'bandit').astype(np.int64),
        'v': df['v'].astype('Int64').fillna(-1).values.astype(np.int64),
        'a': df['a'].astype('Int64').fillna(-1).values.astype(np.int64),
        'c': np.where(np.isnan(df['c'].values), -1, df['c'].values).astype(np.int64),
        'ps': df['ps'].values.astype(np.float64),
    }


def assert_frames_match(cols, want, ps_rtol):
    for k in ('t', 'u', 'z', 'v', 'a', 'c'):
        assert np.array_equal(cols[k], want[k].astype(np.int64)), k
    np.testing.assert_allclose(cols['ps'], want['ps'], rtol=ps_rtol, equal_nan=True)


@pytest.mark.parametrize('name', ['philox_p10', 'philox_p1000_k20', 'philox_random_agent',
                                  'philox_ouc', 'philox_ouc_eps', 'philox_flips_normbeta'])
def test_generate_logs_dataframe_equals_the_reference_log(name):
    meta, want = gu.load(name)
    env = make_env(meta['env_args'])
    df = env.generate_logs(meta['n_users'], make_agent(meta), meta['n_organic'])
    assert list(df.columns) == ['t', 'u', 'z', 'v', 'a', 'c', 'ps', 'ps-a']
    assert [str(df[c].dtype) for c in df.columns] == \
        ['float32', 'UInt16', 'object', 'UInt16', 'UInt16', 'float32', 'float64', 'object']
    assert_frames_match(frame_to_cols(df), want, 1e-6)
    # repeatable, and deepcopy-safe like the reference's harness expects
    df2 = deepcopy(env).generate_logs(meta['n_users'], make_agent(meta), meta['n_organic'])
    pd.testing.assert_frame_equal(df, df2)


@pytest.mark.parametrize('name', ['philox_random_agent', 'philox_ouc_eps', 'philox_p10'])
def test_per_user_gym_path_equals_batched_path(name):
    """reset/step/step_offline with the agent's Python act (batch of one user on the device)
    logs exactly what the all-users-at-once device policy logs."""
    meta, want = gu.load(name)
    n = 25
    env = make_env(meta['env_args'])
    agent = make_agent(meta)
    df_seq = env._generate_logs_per_user(n, agent, 0)
    df_dev = make_env(meta['env_args']).generate_logs(n, make_agent(meta), 0)
    a, b = frame_to_cols(df_seq), frame_to_cols(df_dev)
    assert_frames_match(a, b, 1e-6)
    keep = want['u'] < n + meta['n_organic']
    if meta['n_organic'] == 0:
        assert_frames_match(b, {k: v[keep] for k, v in want.items() if k != 'p_click'}, 1e-6)


class FixedCycleAgent(Agent):
    """A Python-only agent (no device form): cycles through the products; tests the generic path."""

    def __init__(self, config):
        super().__init__(config)
        self.i = 0

    def reset(self):
        self.i = 0

    def act(self, observation, reward, done):
        self.i += 1
        return {**super().act(observation, reward, done),
                'a': self.i % self.config.num_products, 'ps': 1.0, 'ps-a': ()}


def test_arbitrary_python_agent_matches_oracle_step_api():
    from oracle import oracle as orc
    over = dict(random_seed=77, num_products=12, K=4)
    env = make_env(over)
    agent = FixedCycleAgent(Configuration({'num_products': 12}))
    df = env.generate_logs(30, agent)
    o = orc.OracleEnv(Configuration({**recogym.env_1_args, **over}), rng_mode=orc.RNG_PHILOX)
    rows = []
    for user in range(30):
        o.reset(user)
        i = 0
        org, reward, done = o.step(None)
        rows += [(int(r['t']), user, 0, int(r['v']), -1, -1) for r in org]
        while not done:
            i += 1
            t = o.time
            org, reward, done = o.step(i % 12)
            rows.append((t, user, 1, -1, i % 12, reward))
            rows += [(int(r['t']), user, 0, int(r['v']), -1, -1) for r in org]
        i += 1
        rows.append((o.time, user, 1, -1, i % 12, 0))
    want = np.array(rows, dtype=np.int64)
    got = frame_to_cols(df)
    for j, k in enumerate(('t', 'u', 'z', 'v', 'a', 'c')):
        assert np.array_equal(got[k], want[:, j]), k


def test_step_protocol_and_assertions():
    env = make_env(dict(random_seed=42))
    env.reset()
    with pytest.raises(AssertionError):
        env.step(3)                                   # abstract.py:158
    env.reset()
    obs, reward, done, info = env.step(None)
    assert reward is None and info == {} and len(obs.sessions()) >= 1
    assert obs.sessions()[0] == {'t': 0, 'u': 0, 'z': 'pageview', 'v': obs.sessions()[0]['v']}
    if not done:
        with pytest.raises(AssertionError):
            env.step(None)                            # abstract.py:174
    # the Getting-Started loop shape (cell 7): step_offline until done, agent=None
    env.reset()
    observation, reward, done = None, 0, False
    steps = 0
    while not done:
        action, observation, reward, done, info = env.step_offline(observation, reward, done)
        assert (action is None) == (steps == 0)
        if action is not None:
            assert set(action) == {'t', 'u', 'a', 'ps', 'ps-a'} and action['ps'] == 0.1
        steps += 1
    assert steps >= 1


def test_test_agent_and_verify_agents_quantiles():
    from oracle import oracle as orc
    over = dict(random_seed=5, num_products=40, K=10)
    env = make_env(over)
    agents = {
        'random': RandomAgent(Configuration({'num_products': 40, 'random_seed': 1,
                                             'with_ps_all': False})),
        'organic-count': OrganicUserEventCounterAgent(Configuration(
            {**gu.OUC_DEFAULTS, 'weight_history_function': None, 'num_products': 40,
             'random_seed': 2, 'with_ps_all': False})),
    }
    n = 3000
    df = recogym.verify_agents(env, n, agents)
    assert list(df.columns) == ['Agent', '0.025', '0.500', '0.975']
    cfg = Configuration({**recogym.env_1_args, **over})
    for i, (name, agent) in enumerate(agents.items()):
        o = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **agent.device_policy())
        o.generate_logs(n)
        c = o.counters()
        s, f = c['clicks'], c['bandit'] + c['phantom'] - c['clicks']
        assert df['0.500'][i] == beta.ppf(0.5, s + 1, f + 1)
        assert df['0.025'][i] == beta.ppf(0.025, s + 1, f + 1)
        # test_agent evaluates on the users AFTER its 10 offline users (fresh draws, as the reference's stream gives)
        o2 = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **agent.device_policy())
        o2.generate_logs(n, first_user_id=10)
        c2 = o2.counters()
        s2, f2 = c2['clicks'], c2['bandit'] + c2['phantom'] - c2['clicks']
        q = recogym.test_agent(env, agent, 10, n)
        assert q == (beta.ppf(0.5, s2 + 1, f2 + 1), beta.ppf(0.025, s2 + 1, f2 + 1),
                     beta.ppf(0.975, s2 + 1, f2 + 1))
        if i == 0:
            # two epochs without a random reset continue the id sequence: not the first epoch's counts doubled
            o3 = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **agent.device_policy())
            o3.generate_logs(n, first_user_id=10 + (10 + n))
            c3 = o3.counters()
            s3, f3 = s2 + c3['clicks'], f2 + c3['bandit'] + c3['phantom'] - c3['clicks']
            q2 = recogym.test_agent(env, agent, 10, n, num_epochs=2)
            assert q2 == (beta.ppf(0.5, s3 + 1, f3 + 1), beta.ppf(0.025, s3 + 1, f3 + 1), beta.ppf(0.975, s3 + 1, f3 + 1))
            assert q2 != (beta.ppf(0.5, 2 * s2 + 1, 2 * f2 + 1), beta.ppf(0.025, 2 * s2 + 1, 2 * f2 + 1),
                          beta.ppf(0.975, 2 * s2 + 1, 2 * f2 + 1))
    # the organic-count policy beats random on this environment (sanity)
    assert df['0.500'][1] > df['0.500'][0]
    # epochs re-key the env stream
    assert recogym.test_agent(env, agents['random'], 0, 500, num_epochs=2,
                              epoch_with_random_reset=True) != \
        recogym.test_agent(env, agents['random'], 0, 500, num_epochs=1)


def test_verify_agents_with_the_two_frozen_model_policies():
    """BASELINE config 4 in miniature: verify_agents A/B-tests a frozen BanditMFSquare table against a
    frozen LogReg model, both inside the device loop; the quantiles are those of the oracle's
    click counts for the same two policies."""
    from oracle import oracle as orc
    from recogym_amd.agents import LastViewTableAgent, LogregFrozenAgent
    meta, w = gu.load('philox_logreg')
    P = meta['env_args']['num_products']
    env = make_env(meta['env_args'])
    rng = np.random.RandomState(0)
    cfgp = Configuration({'num_products': P, 'with_ps_all': False})
    agents = {
        'bandit-mf': LastViewTableAgent.from_bandit_mf(cfgp, rng.randn(P, 5), rng.randn(P, 5)),
        'logreg-ips': LogregFrozenAgent(cfgp, w['logreg_coef'], w['logreg_intercept'], w['logreg_classes']),
    }
    n = 2000
    df = recogym.verify_agents(env, n, agents)
    cfg = Configuration({**recogym.env_1_args, **meta['env_args']})
    for i, agent in enumerate(agents.values()):
        pol = {k: v for k, v in agent.device_policy().items()}
        o = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
        o.generate_logs(n)
        c = o.counters()
        s_, f_ = c['clicks'], c['bandit'] + c['phantom'] - c['clicks']
        assert df['0.500'][i] == beta.ppf(0.5, s_ + 1, f_ + 1)
        assert df['0.975'][i] == beta.ppf(0.975, s_ + 1, f_ + 1)


def test_test_agent_trains_logreg_from_a_device_log_like_the_per_user_protocol(monkeypatch):
    """test_agent with the trainable LogregMulticlassIpsAgent: the fast path (offline log produced on the
    device in one go, vectorised training feed) returns the same CTR quantiles as the reference's
    per-user offline protocol (env.reset / step_offline / agent.train for every user), because both
    show the agent the same rows and the fit is deterministic."""
    from recogym_amd import bench_agents
    from recogym_amd.agents import LogregMulticlassIpsAgent, logreg_multiclass_ips_args
    over = {'random_seed': 11, 'num_products': 12, 'K': 5}
    cfg = Configuration({**logreg_multiclass_ips_args, 'num_products': 12, 'random_seed': 4, 'max_iter': 400})
    fast = recogym.test_agent(make_env(over), LogregMulticlassIpsAgent(cfg), 250, 1500)
    monkeypatch.delattr(LogregMulticlassIpsAgent, 'train_from_log')      # -> bench_agents._train, row by row
    slow = recogym.test_agent(make_env(over), LogregMulticlassIpsAgent(cfg), 250, 1500)
    assert fast == slow and 0.0 < fast[1] < fast[0] < fast[2] < 0.1


def test_test_agent_trains_bandit_mf_from_a_device_log_like_the_per_user_protocol(monkeypatch):
    """Same for the trainable BanditMFSquareAgent (torch RMSprop on mini-batches): log-fed training from a
    device-generated offline log == the per-user agent.train protocol, so test_agent returns the same
    quantiles either way."""
    import torch
    from recogym_amd.agents import BanditMFSquareAgent, bandit_mf_square_args
    over = {'random_seed': 21, 'num_products': 15, 'K': 5}
    cfg = Configuration({**bandit_mf_square_args, 'num_products': 15})
    rng = np.random.RandomState(1)
    ip, iu = rng.randn(15, 5).astype(np.float32), rng.randn(15, 5).astype(np.float32)
    fast = recogym.test_agent(make_env(over), BanditMFSquareAgent(cfg, ip, iu), 200, 1500, num_organic_offline_users=10)
    monkeypatch.delattr(BanditMFSquareAgent, 'train_from_log')
    slow = recogym.test_agent(make_env(over), BanditMFSquareAgent(cfg, ip, iu), 200, 1500, num_organic_offline_users=10)
    assert fast == slow and 0.0 < fast[0] < 0.1


def test_training_feed_on_device_equals_host_feed():
    """SURVEY §8f-3 on the GPU: the torch feed over Simulator.log_columns_device() builds the same
    CSR training set as the host (numpy) feed, which is pinned against the reference's train_data."""
    from recogym_amd.agents.feature_feed import train_data_from_log, train_data_from_log_torch
    from recogym_amd.sim import Simulator
    cfg = Configuration({**recogym.env_1_args, 'random_seed': 77, 'num_products': 300, 'K': 12})
    sim = Simulator(cfg, 3000, device='cuda:0', policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=3)
    sim.reset_users(0, 3000)
    sim.run()
    dev_cols = sim.log_columns_device()
    host_cols = sim.log_columns()
    sim.close()
    F, A, D, S = train_data_from_log(host_cols, 300)
    t = train_data_from_log_torch(dev_cols, 300)
    assert t['crow'].is_cuda
    assert np.array_equal(t['crow'].cpu().numpy(), F.indptr)
    assert np.array_equal(t['col'].cpu().numpy(), F.indices)
    assert np.array_equal(t['val'].cpu().numpy(), F.data)
    assert np.array_equal(t['actions'].cpu().numpy(), A) and np.array_equal(t['deltas'].cpu().numpy(), D)
    np.testing.assert_array_equal(t['pss'].cpu().numpy(), S)


def test_with_ps_all_for_the_organic_count_agent_on_the_device_path():
    """with_ps_all=True (organic_user_count.py:77-94): the batched device path logs (a, ps) and rebuilds the whole
    distribution of every bandit row from the log; it must equal the per-user path, where the agent's Python
    act() returns the vector itself.  Also: an action outside [0, P) raises IndexError like the reference."""
    over = dict(random_seed=11, num_products=30, K=8)
    cfgs = [dict(gu.OUC_DEFAULTS), dict(gu.OUC_DEFAULTS, epsilon=0.3), dict(gu.OUC_DEFAULTS, select_randomly=False),
            dict(gu.OUC_DEFAULTS, exploit_explore=False, epsilon=0.5, reverse_pop=True)]
    for oc in cfgs:
        def agent():
            return OrganicUserEventCounterAgent(Configuration({**oc, 'weight_history_function': None, 'num_products': 30,
                                                               'random_seed': 3, 'with_ps_all': True}))
        df_dev = make_env(over).generate_logs(40, agent())
        df_seq = make_env(over)._generate_logs_per_user(40, agent(), 0)
        assert len(df_dev) == len(df_seq)
        for k in ('t', 'u', 'a', 'c'):
            assert np.array_equal(df_dev[k].to_numpy(dtype=np.float64, na_value=np.nan),
                                  df_seq[k].to_numpy(dtype=np.float64, na_value=np.nan), equal_nan=True), k
        is_b = (df_dev['z'] == 'bandit').to_numpy()
        for i in np.flatnonzero(is_b):
            np.testing.assert_allclose(df_dev['ps-a'][i], df_seq['ps-a'][i], rtol=1e-15, atol=0)
            assert len(df_dev['ps-a'][i]) == 30
    env = make_env(over)
    env.reset(0)
    env.step(None)
    with pytest.raises(IndexError):
        env.step(30)


@pytest.mark.parametrize('name', ['philox_normal_time', 'philox_normal_time_ouc'])
def test_normal_time_generator_matches_the_reference(name):
    """NormalTimeGenerator (normal_time_generator.py:7-31): `t` is a float clock that advances by |N(mu, sigma)| per event
    and the omega drift after an event is scaled by that time delta (reco_env_v1.py:89-98).  The batched device path and
    the reset / step_offline path both reproduce the reference's own log — indices exactly, the clock at float32 resolution
    (the dtype of the reference's `t` column)."""
    from recogym_amd.envs.features.time import NormalTimeGenerator
    meta, want = gu.load(name)
    nt = meta['normal_time']

    def env():
        tg = NormalTimeGenerator(Configuration({'normal_time_mu': nt['mu'], 'normal_time_sigma': nt['sigma']}))
        return make_env({**meta['env_args'], 'time_generator': tg})

    df = env().generate_logs(meta['n_users'], make_agent(meta), meta['n_organic'])
    assert str(df['t'].dtype) == 'float32'
    np.testing.assert_allclose(df['t'].to_numpy(dtype=np.float64), want['time'], rtol=2e-7, atol=0)
    cols = frame_to_cols(df)
    for k in ('u', 'z', 'v', 'a', 'c'):
        assert np.array_equal(cols[k], want[k].astype(np.int64)), k
    np.testing.assert_allclose(cols['ps'], want['ps'], rtol=1e-12, equal_nan=True)
    # the per-user gym path (one transition per call, clock read back from the device)
    n = 12
    df_seq = env()._generate_logs_per_user(n, make_agent(meta), meta['n_organic'])
    keep = want['u'] < n + meta['n_organic']
    np.testing.assert_allclose(df_seq['t'].to_numpy(dtype=np.float64), want['time'][keep], rtol=2e-7, atol=0)
    cs = frame_to_cols(df_seq)
    for k in ('u', 'z', 'v', 'a', 'c'):
        assert np.array_equal(cs[k], want[k][keep].astype(np.int64)), k


def test_with_ps_all_under_the_normal_time_generator_uses_the_event_index():
    """OrganicUserEventCounter with with_ps_all, exploit_explore and epsilon > 0 under a NormalTimeGenerator: the `t`
    column is the float clock, but the explore flip of a bandit row is the policy draw of the user's EVENT INDEX (what
    the device and act() key it by).  The `ps-a` column rebuilt from the batched device log must be the distribution the
    logged (a, ps) came from: equal to the per-user path's vectors, and ps == ps-a[a] * (eps or 1 - eps)."""
    from recogym_amd.envs.features.time import NormalTimeGenerator
    over = dict(random_seed=17, num_products=30, K=8)
    oc = dict(gu.OUC_DEFAULTS, epsilon=0.3)

    def env():
        tg = NormalTimeGenerator(Configuration({'normal_time_mu': 0.0, 'normal_time_sigma': 1.0}))
        return make_env({**over, 'time_generator': tg})

    def agent():
        return OrganicUserEventCounterAgent(Configuration({**oc, 'weight_history_function': None, 'num_products': 30,
                                                           'random_seed': 5, 'with_ps_all': True}))
    df_dev = env().generate_logs(60, agent())
    df_seq = env()._generate_logs_per_user(60, agent(), 0)
    assert len(df_dev) == len(df_seq)
    for k in ('u', 'a', 'c'):
        assert np.array_equal(df_dev[k].to_numpy(dtype=np.float64, na_value=np.nan),
                              df_seq[k].to_numpy(dtype=np.float64, na_value=np.nan), equal_nan=True), k
    is_b = (df_dev['z'] == 'bandit').to_numpy()
    explored = 0
    for i in np.flatnonzero(is_b):
        np.testing.assert_allclose(df_dev['ps-a'][i], df_seq['ps-a'][i], rtol=1e-15, atol=0)
        pa = df_dev['ps-a'][i][int(df_dev['a'][i])]
        ps = float(df_dev['ps'][i])
        assert np.isclose(ps, 0.7 * pa, rtol=1e-12) or np.isclose(ps, 0.3 * pa, rtol=1e-12)
        explored += int(np.isclose(ps, 0.3 * pa, rtol=1e-12) and not np.isclose(ps, 0.7 * pa, rtol=1e-12))
    assert explored > 10          # the flips are exercised


def test_logreg_select_randomly_samples_like_the_reference():
    """LogregMulticlassIpsAgent with select_randomly=True (logreg_ips.py:61-66): the action is sampled from
    predict_proba with the model's own rng.  Host form (per-user path, HIP kernels underneath) and device form: both reproduce the
    log of the unmodified reference (its fitted model travels with the fixture, its draw injected as the addressed
    policy draw) — actions exactly, `ps` = the sampled class's probability to 1e-12."""
    from recogym_amd.agents import LogregFrozenAgent
    meta, want = gu.load('hostpath_logreg_random')
    cfg = Configuration({'num_products': meta['env_args']['num_products'], 'random_seed': meta['agent_args']['random_seed'],
                         'select_randomly': True, 'with_ps_all': False})
    agent = LogregFrozenAgent(cfg, want['logreg_coef'], want['logreg_intercept'], want['logreg_classes'])
    n = 30
    keep = want['u'] < n
    # the host form (one user at a time) ...
    cols = frame_to_cols(make_env(meta['env_args'])._generate_logs_per_user(n, agent, 0))
    for k in ('t', 'u', 'z', 'v', 'a', 'c'):
        assert np.array_equal(cols[k], want[k][keep].astype(np.int64)), k
    np.testing.assert_allclose(cols['ps'], want['ps'][keep], rtol=1e-12, equal_nan=True)
    # ... and the device form (round 5: rg_config.lr_select_randomly, k_logreg_sample): the whole fixture
    pol = agent.device_policy()
    assert pol is not None and pol['logreg']['select_randomly']
    cols = frame_to_cols(make_env(meta['env_args']).generate_logs(meta['n_users'], agent))
    for k in ('t', 'u', 'z', 'v', 'a', 'c'):
        assert np.array_equal(cols[k], want[k].astype(np.int64)), k
    np.testing.assert_allclose(cols['ps'], want['ps'], rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize('name', ['hostpath_ouc_weight_history', 'hostpath_ouc_weight_history_eps'])
def test_weight_history_agent_through_the_per_user_gym_path(name):
    """An OrganicUserEventCounterAgent with a `weight_history_function` (time-weighted views, agents/abstract.py:343-382) has no
    device policy: env.generate_logs walks it one user at a time (rg_sim_step_user underneath, the agent's act on the host).
    The log equals the unmodified reference's (fixture; counter RNG injected) on every column, `ps` bit for bit."""
    from recogym_amd.agents import organic_user_count_args
    meta, want = gu.load(name)
    aa = meta['agent_args']
    cfg = Configuration({**organic_user_count_args, 'num_products': meta['env_args']['num_products'], 'random_seed': aa['random_seed'],
                         'epsilon': aa.get('epsilon', 0.0), 'weight_history_function': gu.WEIGHT_FUNCS[aa['weight_history']]})
    agent = OrganicUserEventCounterAgent(cfg)
    assert agent.device_policy() is None
    n = 20
    df = make_env(meta['env_args']).generate_logs(n, agent)
    keep = want['u'] < n
    cols = frame_to_cols(df)
    for k in ('t', 'u', 'z', 'v', 'a', 'c'):
        assert np.array_equal(cols[k], want[k][keep].astype(np.int64)), k
    np.testing.assert_array_equal(cols['ps'], want['ps'][keep])


# ---- reco-gym-v0 (recogym/envs/reco_env_v0.py): the cluster toy model behind the same class surface ----
def make_env0(over):
    env = recogym.make('reco-gym-v0')
    env.init_gym({**recogym.env_0_args, **over})
    return env


@pytest.mark.parametrize('name', ['philox_env0_random', 'philox_env0_ouc', 'philox_env0_uniform'])
def test_env0_generate_logs_dataframe_equals_the_reference_log(name):
    """`gym.make('reco-gym-v0')` -> generate_logs on the device (rg_config.env_kind = 1) = the log the unmodified reference's
    RecoEnv0 produced with the same draws injected: every column, `t` the reference's constant 0."""
    meta, want = gu.load(name)
    assert meta['env_id'] == 'reco-gym-v0'
    env = make_env0(meta['env_args'])
    assert np.array_equal(env.click_probs.shape, (meta['env_args']['num_products'],) * 2)
    df = env.generate_logs(meta['n_users'], make_agent(meta), meta['n_organic'])
    assert list(df.columns) == ['t', 'u', 'z', 'v', 'a', 'c', 'ps', 'ps-a']
    got = frame_to_cols(df)
    assert (got['t'] == 0).all() and (want['time'] == 0).all()
    want_cols = {k: v for k, v in want.items() if k not in ('p_click', 'time')}
    assert_frames_match({**got, 't': want['t']}, want_cols, 1e-12)
    df2 = deepcopy(env).generate_logs(meta['n_users'], make_agent(meta), meta['n_organic'])
    pd.testing.assert_frame_equal(df, df2)


def test_env0_per_user_gym_path_equals_batched_path_and_an_arbitrary_agent_matches_the_oracle():
    from oracle import oracle as orc
    from recogym_amd.envs.static_params import draw_env0_tables
    meta, want = gu.load('philox_env0_random')
    n = 20
    env = make_env0(meta['env_args'])
    a = frame_to_cols(env._generate_logs_per_user(n, make_agent(meta), 0))
    b = frame_to_cols(make_env0(meta['env_args']).generate_logs(n, make_agent(meta), 0))
    assert_frames_match(a, b, 1e-12)
    # a Python-only agent through reset / step_offline, against the oracle's step API
    P = meta['env_args']['num_products']
    agent = FixedCycleAgent(Configuration({'num_products': P}))
    df = env.generate_logs(15, agent)
    cfg = Configuration({**recogym.env_0_args, **meta['env_args']})
    o = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, env0=draw_env0_tables(cfg))
    rows = []
    for user in range(15):
        o.reset(user)
        i = 0
        org, reward, done = o.step(None)
        rows += [(user, 0, int(r['v']), -1, -1) for r in org]
        while not done:
            i += 1
            org, reward, done = o.step(i % P)
            rows.append((user, 1, -1, i % P, reward))
            rows += [(user, 0, int(r['v']), -1, -1) for r in org]
        i += 1
        rows.append((user, 1, -1, i % P, 0))
    wantr = np.array(rows, dtype=np.int64)
    got = frame_to_cols(df)
    assert (got['t'] == 0).all()
    for j, k in enumerate(('u', 'z', 'v', 'a', 'c')):
        assert np.array_equal(got[k], wantr[:, j]), k


def test_env0_test_agent_runs_on_the_device_path():
    env = make_env0(dict(random_seed=5, num_products=20, num_clusters=4))
    agent = OrganicUserEventCounterAgent(Configuration({**gu.OUC_DEFAULTS, 'weight_history_function': None, 'num_products': 20,
                                                        'random_seed': 3, 'with_ps_all': False}))
    q = recogym.test_agent(deepcopy(env), deepcopy(agent), 200, 400)
    assert 0.0 < q[1] < q[0] < q[2] < 0.2
    assert q == recogym.test_agent(deepcopy(env), deepcopy(agent), 200, 400)


class StreamedAgent(Agent):
    """An agent with a sequential random stream of its own, like the reference's RandomAgent / EpsilonGreedy
    (`random_agent.py:14-20`): its users must see ONE stream consumed in order, not a copy of it each."""

    def __init__(self, config):
        super().__init__(config)
        self.rng = np.random.RandomState(config.random_seed)

    def act(self, observation, reward, done):
        return {**super().act(observation, reward, done),
                'a': int(self.rng.choice(self.config.num_products)), 'ps': 1.0 / self.config.num_products, 'ps-a': ()}


def test_an_agent_with_its_own_random_stream_keeps_the_sequential_path():
    """`generate_logs` batches B users per launch only for agents whose copies cannot share a random stream.
    With a RandomState inside, the default route is the reference's one-user-at-a-time loop: consecutive users draw from ONE
    stream (the batched path's per-slot copies would all replay the same draws), and the rows are those of the per-user path."""
    over = dict(random_seed=92, num_products=14, K=4)
    agent = StreamedAgent(Configuration({'num_products': 14, 'random_seed': 5}))
    env = make_env(over)
    a = env.generate_logs(60, deepcopy(agent))
    b = make_env(over)._generate_logs_per_user(60, deepcopy(agent), 0)
    pd.testing.assert_frame_equal(a, b)
    # what batching would have done: every user slot replays the same stream
    c = make_env(over)._generate_logs_batched(60, deepcopy(agent), 0)
    first_acts = lambda df: [int(df[(df['u'] == u) & (df['z'] == 'bandit')]['a'].iloc[0]) for u in range(40)]
    assert len(set(first_acts(c))) == 1 and len(set(first_acts(a))) > 3
    env.close()


@pytest.mark.parametrize('sizes', [(70, 0, 4096), (33, 5, 16), (3, 0, 1)])
def test_batched_episode_path_equals_the_per_user_path_and_the_oracle(sizes):
    """`generate_logs(n, arbitrary agent)` drives B users per rg_sim_step launch (a copy of the agent per user slot): the rows of
    the one-user-at-a-time path, several batches and warm-up users included, and the oracle's step API."""
    n, n_org, batch = sizes
    over = dict(random_seed=91, num_products=14, K=4)
    agent = FixedCycleAgent(Configuration({'num_products': 14}))
    env = make_env(over)
    a = env._generate_logs_batched(n, deepcopy(agent), n_org, batch=batch)
    b = make_env(over)._generate_logs_per_user(n, deepcopy(agent), n_org)
    assert [str(x) for x in a.dtypes] == [str(x) for x in b.dtypes]
    pd.testing.assert_frame_equal(a, b)
    assert len(env.generate_logs(n, deepcopy(agent), n_org)) == len(a)        # (the default route of generate_logs)

@pytest.mark.gpu
def test_rccl_communicator_and_all_reduce_on_this_box():
    """The only piece of the multi-GPU path a one-GPU box can run on the real backend: an RCCL ("nccl") process group of ONE rank on
    the device and an all_reduce of the counters' tensor through it (tools/rccl_probe.py; two ranks on one device are refused by RCCL
    — "Duplicate GPU detected", profiles/r6/rccl_probe_2ranks_one_device.txt — so the exchange between ranks is covered by the gloo
    tests only)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rccl_probe.py'), '1'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'RCCL ok: backend nccl, world 1' in r.stdout
