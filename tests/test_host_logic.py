"""CPU tests of the host side: policies' Python form vs the oracle, DataFrame assembly, user
sharding + counter all-reduce under a 2-process gloo group (the multi-GPU path, on CPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_util as gu
from oracle import oracle as orc
from recogym_amd import _abi, parallel, rng
from recogym_amd.agents import OrganicUserEventCounterAgent, RandomAgent
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.context import DefaultContext
from recogym_amd.envs.observation import Observation
from recogym_amd.envs.reco_env_v1 import (device_policy_of, env_1_args, rows_to_dataframe)
from recogym_amd.envs.session import OrganicSessions
from recogym_amd.sim import ROW_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_rng_matches_c_header():
    for c in [(0, 0, 0, 0), (1, 2, 3, 4), (2 ** 32 - 1, 7, 0, 1)]:
        for k in [(0, 0), (42, 0), (0xdeadbeef, 0x1234)]:
            assert rng.philox4x32_10(*c, *k) == tuple(int(x) for x in orc.philox(c, k))


@pytest.mark.parametrize('kind,agent_args', [
    ('random', dict(random_seed=5)),
    ('ouc', dict(random_seed=11)),
    ('ouc', dict(random_seed=12, epsilon=0.3)),
    ('ouc', dict(random_seed=13, select_randomly=False)),
    ('ouc', dict(random_seed=14, exploit_explore=False, epsilon=0.5, reverse_pop=True)),
])
def test_python_agents_equal_the_oracle_policies(kind, agent_args):
    """Drive the oracle env step by step with the host agents' actions: every (a, ps) the
    Python act returns must be the oracle's policy_act for the same (user, t, views)."""
    P = 20
    cfg = Configuration({**env_1_args, 'random_seed': 3, 'num_products': P, 'K': 6})
    aa = {'num_products': P, 'with_ps_all': False, **agent_args}
    if kind == 'random':
        agent = RandomAgent(Configuration(aa))
        meta = dict(agent='random', agent_args=aa)
    else:
        agent = OrganicUserEventCounterAgent(Configuration({**gu.OUC_DEFAULTS, **aa,
                                                             'weight_history_function': None}))
        meta = dict(agent='ouc', agent_args=aa)
    pol = gu.policy_args(meta)
    assert device_policy_of(agent)['policy'] == pol['policy']
    env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **pol)
    n_checked = 0
    for user in range(40):
        env.reset(user)
        agent.reset()
        rows, reward, done = env.step(None)
        while True:
            sess = OrganicSessions()
            for r in rows:
                sess.next(DefaultContext(int(r['t']), user), int(r['v']))
            obs = Observation(DefaultContext(env.time, user), sess)
            want_a, want_ps = env.policy_act()
            got = agent.act(obs, reward, done)
            assert got['a'] == want_a and got['t'] == env.time and got['u'] == user
            assert got['ps'] == pytest.approx(want_ps, rel=1e-15)
            n_checked += 1
            if done:
                break
            rows, reward, done = env.step(got['a'])
    assert n_checked > 1000


def test_rows_to_dataframe_matches_reference_schema():
    rows = np.zeros(4, dtype=ROW_DTYPE)
    rows['u'] = [0, 0, 0, 1]; rows['t'] = [0, 1, 2, 0]; rows['z'] = [0, 1, 1, 0]
    rows['v'] = [3, -1, -1, 9]; rows['a'] = [-1, 4, 2, -1]; rows['c'] = [-1, 1, 0, -1]
    rows['ps'] = [np.nan, 0.1, 0.1, np.nan]
    df = rows_to_dataframe(rows, 10)
    assert list(df.columns) == ['t', 'u', 'z', 'v', 'a', 'c', 'ps', 'ps-a']
    assert str(df['t'].dtype) == 'float32' and str(df['c'].dtype) == 'float32'
    assert str(df['u'].dtype) == 'UInt16' and str(df['v'].dtype) == 'UInt16'
    assert str(df['a'].dtype) == 'UInt16' and str(df['ps'].dtype) == 'float64'
    assert list(df['z']) == ['organic', 'bandit', 'bandit', 'organic']
    assert df['v'].isna().tolist() == [False, True, True, False]
    assert df['a'].isna().tolist() == [True, False, False, True]
    assert np.isnan(df['c'][0]) and df['c'][1] == 1.0
    assert df['ps-a'][0] is None and df['ps-a'][1] == ()
    # the reference's CTR reduction works on it unchanged (bench_agents.py:204-206)
    rewards = df[~np.isnan(df['a'])]['c'] if False else df[df['z'] == 'bandit']['c']
    assert rewards.sum() == 1 and rewards.shape[0] == 2
    wide = rows.copy(); wide['u'] = [0, 0, 0, 70000]
    assert str(rows_to_dataframe(wide, 100000)['u'].dtype) == 'UInt32'
    assert str(rows_to_dataframe(wide, 100000)['v'].dtype) == 'UInt32'


def test_shard_ranges_tile_the_users():
    for n in (0, 1, 7, 10_000_000):
        for ws in (1, 2, 3, 8):
            rs = [parallel.shard_range(n, r, ws) for r in range(ws)]
            assert rs[0][0] == 0 and sum(c for _, c in rs) == n
            for (f0, c0), (f1, _) in zip(rs, rs[1:]):
                assert f1 == f0 + c0
            assert max(c for _, c in rs) - min(c for _, c in rs) <= 1


WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import torch.distributed as dist
from oracle import oracle as orc
from recogym_amd import _abi, bench_agents, evaluate_agent
from recogym_amd.agents import RandomAgent
from recogym_amd.envs.configuration import Configuration
from recogym_amd.envs.reco_env_v1 import env_1_args

class OracleBackedEnv:
    """Stands in for RecoEnv1.simulate on a CPU-only box: same contract, oracle arithmetic.
    (Test double only — the product has no CPU path.)"""
    def __init__(self, cfg): self.config = cfg
    def __deepcopy__(self, memo): return OracleBackedEnv(self.config)
    def simulate(self, num_users, agent=None, num_organic_users=0, first_user_id=0, log=True, device=None):
        pol = agent.device_policy()
        env = orc.OracleEnv(self.config, rng_mode=orc.RNG_PHILOX, **pol)
        env.generate_logs(num_users, first_user_id=first_user_id)
        c = env.counters()
        class S:
            device = None
            def close(self): pass
        return dict(clicks=c['clicks'], bandit=c['bandit'], phantom=c['phantom'], organic=c['organic']), S()

dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + sys.argv[2],
                        rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
cfg = Configuration({**env_1_args, 'random_seed': 9, 'num_products': 30, 'K': 8})
agent = RandomAgent(Configuration({'num_products': 30, 'random_seed': 4, 'with_ps_all': False}))
s, f = bench_agents.evaluate_counts(OracleBackedEnv(cfg), agent, 301)
q = bench_agents.test_agent(OracleBackedEnv(cfg), agent, 0, 301)
df = evaluate_agent.verify_agents(OracleBackedEnv(cfg), 301, {'r': agent})
print(json.dumps(dict(rank=dist.get_rank(), s=s, f=f, q=list(q), v=float(df['0.500'][0]))))
dist.destroy_process_group()
'''


def test_two_rank_gloo_counter_allreduce_equals_single_process(tmp_path):
    """world_size 2 over gloo: each rank simulates its shard of user ids, the all-reduced
    (successes, failures) and the Beta quantiles equal the single-process run on every rank."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 2000)
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), '2'],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        outs.append(eval(o.strip().splitlines()[-1].replace('true', 'True')))
    # single process reference
    cfg = Configuration({**env_1_args, 'random_seed': 9, 'num_products': 30, 'K': 8})
    env1 = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, policy=_abi.RG_POLICY_RANDOM_AGENT,
                         policy_seed=4)
    env1.generate_logs(301)
    c = env1.counters()
    for o in outs:
        assert o['s'] == c['clicks'] and o['f'] == c['bandit'] + c['phantom'] - c['clicks']
    assert outs[0]['q'] == outs[1]['q'] and outs[0]['v'] == outs[1]['v'] == outs[0]['q'][0]


def test_log_materialisation_paths_agree():
    """the device-decoded column path (columns_to_dataframe) and the host decode of raw
    rg_event records (raw_log_to_dataframe) build the same DataFrame as the reference-shaped two-step
    path rows_to_dataframe(decode_rows(raw)) — dtypes included, ids beyond the UInt16 ceiling too."""
    import pandas as pd
    from recogym_amd import _abi
    from recogym_amd.envs.reco_env_v1 import (columns_to_dataframe, raw_log_to_dataframe,
                                              rows_to_dataframe)
    from recogym_amd.sim import decode_rows
    rng = np.random.RandomState(0)
    for n, users, P, uniform in ((5000, 300, 50, None), (5000, 70000, 70000, 1.0 / 70000)):
        raw = np.zeros((n, 4), dtype=np.uint32)
        raw[:, 0] = np.sort(rng.randint(0, users, n))
        raw[:, 1] = rng.randint(0, 300, n)
        is_b = rng.rand(n) < 0.7
        click = is_b & (rng.rand(n) < 0.1)
        raw[:, 2] = (rng.randint(0, P, n) | np.where(is_b, _abi.RG_EV_BANDIT, 0)
                     | np.where(click, _abi.RG_EV_CLICK, 0)).astype(np.uint32)
        raw[:, 3] = rng.rand(n).astype(np.float32).view(np.uint32)
        want = rows_to_dataframe(decode_rows(raw.view(np.int32), uniform), P)
        pd.testing.assert_frame_equal(raw_log_to_dataframe(raw.view(np.int32), P, uniform), want)
        code = raw[:, 2]
        idx = (code & _abi.RG_EV_INDEX_MASK).astype(np.int32)
        ps = raw[:, 3].copy().view(np.float32).astype(np.float64) if uniform is None else np.full(n, uniform)
        cols = dict(t=raw[:, 1].astype(np.float32), u=raw[:, 0].astype(np.int32), is_bandit=is_b,
                    v=np.where(is_b, 0, idx).astype(np.int32), a=np.where(is_b, idx, 0).astype(np.int32),
                    c=np.where(is_b, click.astype(np.float32), np.float32('nan')).astype(np.float32),
                    ps=np.where(is_b, ps, np.nan))
        pd.testing.assert_frame_equal(columns_to_dataframe(cols, P), want)


def test_frozen_logreg_act_is_sklearn_predict():
    """LogregFrozenAgent.act (the host form of RG_POLICY_LOGREG_FROZEN) on the model stored with the
    reference's LogReg fixture: for random view-count vectors the action equals sklearn's own
    predict() computed from the same arrays (decision_function of a CSR row: ascending products,
    multiply then add, intercept last) — including the two-class form."""
    import warnings
    from scipy import sparse
    from sklearn.linear_model import LogisticRegression
    import golden_util as gu
    from recogym_amd.agents import LogregFrozenAgent
    from recogym_amd.envs.configuration import Configuration
    from recogym_amd.envs.context import DefaultContext
    from recogym_amd.envs.observation import Observation
    from recogym_amd.envs.session import OrganicSessions
    meta, cols = gu.load('philox_logreg')
    P = meta['env_args']['num_products']
    lr = LogisticRegression()
    lr.coef_, lr.intercept_, lr.classes_ = cols['logreg_coef'], cols['logreg_intercept'], cols['logreg_classes']
    rng = np.random.RandomState(5)
    models = [lr]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        X = rng.poisson(0.3, size=(200, P)); y = (X[:, 0] + rng.rand(200) > 0.8).astype(int) * 7 + 3
        models.append(LogisticRegression(max_iter=500).fit(X, y))          # two classes {3, 10}
    for m in models:
        agent = LogregFrozenAgent.from_sklearn(Configuration({'num_products': P}), m)
        for trial in range(300):
            agent.reset()
            sessions = OrganicSessions()
            counts = np.zeros(P, dtype=np.int16)
            for t in range(rng.randint(0, 15)):
                v = int(rng.randint(0, P))
                sessions.next(DefaultContext(t, 0), v)
                counts[v] += 1
            a = agent.act(Observation(DefaultContext(99, 0), sessions), 0, False)
            assert a['a'] == m.predict(sparse.csr_matrix(counts.reshape(1, P)))[0] and a['ps'] == 1.0


def test_self_launch_starts_its_own_ranks():
    """`python bench.py --gpus N` must work without a launcher (the driver starts it that way): the script re-runs
    itself under torch.distributed.run.  Same code path (parallel.self_launch + parallel.init_from_env) on CPU/gloo
    with 2 ranks: shards tile the id range and the all-reduce sees both ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'launch_probe.py'), '--procs', '2', '--users', '1001'],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('PROBE')][-1]
    assert line == f'PROBE world=2 users=1001 ranks=2 idsum={sum(range(1001))}', line
    # and a single process needs no launcher at all
    out1 = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'launch_probe.py'), '--procs', '1'],
                          capture_output=True, text=True, timeout=300, env=env)
    assert out1.returncode == 0 and 'PROBE world=1 users=1001 ranks=1' in out1.stdout, out1.stderr[-2000:]


def test_bench_workloads_and_cli_are_consistent():
    """bench.py's argument surface the driver relies on (--gpus/--steps/--warmup, defaults that finish in
    minutes) and its workload table (metric text follows the workload; strong scaling keeps the total fixed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.WORKLOADS['c3'][0] == dict(num_products=10000, K=20, sigma_omega=0.0) and bench.WORKLOADS['c3'][3] == 'ouc'
    assert bench.WORKLOADS['c4shard'][1] * 8 == bench.WORKLOADS['c4shard'][2] == 10_000_000
    from recogym_amd import parallel
    shards = [parallel.shard_range(10_000_000, r, 8) for r in range(8)]
    assert sum(c for _, c in shards) == 10_000_000 and shards[0][0] == 0
    assert all(shards[i][0] + shards[i][1] == shards[i + 1][0] for i in range(7))


def test_gym_registration_hook(tmp_path, monkeypatch):
    """recogym_amd.register_with_gym() registers 'reco-gym-v1' with an importable `gym` (here: a stand-in with the
    registration surface the reference uses, recogym/__init__.py:35-45) and leaves an existing registration alone
    unless forced; without gym it reports False."""
    import recogym_amd
    pkg = tmp_path / 'gym'
    (pkg / 'envs').mkdir(parents=True)
    (pkg / '__init__.py').write_text('from .envs import registration\n')
    (pkg / 'envs' / '__init__.py').write_text('from . import registration\n')
    (pkg / 'envs' / 'registration.py').write_text(
        'registry = {}\n'
        'def register(id, entry_point, **kw):\n'
        '    if id in registry: raise ValueError("Cannot re-register id: " + id)\n'
        '    registry[id] = entry_point\n')
    for m in [k for k in sys.modules if k == 'gym' or k.startswith('gym.')]:
        monkeypatch.delitem(sys.modules, m)
    monkeypatch.syspath_prepend(str(tmp_path))
    assert recogym_amd.register_with_gym() is True
    import gym.envs.registration as reg
    assert reg.registry['reco-gym-v1'] == 'recogym_amd.envs.reco_env_v1:RecoEnv1'
    reg.registry['reco-gym-v1'] = 'recogym.envs.reco_env_v1:RecoEnv1'        # the reference got there first
    assert recogym_amd.register_with_gym() is False
    assert recogym_amd.register_with_gym(force=True) is True
    assert reg.registry['reco-gym-v1'] == 'recogym_amd.envs.reco_env_v1:RecoEnv1'


def test_division_by_reciprocal_is_correctly_rounded():
    """The device computes count / sum of the organic-count policy as q = RN(c y), r = c - sum q (one fma),
    RN(q + r y) with y = RN(1 / sum) — one true division per act.  That equals IEEE division for the operands the
    policy has (positive integers below 2^32): checked against exact rational arithmetic, exhaustively for small
    operands and on random large ones."""
    from fractions import Fraction
    import random

    def rn(fr):                     # nearest double of an exact rational (Python rounds int/int division correctly)
        return fr.numerator / fr.denominator

    def dev(c, s):
        y = 1.0 / s
        q = c * y
        r = rn(Fraction(c) - Fraction(s) * Fraction(q))          # fma(-s, q, c): exact, then rounded once
        assert Fraction(r) == Fraction(c) - Fraction(s) * Fraction(q)      # ... and in fact representable
        return rn(Fraction(q) + Fraction(r) * Fraction(y))       # fma(r, y, q)

    for s in range(1, 400):
        for c in range(1, s + 1):
            assert dev(float(c), float(s)) == c / s, (c, s)
    rnd = random.Random(7)
    for _ in range(40000):
        s = rnd.randrange(1, 1 << rnd.randrange(1, 33))
        c = rnd.randrange(1, s + 1)
        assert dev(float(c), float(s)) == c / s, (c, s)


def test_bandit_mf_table_is_built_in_row_blocks():
    """LastViewTableAgent.from_bandit_mf: the P x P logit matrix is taken a block of rows at a time (2 GB as one
    broadcast at P = 10^4); blocks of any size give the table and winning logits of the one-shot form."""
    from recogym_amd.agents import LastViewTableAgent
    rng = np.random.RandomState(4)
    P, E = 700, 5
    Ep, Eu = rng.randn(P, E), rng.randn(P, E)
    ag = LastViewTableAgent.from_bandit_mf(Configuration({'num_products': P}), Ep, Eu)
    full = (Eu.astype(np.float32)[:, None, :] * Ep.astype(np.float32)[None, :, :]).sum(axis=2)
    assert np.array_equal(ag.table, full.argmax(axis=1))
    assert np.array_equal(ag.ps, full.max(axis=1).astype(np.float64))
    big = LastViewTableAgent.from_bandit_mf(Configuration({'num_products': 6000}), rng.randn(6000, 5), rng.randn(6000, 5))
    assert big.table.shape == (6000,) and big.table.max() < 6000


def test_reco_env_is_a_gym_env_where_gym_is_importable(tmp_path, monkeypatch):
    """RecoEnv1 derives from gym.Env when a gym with that surface is importable (the reference's AbstractEnv does,
    recogym/envs/abstract.py:46-57) — here a stand-in package with `Env` and `spaces.Discrete` — and from object otherwise;
    init_gym sets action_space / observation_space = Discrete(num_products)."""
    import importlib
    pkg = tmp_path / 'gym'
    (pkg / 'envs').mkdir(parents=True)
    (pkg / '__init__.py').write_text('from . import spaces\nfrom .envs import registration\n'
                                     'class Env:\n    metadata = {}\n    def __init__(self):\n        self.gym_env_init_ran = True\n')
    (pkg / 'spaces.py').write_text('class Discrete:\n    def __init__(self, n):\n        self.n = n\n        self.from_gym = True\n')
    (pkg / 'envs' / '__init__.py').write_text('from . import registration\n')
    (pkg / 'envs' / 'registration.py').write_text('registry = {}\ndef register(id, entry_point, **kw):\n    registry[id] = entry_point\n')
    for m in [k for k in sys.modules if k == 'gym' or k.startswith('gym.')]:
        monkeypatch.delitem(sys.modules, m)
    monkeypatch.syspath_prepend(str(tmp_path))
    import recogym_amd.envs.reco_env_v1 as mod
    try:
        mod = importlib.reload(mod)
        import gym
        assert issubclass(mod.RecoEnv1, gym.Env)
        env = mod.RecoEnv1()
        assert env.gym_env_init_ran
        # (init_gym draws the tables on the host; no device needed)
        env.init_gym({**mod.env_1_args, 'random_seed': 3, 'num_products': 12})
        assert env.action_space.n == 12 and env.action_space.from_gym and env.observation_space.n == 12
    finally:
        for m in [k for k in sys.modules if k == 'gym' or k.startswith('gym.')]:
            sys.modules.pop(m, None)
        sys.path.remove(str(tmp_path))
        mod = importlib.reload(mod)
    assert mod._EnvBase is object or mod._EnvBase.__module__.split('.')[0] in ('gym', 'gymnasium')


def test_blocked_generate_beta_pairing_equals_the_reference():
    """SURVEY.md 8f-4, table construction at large P: flip_index_blocked (row blocks of Gamma Gamma^T, a bounded candidate
    set per round, no P x P matrix) gives (a) the pairing of the reference's full-argsort algorithm as restated by flip_index,
    for block sizes and candidate budgets that force several rounds, and (b) the pairing the UNMODIFIED reference computed
    at P = 2 000 (tests/golden/flips_p2000.npz, made by tests/make_golden.py flips)."""
    import json
    from numpy.random.mtrand import RandomState
    from recogym_amd.envs import static_params as sp
    rng = np.random.RandomState(11)
    for P, K, F, rows, keep in [(300, 8, 20, 64, 50), (1500, 20, 60, 170, 500), (1000, 5, 300, 333, 256), (64, 5, 10, 7, 16)]:
        G = rng.normal(size=(P, K))
        assert np.array_equal(sp.flip_index(G, F), sp.flip_index_blocked(G, F, block_rows=rows, keep=keep)), (P, K, F)
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'flips_p2000.npz'))
    meta = json.loads(str(z['meta']))
    a = meta['env_args']
    r = RandomState(a['random_seed'])
    Gamma = r.normal(size=(a['num_products'], a['K']))
    assert np.array_equal(sp.flip_index_blocked(Gamma, meta['flips'], block_rows=300), z['index'])
    assert np.array_equal(sp.flip_index(Gamma, meta['flips']), z['index'])
    # draw_tables switches to the blocked form above FLIP_BLOCKED_ABOVE products: same tables either way
    cfg = Configuration({**{k: v for k, v in a.items()}, 'num_products': 600, 'number_of_flips': 25})
    full = sp.draw_tables(cfg)
    old, sp.FLIP_BLOCKED_ABOVE = sp.FLIP_BLOCKED_ABOVE, 100
    try:
        blocked = sp.draw_tables(cfg)
    finally:
        sp.FLIP_BLOCKED_ABOVE = old
    assert all(np.array_equal(x, y) for x, y in zip(full, blocked))


def test_test_agent_with_cache_trains_from_the_pickled_offline_log(tmp_path, monkeypatch):
    """test_agent(with_cache=True) (reference: bench_agents.py:17-63,90-166): the offline log is generate_logs' DataFrame pickled
    by a hash of the env configuration and the user counts; the agent is trained from it — the first call generates and writes
    it, the second reads it back; agent.train sees every bandit row with the organic session before it, the user's last row
    closing the episode, an organic-only user as one call without an action."""
    from recogym_amd import bench_agents
    from recogym_amd.envs.reco_env_v1 import rows_to_dataframe
    monkeypatch.setenv('RECOGYM_CACHE_DIR', str(tmp_path))
    cfg = Configuration({**env_1_args, 'random_seed': 13, 'num_products': 12, 'K': 4})
    calls = dict(generate=0)

    class Env:                      # test double for RecoEnv1 on a CPU-only box: same contract, oracle arithmetic
        config = cfg
        agent = None
        def __deepcopy__(self, memo): return self
        def generate_logs(self, num_offline_users, agent=None, num_organic_offline_users=0, first_user_id=0):
            calls['generate'] += 1
            rows = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(num_offline_users, num_organic_offline_users, first_user_id)
            return rows_to_dataframe(rows, cfg.num_products)
        def simulate(self, num_users, agent=None, num_organic_users=0, first_user_id=0, log=True, device=None):
            env = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX, **agent.device_policy())
            env.generate_logs(num_users, first_user_id=first_user_id)
            c = env.counters()
            class S:
                device = None
                def close(self): pass
            return dict(clicks=c['clicks'], bandit=c['bandit'], phantom=c['phantom'], organic=c['organic']), S()

    seen = []

    class Learner(RandomAgent):
        def train(self, observation, action, reward, done=False):
            seen.append((len(observation.sessions()), None if action is None else action['a'], reward, done))

    agent = Learner(Configuration({'num_products': 12, 'random_seed': 4, 'with_ps_all': False}))
    q1 = bench_agents.test_agent(Env(), agent, 30, 50, num_organic_offline_users=5, with_cache=True)
    assert calls['generate'] == 1 and len(os.listdir(tmp_path)) == 1
    first = list(seen)
    seen.clear()
    q2 = bench_agents.test_agent(Env(), agent, 30, 50, num_organic_offline_users=5, with_cache=True)
    assert calls['generate'] == 1 and q1 == q2 and seen == first           # read back, same training calls
    want = orc.OracleEnv(cfg, rng_mode=orc.RNG_PHILOX).generate_logs(30, 5)
    assert len(first) == int((want['z'] == 1).sum()) + 5                   # every bandit row (phantom included) + 5 warm-up users
    assert sum(1 for s in first if s[3]) == 35                             # one closing call per user
    assert [s for s in first if s[1] is None] == [s for s in first[:5]] and all(s[0] > 0 for s in first[:5])
    assert bench_agents._cache_file_name(Env(), 5, 30) != bench_agents._cache_file_name(Env(), 5, 31)


@pytest.mark.parametrize('name', ['hostpath_ouc_weight_history', 'hostpath_ouc_weight_history_eps'])
def test_weight_history_function_reproduces_the_reference_agent(name):
    """`weight_history_function` (organic_user_count.py:19; ViewsFeaturesProvider with history, agents/abstract.py:343-382):
    the agent acts on time-weighted views instead of counts.  The logs of the unmodified reference (counter RNG injected, an
    exponential / an inverse weight function, with and without exploration) are replayed row by row through
    OrganicUserEventCounterAgent.act: every logged action exactly, every propensity bit for bit (float32 arithmetic where the
    reference's is).  No device involved: a weighted agent has no device policy (per-user host path)."""
    meta, cols = gu.load(name)
    aa = meta['agent_args']
    from recogym_amd.agents import organic_user_count_args
    cfg = Configuration({**organic_user_count_args, 'num_products': meta['env_args']['num_products'], 'random_seed': aa['random_seed'],
                         'epsilon': aa.get('epsilon', 0.0), 'weight_history_function': gu.WEIGHT_FUNCS[aa['weight_history']]})
    agent = OrganicUserEventCounterAgent(cfg)
    assert agent.device_policy() is None and device_policy_of(agent) is None
    cur, sess, checked = None, OrganicSessions(), 0
    for i in range(len(cols['t'])):
        u, t = int(cols['u'][i]), int(cols['t'][i])
        if u != cur:
            cur, sess = u, OrganicSessions()
            agent.reset()
        if cols['z'][i] == 0:
            sess.next(DefaultContext(t, u), int(cols['v'][i]))
            continue
        got = agent.act(Observation(DefaultContext(t, u), sess), 0, False)
        assert got['a'] == int(cols['a'][i]), (i, got['a'], int(cols['a'][i]))
        assert float(got['ps']) == float(cols['ps'][i]), (i, got['ps'], cols['ps'][i])
        sess = OrganicSessions()
        checked += 1
    assert checked > 5000


def test_weight_history_function_in_the_logreg_training_feed_and_act():
    """LogregMulticlassIpsAgent with a `weight_history_function` in its config (agents/abstract.py:199-263: train_data's
    weighted features; :343-382: the feature provider's).  Fixture of the unmodified reference (exponential weights): (a) the
    training set our feed builds from the reference's training log equals the reference's train_data — CSR structure exactly,
    values bit for bit (float32 sums widened to float64); (b) the fitted model's logged actions are reproduced by
    LogregFrozenAgent.act over the weighted features, row by row."""
    from scipy import sparse
    from recogym_amd.agents import LogregFrozenAgent
    from recogym_amd.agents.feature_feed import train_data_from_log
    meta, cols = gu.load('hostpath_logreg_weight_history')
    wf = gu.WEIGHT_FUNCS[meta['agent_args']['weight_history']]
    P = meta['env_args']['num_products']
    is_b = cols['trainlog_z'] == 1
    log = dict(t=cols['trainlog_t'].astype(np.float32), u=cols['trainlog_u'].astype(np.int32), is_bandit=is_b,
               v=np.where(is_b, 0, cols['trainlog_v']), a=np.where(is_b, cols['trainlog_a'], 0),
               c=np.where(is_b, cols['trainlog_c'], np.nan).astype(np.float32), ps=cols['trainlog_ps'])
    feats, actions, deltas, pss = train_data_from_log(log, P, weight_history_function=wf)
    feats = sparse.csr_matrix(feats); feats.sort_indices()
    assert np.array_equal(feats.indptr, cols['train_indptr']) and np.array_equal(feats.indices, cols['train_indices'])
    assert np.array_equal(feats.data.astype(np.float64), cols['train_data'])
    assert np.array_equal(actions, cols['train_actions']) and np.array_equal(deltas, cols['train_deltas'])
    np.testing.assert_array_equal(pss, cols['train_pss'])
    cfg = Configuration({'num_products': P, 'random_seed': meta['agent_args']['random_seed'], 'select_randomly': False,
                         'with_ps_all': False, 'weight_history_function': wf})
    agent = LogregFrozenAgent(cfg, cols['logreg_coef'], cols['logreg_intercept'], cols['logreg_classes'])
    assert agent.device_policy() is None
    cur, sess, checked = None, OrganicSessions(), 0
    for i in range(len(cols['t'])):
        u, t = int(cols['u'][i]), int(cols['t'][i])
        if u != cur:
            cur, sess = u, OrganicSessions()
            agent.reset()
        if cols['z'][i] == 0:
            sess.next(DefaultContext(t, u), int(cols['v'][i]))
            continue
        got = agent.act(Observation(DefaultContext(t, u), sess), 0, False)
        assert got['a'] == int(cols['a'][i]) and got['ps'] == 1.0, (i, got['a'], int(cols['a'][i]))
        sess = OrganicSessions()
        checked += 1
    assert checked > 3000


def test_generate_logs_batches_only_agents_without_a_sequential_stream_of_their_own():
    """`generate_logs` may advance an arbitrary Python agent B users at a time with a COPY of it per user slot
    only where the copies cannot share a random stream — an agent that owns a RandomState / Generator (the reference's
    RandomAgent, EpsilonGreedy, NnIpsAgent: `random_agent.py:14-20`) keeps the reference's one-user-at-a-time loop, whose
    single stream is consumed sequentially (`abstract.py:292-316`); `agent.batch_safe` overrides the inspection.  The copies
    share large read-only model state instead of multiplying it by the batch size."""
    from copy import deepcopy
    from recogym_amd.envs.reco_env_v1 import batch_safe, _shared_state_memo, _approx_owned_bytes

    class Streamed:
        def __init__(self):
            self.rng = np.random.RandomState(3)

    class Nested:
        def __init__(self):
            self.policy = {'explore': [Streamed()]}

    class Generator:
        def __init__(self):
            self.g = np.random.default_rng(5)

    class Model:
        def __init__(self):
            self.coef = np.zeros((1000, 100))
            self.counts = np.zeros(4)

    assert not batch_safe(Streamed()) and not batch_safe(Nested()) and not batch_safe(Generator())
    assert batch_safe(Model())
    s = Streamed(); s.batch_safe = True
    m = Model(); m.batch_safe = False
    assert batch_safe(s) and not batch_safe(m)
    m = Model()
    memo = _shared_state_memo(m)
    c = deepcopy(m, dict(memo))
    assert c.coef is m.coef and c.counts is not m.counts           # 800 KB shared, the per-user state owned
    assert _approx_owned_bytes(m, memo) < 4096
    # the framework's own agents draw by address (user, t): batch-safe by construction
    from recogym_amd.agents import RandomAgent, OrganicUserEventCounterAgent
    from recogym_amd.envs.configuration import Configuration
    assert batch_safe(RandomAgent(Configuration({'num_products': 10, 'random_seed': 1, 'with_ps_all': True})))
