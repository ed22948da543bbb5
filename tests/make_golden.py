"""Generates tests/golden/* by running the UNMODIFIED reference in this container.

    python tests/make_golden.py            # needs /root/reference; ~1-2 minutes

Two families (see oracle/recogym_oracle.c header):
  mt_*.npz      logs of the reference on its own sequential MT19937 stream (as shipped)
  philox_*.npz  logs of the reference's arithmetic with the counter RNG injected via `env.rng`
  notebook_getting_started.json  the known-answer vectors stored in the reference's
                `Getting Started.ipynb` (cells 7 and 9), transcribed from the notebook file
                itself, plus the same sequences re-run here.
Every fixture stores the env args / agent args it was made with, so the tests rebuild the
identical configuration without the reference.
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh  # noqa: E402

GOLDEN = os.path.join(HERE, 'golden')

BASE = dict(num_products=10, num_users=100, prob_leave_bandit=0.01, prob_leave_organic=0.01,
            prob_bandit_to_organic=0.05, prob_organic_to_bandit=0.25, normalize_beta=False,
            with_ps_all=False, K=5, sigma_omega_initial=1, sigma_omega=0.1, number_of_flips=0,
            sigma_mu_organic=3, change_omega_for_bandits=False, num_clusters=2, phi_var=0.1)


def make_agent(kind, agent_args):
    rh.import_reference()
    from recogym import Configuration
    if kind == 'random':
        from recogym.agents import RandomAgent, random_args
        return RandomAgent(Configuration({**random_args, **agent_args}))
    if kind == 'ouc':
        from recogym.agents import OrganicUserEventCounterAgent, organic_user_count_args
        args = dict(agent_args)
        if args.get('weight_history'):       # by name: tests/golden_util.WEIGHT_FUNCS
            import golden_util as gu
            args['weight_history_function'] = gu.WEIGHT_FUNCS[args.pop('weight_history')]
        return OrganicUserEventCounterAgent(
            Configuration({**organic_user_count_args, **args}))
    if kind == 'bmf':
        import torch
        from recogym.agents import BanditMFSquare, bandit_mf_square_args
        torch.manual_seed(agent_args.get('torch_seed', 0))
        return BanditMFSquare(Configuration({**bandit_mf_square_args,
                                             'num_products': agent_args['num_products'],
                                             'embed_dim': agent_args.get('embed_dim', 5)}))
    raise ValueError(kind)


def save(name, arrays, meta):
    small = {}
    for k, v in arrays.items():
        if k in ('z', 'c'):
            small[k] = v.astype(np.int8)
        elif k in ('ps', 'p_click', 'time') or k.startswith('bmf_'):
            small[k] = v
        else:
            small[k] = v.astype(np.int32)
    path = os.path.join(GOLDEN, name + '.npz')
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **small)
    print(f'{name}: {len(arrays["t"])} rows -> {os.path.getsize(path) / 1024:.0f} KiB')


def run_case(name, env_over, n_users, n_organic=0, agent_kind=None, agent_args=None,
             injected=False, normal_time=None, env_id='reco-gym-v1'):
    args = {**BASE, **env_over}
    if env_id == 'reco-gym-v0':       # env_0_args (reco_env_v0.py:7-13): no latent-factor keys
        args = {k: v for k, v in args.items() if k not in ('K', 'sigma_omega_initial', 'sigma_omega', 'number_of_flips',
                                                           'sigma_mu_organic', 'change_omega_for_bandits')}
    if normal_time is not None:       # {'mu':, 'sigma':}: the reference's NormalTimeGenerator, passed the way init_gym takes it
        rh.import_reference()
        from recogym import Configuration
        from recogym.envs.features.time import NormalTimeGenerator
        args = {**args, 'time_generator': NormalTimeGenerator(Configuration(
            {'random_seed': args['random_seed'], 'normal_time_mu': normal_time['mu'], 'normal_time_sigma': normal_time['sigma']}))}
    env = rh.make_reference_env(args, env_id)
    agent = None
    agent_args = dict(agent_args or {})
    if agent_kind:
        agent_args.setdefault('num_products', args['num_products'])
        agent = make_agent(agent_kind, agent_args)
    p_click = None
    if injected:
        rng = rh.inject_counter_rng(env, None if agent_kind == 'bmf' else agent, agent_args.get('random_seed'))
    df = env.generate_logs(n_users, agent, n_organic)
    times = None
    if normal_time is not None or env_id == 'reco-gym-v0':       # 't' holds the generator's clock (reco-gym-v0: the constant reset() set): keep it as `time`, and the event index as t
        times = df['t'].to_numpy(dtype=np.float64)
        df = df.copy()
        df['t'] = df.groupby(df['u'].astype('int64')).cumcount().astype(np.float32)
    arrays = rh.log_to_arrays(df)
    if times is not None:
        arrays['time'] = times
    if injected:
        # p_click of every REAL bandit row, in row order (phantom rows are never drawn)
        pc = np.full(len(df), np.nan)
        is_b = arrays['z'] == 1
        last_of_user = np.r_[arrays['u'][1:] != arrays['u'][:-1], True]
        real = is_b & ~last_of_user
        # organic-only users end with an organic row, main users with the phantom bandit row
        assert real.sum() == len(rng.p_click_log), (real.sum(), len(rng.p_click_log))
        pc[real] = rng.p_click_log
        arrays['p_click'] = pc
    meta = dict(env_args={k: v for k, v in args.items() if k != 'time_generator'}, n_users=n_users, n_organic=n_organic,
                agent=agent_kind, agent_args=agent_args, rng='philox' if injected else 'mt')
    if env_id != 'reco-gym-v1':
        meta['env_id'] = env_id
    if normal_time is not None:
        meta['normal_time'] = normal_time
    if agent_kind == 'bmf':      # the (untrained) embeddings the frozen device policy is built from
        arrays['bmf_product_embedding'] = agent.product_embedding.weight.detach().numpy().astype(np.float64)
        arrays['bmf_user_embedding'] = agent.user_embedding.weight.detach().numpy().astype(np.float64)
    save(name, arrays, meta)


def normal_time_goldens():
    """The reference with its NormalTimeGenerator (float clock, drift scaled by the time delta, reco_env_v1.py:89-98)."""
    S = dict(random_seed=42)
    run_case('philox_normal_time', {**S, 'num_products': 30, 'K': 6, 'sigma_omega': 0.2}, 150, n_organic=5, injected=True,
             normal_time=dict(mu=0.3, sigma=1.2))
    run_case('philox_normal_time_ouc', {**S, 'num_products': 40, 'K': 8, 'sigma_omega': 0.1, 'change_omega_for_bandits': True}, 120,
             agent_kind='ouc', agent_args=dict(random_seed=9), injected=True, normal_time=dict(mu=0.0, sigma=1.0))


def notebook_goldens():
    """Transcribe cells 7 and 9 of `Getting Started.ipynb` and re-run them on the reference."""
    nb = json.load(open(os.path.join(rh.REFERENCE_ROOT, 'Getting Started.ipynb')))
    out = {}
    for cell_no in (7, 9):
        text = ''.join(nb['cells'][cell_no]['outputs'][0]['text'])
        steps = []
        for line in text.strip().split('\n'):
            m = re.match(r'Step: (\d+) - Action: (.*) - Observation: (\[.*\]) - Reward: (\w+)', line)
            act = m.group(2)
            if act.startswith('{'):
                a = int(re.search(r"'a': (\d+)", act).group(1))
                t = int(re.search(r"'t': (\d+)", act).group(1))
            elif act == 'None':
                a, t = None, None
            else:
                a, t = int(act), None
            views = [[int(x), int(y)] for x, y in
                     re.findall(r"\{'t': (\d+), 'u': 0, 'z': 'pageview', 'v': (\d+)\}", m.group(3))]
            rew = None if m.group(4) == 'None' else int(m.group(4))
            steps.append(dict(step=int(m.group(1)), action=a, action_t=t, views=views, reward=rew))
        out[f'cell{cell_no}_stored'] = steps

    # the same two cells executed here on the unmodified reference
    args = {**BASE, 'random_seed': 42}
    env = rh.make_reference_env(args)
    env.reset()
    observation, reward, done = None, 0, False
    rerun7 = []
    i = 0
    while not done:
        action, observation, reward, done, info = env.step_offline(observation, reward, done)
        rerun7.append(dict(step=i, action=None if action is None else int(action['a']),
                           action_t=None if action is None else int(action['t']),
                           views=[[int(s['t']), int(s['v'])] for s in observation.sessions()],
                           reward=None if reward is None else int(reward)))
        i += 1
    out['cell7_rerun'] = rerun7
    out['cell7_done'] = bool(done)
    actions = [None, 1, 2, 3, 4, 5]
    env.reset()
    done = False
    rerun9 = []
    i = 0
    while not done and i < len(actions):
        observation, reward, done, info = env.step(actions[i])
        rerun9.append(dict(step=i, action=actions[i], action_t=None,
                           views=[[int(s['t']), int(s['v'])] for s in observation.sessions()],
                           reward=None if reward is None else int(reward)))
        i += 1
    out['cell9_rerun'] = rerun9
    out['env_args'] = args
    with open(os.path.join(GOLDEN, 'notebook_getting_started.json'), 'w') as f:
        json.dump(out, f, indent=1)
    same7 = out['cell7_stored'] == out['cell7_rerun']
    same9 = out['cell9_stored'] == out['cell9_rerun']
    print(f'notebook goldens: cell7 stored==rerun {same7}, cell9 stored==rerun {same9}')
    assert same7 and same9


def run_logreg_case(name, env_over, n_train, n_users, injected=True, select_randomly=False, weight_history=None):
    """LogregMulticlassIpsAgent of the UNMODIFIED reference (agents/logreg_ips.py): trained by the
    reference's own build() (train_data + sklearn fit) on a uniform-policy log of n_train users that the
    reference generated, then run through generate_logs with the counter RNG injected into the env (the
    agent itself draws nothing: select_randomly = False).  The fitted arrays travel with the fixture."""
    rh.import_reference()
    from recogym import Configuration
    from recogym.agents import LogregMulticlassIpsAgent, logreg_multiclass_ips_args
    args = {**BASE, **env_over}
    train_env = rh.make_reference_env({**args, 'random_seed': args['random_seed'] + 1000})
    train_log = train_env.generate_logs(n_train)
    extra = {}
    if weight_history:       # ViewsFeaturesProvider / train_data with a weight function (agents/abstract.py:199-263,343-382)
        import golden_util as gu
        extra['weight_history_function'] = gu.WEIGHT_FUNCS[weight_history]
    agent = LogregMulticlassIpsAgent(Configuration({**logreg_multiclass_ips_args,
                                                    'num_products': args['num_products'],
                                                    'random_seed': 7, 'select_randomly': select_randomly, **extra}))
    d = agent.model_builder.data
    for _, r in train_log.iterrows():                       # ModelBuilder.train's bookkeeping, row by row
        bandit = r['z'] == 'bandit'
        d['t'].append(int(r['t'])); d['u'].append(int(r['u'])); d['z'].append(r['z'])
        d['v'].append(None if bandit else int(r['v']))
        d['a'].append(int(r['a']) if bandit else None)
        d['c'].append(int(r['c']) if bandit else None)
        d['ps'].append(float(r['ps']) if bandit else None)
    env = rh.make_reference_env(args)
    if injected:                                            # else: the reference exactly as shipped (MT19937)
        # select_randomly = False draws nothing; True samples from predict_proba with the model's own rng (seed 7)
        rh.inject_counter_rng(env, agent if select_randomly else None, 7 if select_randomly else None)
    df = env.generate_logs(n_users, agent)                  # first act() builds the model
    lr = agent.model.logreg
    arrays = rh.log_to_arrays(df)
    train_extra = {}
    if weight_history:       # the training log and the training set the reference built from it (weighted features)
        tl = rh.log_to_arrays(train_log)
        feats, t_actions, t_deltas, t_pss = agent.model_builder.train_data()
        from scipy import sparse as _sp
        dense_dtype = None if _sp.issparse(feats) else str(feats.dtype)          # (the logreg builder asks for the dense form)
        feats = _sp.csr_matrix(feats); feats.sort_indices()
        train_extra = dict(trainlog_t=tl['t'].astype(np.int32), trainlog_u=tl['u'].astype(np.int32), trainlog_z=tl['z'].astype(np.int8),
                           trainlog_v=tl['v'].astype(np.int32), trainlog_a=tl['a'].astype(np.int32), trainlog_c=tl['c'].astype(np.int8),
                           trainlog_ps=tl['ps'], train_data=feats.data, train_indices=feats.indices.astype(np.int32),
                           train_indptr=feats.indptr.astype(np.int64), train_actions=t_actions, train_deltas=t_deltas, train_pss=t_pss)
    arrays['logreg_coef'] = np.asarray(lr.coef_, dtype=np.float64)
    arrays['logreg_intercept'] = np.asarray(lr.intercept_, dtype=np.float64)
    arrays['logreg_classes'] = np.asarray(lr.classes_, dtype=np.int64)
    meta = dict(env_args=args, n_users=n_users, n_organic=0, agent='logreg',
                agent_args=dict(n_train=n_train, clicks_in_training=int(np.nansum(train_log['c'].to_numpy(dtype=float))),
                                select_randomly=bool(select_randomly), random_seed=7, weight_history=weight_history),
                rng='philox' if injected else 'mt')
    small = {}
    for k, v in arrays.items():
        if k in ('z', 'c'):
            small[k] = v.astype(np.int8)
        elif k == 'ps' or k.startswith('logreg_'):
            small[k] = v
        else:
            small[k] = v.astype(np.int32)
    path = os.path.join(GOLDEN, name + '.npz')
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **small, **train_extra)
    print(f'{name}: {len(arrays["t"])} rows, {len(lr.classes_)} classes, actions used '
          f'{len(np.unique(arrays["a"][arrays["z"] == 1]))} -> {os.path.getsize(path) / 1024:.0f} KiB')


def bandit_mf_training_golden(fixture):
    """BanditMFSquare of the UNMODIFIED reference (agents/bandit_mf.py) trained on an existing fixture log
    through its own train(): one call per bandit row with the observation that preceded it (what the
    offline protocol of bench_agents.py:168-190 does).  Stores the initial and the trained embeddings."""
    rh.import_reference()
    import torch
    from recogym import Configuration, Observation, DefaultContext
    from recogym.envs.session import OrganicSessions
    from recogym.agents import BanditMFSquare, bandit_mf_square_args
    f = np.load(os.path.join(GOLDEN, fixture + '.npz'), allow_pickle=False)
    meta = json.loads(str(f['meta']))
    P = meta['env_args']['num_products']
    torch.manual_seed(5)
    agent = BanditMFSquare(Configuration({**bandit_mf_square_args, 'num_products': P}))
    init_p = agent.product_embedding.weight.detach().numpy().copy()
    init_u = agent.user_embedding.weight.detach().numpy().copy()
    sessions, cur = OrganicSessions(), None
    for i in range(len(f['t'])):
        u, t = int(f['u'][i]), int(f['t'][i])
        if u != cur:
            cur, sessions = u, OrganicSessions()
        if int(f['z'][i]) == 0:
            sessions.next(DefaultContext(t, u), int(f['v'][i]))
        else:
            action = {'t': t, 'u': u, 'a': int(f['a'][i]), 'ps': float(f['ps'][i]), 'ps-a': ()}
            agent.train(Observation(DefaultContext(t, u), sessions), action, int(f['c'][i]), False)
            sessions = OrganicSessions()
    path = os.path.join(GOLDEN, 'bandit_mf_training_' + fixture + '.npz')
    np.savez_compressed(path, init_product=init_p, init_user=init_u,
                        product=agent.product_embedding.weight.detach().numpy(),
                        user=agent.user_embedding.weight.detach().numpy(),
                        steps=np.array(agent.curr_step))
    print(f'bandit_mf_training_{fixture}: {agent.curr_step} train calls -> {os.path.getsize(path) / 1024:.0f} KiB')


def c5_trained_models(name='c5_trained_p100', P=100, K=20, n_train=1000, seed=42):
    """The two policies of BASELINE config 5 FITTED BY THE REFERENCE'S OWN CODE, at a size it can train: a uniform-policy
    log of `n_train` users generated by the unmodified reference (env defaults, sigma_omega = 0.1), then
    LogregMulticlassIpsAgent's build() (train_data + sklearn multinomial fit, agents/logreg_ips.py:89-99) and BanditMFSquare's
    train() (one call per bandit row with the observation that preceded it, agents/bandit_mf.py:89-126).  Only the fitted
    arrays travel (bench.py's `c5trained` workload and the sampled-oracle check run them; margins between class scores — hence
    the refine rate of the fp16 screen — are those of fitted models, not of random weights)."""
    rh.import_reference()
    import torch
    from recogym import Configuration, Observation, DefaultContext
    from recogym.envs.session import OrganicSessions
    from recogym.agents import (BanditMFSquare, bandit_mf_square_args, LogregMulticlassIpsAgent,
                                logreg_multiclass_ips_args)
    args = {**BASE, 'random_seed': seed, 'num_products': P, 'K': K}
    train_env = rh.make_reference_env({**args, 'random_seed': seed + 1000})
    log = train_env.generate_logs(n_train)
    lr_agent = LogregMulticlassIpsAgent(Configuration({**logreg_multiclass_ips_args, 'num_products': P, 'random_seed': 7,
                                                       'select_randomly': False}))
    torch.manual_seed(5)
    mf_agent = BanditMFSquare(Configuration({**bandit_mf_square_args, 'num_products': P}))
    d = lr_agent.model_builder.data
    sessions, cur = OrganicSessions(), None
    t_arr, u_arr, z_arr = log['t'].to_numpy(), log['u'].to_numpy(), log['z'].to_numpy()
    v_arr, a_arr, c_arr, ps_arr = log['v'].to_numpy(), log['a'].to_numpy(), log['c'].to_numpy(), log['ps'].to_numpy()
    for i in range(len(log)):
        t, u, bandit = int(t_arr[i]), int(u_arr[i]), z_arr[i] == 'bandit'
        d['t'].append(t); d['u'].append(u); d['z'].append(z_arr[i])
        d['v'].append(None if bandit else int(v_arr[i]))
        d['a'].append(int(a_arr[i]) if bandit else None)
        d['c'].append(int(c_arr[i]) if bandit else None)
        d['ps'].append(float(ps_arr[i]) if bandit else None)
        if u != cur:
            cur, sessions = u, OrganicSessions()
        if not bandit:
            sessions.next(DefaultContext(t, u), int(v_arr[i]))
        else:
            action = {'t': t, 'u': u, 'a': int(a_arr[i]), 'ps': float(ps_arr[i]), 'ps-a': ()}
            mf_agent.train(Observation(DefaultContext(t, u), sessions), action, int(c_arr[i]), False)
            sessions = OrganicSessions()
    _, model = lr_agent.model_builder.build()               # the reference's train_data + sklearn fit
    lr = model.logreg
    path = os.path.join(GOLDEN, name + '.npz')
    meta = dict(env_args=args, n_train=n_train, train_rows=int(len(log)), clicks_in_training=int(np.nansum(c_arr.astype(float))),
                bandit_mf_train_calls=int(mf_agent.curr_step), logreg_classes=int(len(lr.classes_)),
                what='policies of BASELINE config 5 fitted by the unmodified reference (logreg_ips.py build, bandit_mf.py train)')
    np.savez_compressed(path, meta=np.array(json.dumps(meta)),
                        logreg_coef=np.asarray(lr.coef_, dtype=np.float64), logreg_intercept=np.asarray(lr.intercept_, dtype=np.float64),
                        logreg_classes=np.asarray(lr.classes_, dtype=np.int64),
                        bmf_product_embedding=mf_agent.product_embedding.weight.detach().numpy(),
                        bmf_user_embedding=mf_agent.user_embedding.weight.detach().numpy())
    print(f'{name}: {len(log)} training rows, {meta["clicks_in_training"]} clicks, {len(lr.classes_)} classes, '
          f'{mf_agent.curr_step} BanditMF train calls -> {os.path.getsize(path) / 1024:.0f} KiB')


def flips_index_golden(name='flips_p2000', P=2000, K=20, flips=150, seed=42):
    """generate_beta's pairing of the UNMODIFIED reference (reco_env_v1.py:147-168) at a size where its P x P argsort is still
    cheap: the permutation `index` with beta = Gamma[index] recovered from the reference env's own tables (SURVEY.md 8f-4:
    recogym_amd.envs.static_params.flip_index_blocked must reproduce it without the P x P matrix)."""
    rh.import_reference()
    args = {**BASE, 'random_seed': seed, 'num_products': P, 'K': K, 'number_of_flips': flips}
    env = rh.make_reference_env(args)
    G, B = np.asarray(env.Gamma), np.asarray(env.beta)
    # row i of beta is row index[i] of Gamma: match rows through a hash of their bytes (rows are distinct normal draws)
    where = {G[j].tobytes(): j for j in range(P)}
    index = np.array([where[B[i].tobytes()] for i in range(P)], dtype=np.int32)
    assert (index != np.arange(P)).sum() == 2 * flips
    path = os.path.join(GOLDEN, name + '.npz')
    np.savez_compressed(path, meta=np.array(json.dumps(dict(env_args=args, flips=flips))), index=index)
    print(f'{name}: {flips} flips at P = {P} -> {os.path.getsize(path) / 1024:.0f} KiB')


def train_feed_golden(fixture):
    """The training set the UNMODIFIED reference builds from a log: the rows of an
    existing fixture are pushed through ModelBuilder.train's bookkeeping (agents/abstract.py:55-83:
    one dict entry per organic session row / per action) and AbstractFeatureProvider.train_data
    (agents/abstract.py:190-279) is run on them."""
    rh.import_reference()
    from recogym import Configuration
    from recogym.agents.abstract import AbstractFeatureProvider
    f = np.load(os.path.join(GOLDEN, fixture + '.npz'), allow_pickle=False)
    meta = json.loads(str(f['meta']))
    fp = AbstractFeatureProvider(Configuration({'num_products': meta['env_args']['num_products']}),
                                 is_sparse=True)
    for i in range(len(f['t'])):
        bandit = int(f['z'][i]) == 1
        fp.data['t'].append(int(f['t'][i]))
        fp.data['u'].append(int(f['u'][i]))
        fp.data['z'].append('bandit' if bandit else 'organic')
        fp.data['v'].append(None if bandit else int(f['v'][i]))
        fp.data['a'].append(int(f['a'][i]) if bandit else None)
        fp.data['c'].append(int(f['c'][i]) if bandit else None)
        fp.data['ps'].append(float(f['ps'][i]) if bandit else None)
    feats, actions, deltas, pss = fp.train_data()
    feats = feats.tocsr()
    feats.sort_indices()
    path = os.path.join(GOLDEN, 'train_feed_' + fixture + '.npz')
    np.savez_compressed(path, data=feats.data, indices=feats.indices, indptr=feats.indptr,
                        shape=np.array(feats.shape), actions=actions, deltas=deltas, pss=pss,
                        data_dtype=np.array(str(feats.dtype)))
    print(f'train_feed_{fixture}: {feats.shape[0]} rows, nnz {feats.nnz}, dtype {feats.dtype} -> '
          f'{os.path.getsize(path) / 1024:.0f} KiB')


def main():
    assert rh.reference_available(), 'needs /root/reference'
    os.makedirs(GOLDEN, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'train_feed':     # only the §8f-3 fixtures (fast)
        for fx in ('philox_ouc', 'mt_random_agent'):
            train_feed_golden(fx)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'bmf_training':
        bandit_mf_training_golden('mt_random_agent')
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'mt_bmf':
        run_case('mt_bandit_mf', {'random_seed': 42, 'num_products': 40, 'K': 10}, 120, agent_kind='bmf',
                 agent_args=dict(torch_seed=3, embed_dim=5))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'logreg':         # only the §8f-1 LogReg fixture
        run_logreg_case('philox_logreg', {'random_seed': 42, 'num_products': 30, 'K': 8}, 1500, 200)
        run_logreg_case('mt_logreg', {'random_seed': 42, 'num_products': 30, 'K': 8}, 1500, 120, injected=False)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'r2':             # round-2 additions only
        S = dict(random_seed=42)
        # the reference's dtype ceiling (np.int16 product ids), BASELINE config 4's K and drift
        run_case('philox_p32767_k64_drift', {**S, 'num_products': 32767, 'K': 64, 'sigma_omega': 0.1}, 10,
                 injected=True)
        # BASELINE config 3's table shape with its agent in the loop
        run_case('philox_ouc_p10000', {**S, 'num_products': 10000, 'K': 20, 'sigma_omega': 0.0}, 40,
                 agent_kind='ouc', agent_args=dict(random_seed=11), injected=True)
        normal_time_goldens()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'weight_history':  # ViewsFeaturesProvider's time-weighted history (agents/abstract.py:343-382)
        run_case('hostpath_ouc_weight_history', {'random_seed': 42, 'num_products': 30, 'K': 8}, 120, agent_kind='ouc',
                 agent_args=dict(random_seed=77, weight_history='exp_0.2'), injected=True)
        run_logreg_case('hostpath_logreg_weight_history', {'random_seed': 42, 'num_products': 20, 'K': 6}, 150, 60, weight_history='exp_0.2')
        run_case('hostpath_ouc_weight_history_eps', {'random_seed': 43, 'num_products': 30, 'K': 8}, 80, agent_kind='ouc',
                 agent_args=dict(random_seed=78, weight_history='inverse', epsilon=0.2), injected=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'env0':            # reco-gym-v0, the cluster toy model (reco_env_v0.py)
        V0 = 'reco-gym-v0'
        run_case('mt_env0_default', {'random_seed': 42}, 400, env_id=V0)                       # env_0_args as shipped
        run_case('mt_env0_ouc', {'random_seed': 7, 'num_products': 12, 'num_clusters': 3}, 300, n_organic=20, agent_kind='ouc',
                 agent_args=dict(random_seed=5), env_id=V0)
        run_case('philox_env0_random', {'random_seed': 11, 'num_products': 20, 'num_clusters': 4}, 300, agent_kind='random',
                 agent_args=dict(random_seed=9), injected=True, env_id=V0)
        run_case('philox_env0_ouc', {'random_seed': 13, 'num_products': 30, 'num_clusters': 2, 'phi_var': 0.5}, 250, n_organic=15,
                 agent_kind='ouc', agent_args=dict(random_seed=3), injected=True, env_id=V0)
        run_case('philox_env0_uniform', {'random_seed': 17, 'num_products': 64, 'num_clusters': 8,
                                         'prob_organic_to_bandit': 0.4, 'prob_bandit_to_organic': 0.1}, 200, injected=True, env_id=V0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'flips':           # generate_beta's pairing at P = 2 000
        flips_index_golden()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'c5_trained':      # the fitted policies of bench.py's c5trained workload
        c5_trained_models()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'logreg_random':
        run_logreg_case('hostpath_logreg_random', {'random_seed': 42, 'num_products': 10, 'K': 5}, 800, 100, select_randomly=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'normal_time':
        normal_time_goldens()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'mt_edges':         # round 4: edge shapes on the reference as shipped (oracle pin only)
        run_case('mt_p2_k1', dict(random_seed=3, num_products=2, K=1), 300, n_organic=5)
        run_case('mt_large_drift', dict(random_seed=4, num_products=25, K=4, sigma_omega=0.8, sigma_omega_initial=2.0), 120)
        run_case('mt_ouc_organic_users', dict(random_seed=5, num_products=12, K=3), 120, n_organic=30,
                 agent_kind='ouc', agent_args=dict(random_seed=31))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'transition':       # round 4: a transition matrix and initial scales that are not the defaults
        # (prob_leave_bandit is set but the reference never reads it: prob_leave_organic fills the stop column of both rows,
        # reco_env_v1.py:54-61 — these logs pin that on the unmodified reference)
        T = dict(random_seed=77, num_products=40, K=6, prob_leave_bandit=0.2, prob_leave_organic=0.03, prob_bandit_to_organic=0.15,
                 prob_organic_to_bandit=0.4, sigma_omega_initial=0.5, sigma_mu_organic=1.5, sigma_omega=0.2)
        run_case('mt_transition_probs', T, 150, n_organic=10)
        run_case('philox_transition_probs', T, 150, n_organic=10, agent_kind='ouc', agent_args=dict(random_seed=21), injected=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'long_runs':        # round 5: long bandit runs at sigma_omega = 0 (k_walk2's helpers take up to
        # seven events of a run per iteration), long view histories (60 products: lines fill up), organic-only users among them
        T = dict(random_seed=91, K=20, sigma_omega=0.0, prob_leave_organic=0.004, prob_bandit_to_organic=0.03, prob_organic_to_bandit=0.3)
        run_case('philox_long_runs_ouc', {**T, 'num_products': 60}, 110, n_organic=6, agent_kind='ouc', agent_args=dict(random_seed=17), injected=True)
        run_case('philox_long_runs_random', {**T, 'num_products': 400, 'random_seed': 92}, 90, agent_kind='random', agent_args=dict(random_seed=18), injected=True)
        return
    notebook_goldens()
    S = dict(random_seed=42)
    # --- reference as shipped (sequential MT19937) ---
    run_case('mt_config1', {**S, 'sigma_omega': 0.0}, 1000)                    # BASELINE config 1
    run_case('mt_drift_organic_users', {**S}, 200, n_organic=20)
    run_case('mt_random_agent', {**S, 'num_products': 50, 'K': 20}, 200,
             agent_kind='random', agent_args=dict(random_seed=7))
    run_case('mt_ouc', {**S, 'num_products': 50, 'K': 20}, 200,
             agent_kind='ouc', agent_args=dict(random_seed=11))
    run_case('mt_ouc_eps', {**S, 'num_products': 50, 'K': 20}, 150,
             agent_kind='ouc', agent_args=dict(random_seed=12, epsilon=0.3))
    run_case('mt_ouc_argmax', {**S, 'num_products': 50, 'K': 20}, 100,
             agent_kind='ouc', agent_args=dict(random_seed=13, select_randomly=False))
    run_case('mt_ouc_noexplore_revpop', {**S, 'num_products': 20, 'K': 8}, 100,
             agent_kind='ouc', agent_args=dict(random_seed=14, exploit_explore=False,
                                               epsilon=0.5, reverse_pop=True))
    run_case('mt_flips_normbeta', {**S, 'num_products': 30, 'number_of_flips': 5,
                                   'normalize_beta': True}, 150)
    run_case('mt_change_omega', {**S, 'change_omega_for_bandits': True}, 150)
    run_case('mt_bandit_mf', {**S, 'num_products': 40, 'K': 10}, 120, agent_kind='bmf',
             agent_args=dict(torch_seed=3, embed_dim=5))
    # --- reference arithmetic + injected counter RNG ---
    run_case('philox_p10', {**S}, 300, n_organic=10, injected=True)
    run_case('philox_p10_sigma0', {**S, 'sigma_omega': 0.0}, 300, injected=True)
    run_case('philox_p1000_k20', {**S, 'num_products': 1000, 'K': 20, 'sigma_omega': 0.0}, 80,
             injected=True)
    run_case('philox_p10000_k20', {**S, 'num_products': 10000, 'K': 20}, 24, injected=True)
    run_case('philox_p2000_k64', {**S, 'num_products': 2000, 'K': 64}, 30, injected=True)
    run_case('philox_p257_k7', {**S, 'num_products': 257, 'K': 7, 'random_seed': 5}, 100,
             injected=True)
    run_case('philox_random_agent', {**S, 'num_products': 100, 'K': 20}, 120,
             agent_kind='random', agent_args=dict(random_seed=5), injected=True)
    run_case('philox_ouc', {**S, 'num_products': 50, 'K': 20}, 150,
             agent_kind='ouc', agent_args=dict(random_seed=11), injected=True)
    run_case('philox_ouc_eps', {**S, 'num_products': 50, 'K': 20}, 120,
             agent_kind='ouc', agent_args=dict(random_seed=12, epsilon=0.2), injected=True)
    run_case('philox_flips_normbeta', {**S, 'num_products': 30, 'number_of_flips': 5,
                                       'normalize_beta': True}, 120, injected=True)
    run_case('philox_change_omega', {**S, 'change_omega_for_bandits': True}, 120, injected=True)
    run_case('philox_bandit_mf', {**S, 'num_products': 40, 'K': 10}, 150, agent_kind='bmf',
             agent_args=dict(torch_seed=3, embed_dim=5), injected=True)
    run_case('philox_p32767_k64_drift', {**S, 'num_products': 32767, 'K': 64, 'sigma_omega': 0.1}, 10,
             injected=True)
    run_case('philox_ouc_p10000', {**S, 'num_products': 10000, 'K': 20, 'sigma_omega': 0.0}, 40,
             agent_kind='ouc', agent_args=dict(random_seed=11), injected=True)
    normal_time_goldens()
    for fx in ('philox_ouc', 'mt_random_agent'):
        train_feed_golden(fx)
    run_logreg_case('philox_logreg', {**S, 'num_products': 30, 'K': 8}, 1500, 200)
    run_logreg_case('mt_logreg', {**S, 'num_products': 30, 'K': 8}, 1500, 120, injected=False)
    run_logreg_case('hostpath_logreg_random', {**S, 'num_products': 10, 'K': 5}, 800, 100, select_randomly=True)
    bandit_mf_training_golden('mt_random_agent')


if __name__ == '__main__':
    main()
