"""test_agent — reference: recogym/bench_agents.py:215-251 (+ _collect_stats, :66-212).

Train an agent on offline users, evaluate it on online users, return the (median, 2.5 %, 97.5 %)
quantiles of the Beta(successes + 1, failures + 1) CTR posterior.  The evaluation is the hot
path: with a device-capable agent it runs as one batched simulation per rank and only the
counters leave the GPU; ranks (if torch.distributed is initialised) simulate disjoint user-id
ranges and all-reduce {clicks, impressions}.  Training keeps the reference's per-user protocol
(agent.train on every observation) and is skipped for agents that have nothing to learn.
"""
import hashlib
import json
import os
import pickle
from copy import deepcopy
from pathlib import Path

from scipy.stats.distributions import beta

from . import parallel
from .envs.reco_env_v1 import device_policy_of


def evaluate_counts(env, agent, num_users, first_user_id=0):
    """-> (successes, failures) of `agent` over users first_user_id .. first_user_id + num_users - 1, summed
    over all ranks.
    successes = sum of c over bandit rows, failures = bandit rows - successes; the phantom row
    of every user counts as a failure exactly as in the reference (bench_agents.py:203-206)."""
    rank, ws, _ = parallel.world()
    if device_policy_of(agent) is not None:
        first, count = parallel.shard_range(num_users, rank, ws)
        clicks = shown = 0
        dev = None
        if count:
            cnt, sim = env.simulate(count, agent, first_user_id=first_user_id + first, log=False)
            clicks, shown = cnt['clicks'], cnt['bandit'] + cnt['phantom']
            dev = sim.device
            sim.close()
        clicks, shown = parallel.all_reduce_counts([clicks, shown], dev)
        return clicks, shown - clicks
    # arbitrary Python agent: per-user path on rank 0's env (no sharding: agents are stateful)
    data = env.generate_logs(num_users, agent, first_user_id=first_user_id)
    rewards = data[data['z'] == 'bandit']['c']
    successes = int(rewards.sum())
    return successes, int(rewards.shape[0]) - successes


def _learns(agent):
    """Does agent.train do anything?  (Agent.train is a no-op for the device policies.)"""
    from .agents.abstract import Agent
    fn = getattr(type(agent), 'train', None)
    return fn is not None and fn is not Agent.train and hasattr(agent, 'train')


def _train(env, agent, num_offline_users, num_organic_offline_users, first_user_id=0):
    """The reference's offline protocol (bench_agents.py:168-190)."""
    uid = first_user_id
    for _ in range(num_organic_offline_users):
        env.reset(uid)
        uid += 1
        obs, _, _, _ = env.step(None)
        agent.train(obs, None, None, True)
    for _ in range(num_offline_users):
        env.reset(uid)
        uid += 1
        new_obs, _, done, reward = env.step(None)
        while not done:
            old_obs = new_obs
            action, new_obs, reward, done, _ = env.step_offline(old_obs, reward, done)
            agent.train(old_obs, action, reward, False)
        old_obs = new_obs
        action, _, reward, done, _ = env.step_offline(old_obs, reward, done)
        agent.train(old_obs, action, reward, True)


# the reference's offline-log cache (bench_agents.py:17-63): one pickle of generate_logs' DataFrame per (env config, users)
CACHE_DIR = os.path.join(os.path.join(str(Path.home()), '.reco-gym'), 'cache')


def _cache_file_name(env, num_organic_offline_users, num_offline_users):
    """The reference's key: a hash over the env configuration's fields and the two user counts (the epoch is not part of it
    there either: with the cache on, every epoch trains on the same log)."""
    c = env.config
    fields = ('K', 'change_omega_for_bandits', 'normalize_beta', 'num_clusters', 'num_products', 'num_users', 'number_of_flips',
              'phi_var', 'prob_bandit_to_organic', 'prob_leave_bandit', 'prob_leave_organic', 'prob_organic_to_bandit',
              'random_seed', 'sigma_mu_organic', 'sigma_omega', 'sigma_omega_initial', 'with_ps_all')
    key = (tuple([str(type(getattr(c, 'agent', None)))] + [getattr(c, f, None) for f in fields]),
           num_organic_offline_users, num_offline_users)
    return f'{hashlib.sha1(json.dumps(key, default=str).encode()).hexdigest()}.pkl'


def _cached_data(env, num_organic_offline_users, num_offline_users):
    cache_dir = os.environ.get('RECOGYM_CACHE_DIR', CACHE_DIR)
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, _cache_file_name(env, num_organic_offline_users, num_offline_users))
    if os.path.exists(path):
        try:
            with open(path, 'rb') as fh:
                return pickle.load(fh, fix_imports=False)
        except Exception:       # noqa: BLE001 — a truncated / foreign file (a run killed mid-write): regenerate it
            pass
    data = env.generate_logs(num_offline_users=num_offline_users, num_organic_offline_users=num_organic_offline_users)
    # written beside its final name and renamed into place: a reader (another rank, another process) sees the old file, no
    # file, or the whole new one — never half of it
    tmp = f'{path}.{os.getpid()}.tmp'
    with open(tmp, 'wb') as fh:
        pickle.dump(data, fh, protocol=pickle.HIGHEST_PROTOCOL, fix_imports=False)
    os.replace(tmp, path)
    return data


def _train_from_dataframe(agent, data):
    """agent.train over a cached log, the way the offline protocol would have called it live: per user, every bandit row with
    the organic session that preceded it; the user's last row closes the episode (done = True), an organic-only user is one
    call without an action (bench_agents.py:90-190)."""
    import numpy as np
    from .envs.context import DefaultContext
    from .envs.observation import Observation
    from .envs.session import OrganicSessions
    t, u, z = data['t'].to_numpy(), data['u'].to_numpy(dtype=np.int64), (data['z'] == 'bandit').to_numpy()
    v, a, c = data['v'].to_numpy(dtype=float, na_value=np.nan), data['a'].to_numpy(dtype=float, na_value=np.nan), data['c'].to_numpy(dtype=float)
    ps = data['ps'].to_numpy(dtype=float)
    ps_a = data['ps-a'].to_list() if 'ps-a' in data else [()] * len(data)
    n = len(data)
    last_of_user = np.r_[u[1:] != u[:-1], True] if n else np.zeros(0, dtype=bool)
    sessions = OrganicSessions()
    for i in range(n):
        ctx = DefaultContext(t[i], int(u[i]))
        if not z[i]:
            sessions.next(ctx, int(v[i]))
            if last_of_user[i]:                       # an organic-only (warm-up) user
                agent.train(Observation(ctx, sessions), None, None, True)
                sessions = OrganicSessions()
            continue
        action = {'t': t[i], 'u': int(u[i]), 'a': int(a[i]), 'ps': float(ps[i]), 'ps-a': ps_a[i] if ps_a[i] is not None else ()}
        agent.train(Observation(ctx, sessions), action, int(c[i]), bool(last_of_user[i]))
        sessions = OrganicSessions()


def test_agent(env, agent, num_offline_users=1000, num_online_users=100,
               num_organic_offline_users=0, num_epochs=1, epoch_with_random_reset=False,
               with_cache=False):
    # The reference draws every user from one sequential stream: the online (evaluation) users are fresh draws that
    # follow the offline (training) users, and without a random reset every epoch continues that stream.  With
    # addressed draws (seed, user id, t) the same is obtained by user-id ranges: an epoch takes the ids
    # [base, base + organic + offline) for training and the next `num_online_users` ids for evaluation; without
    # epoch_with_random_reset the next epoch starts where this one ended (with it, the epoch re-keys the draws
    # and ids restart at 0, as the reference's reset_random_seed(epoch) restarts its stream).
    # with_cache (bench_agents.py:50-63,90-166): the offline log is generate_logs' DataFrame, pickled under ~/.reco-gym/cache
    # (RECOGYM_CACHE_DIR overrides) by a hash of the env configuration and the user counts, and the agent is trained from it
    successes = failures = 0
    per_epoch = num_organic_offline_users + num_offline_users + num_online_users
    for epoch in range(num_epochs):
        base = 0 if epoch_with_random_reset else epoch * per_epoch
        eval_first = base + num_organic_offline_users + num_offline_users
        new_agent = deepcopy(agent)
        if epoch_with_random_reset:
            train_env = deepcopy(env)
            train_env.reset_random_seed(epoch)
            eval_env = deepcopy(env)
            eval_env.reset_random_seed(epoch)
        else:
            train_env = eval_env = env
        if with_cache and (_learns(new_agent) or hasattr(new_agent, 'train_from_log')):
            data = _cached_data(train_env, num_organic_offline_users, num_offline_users)
            if hasattr(new_agent, 'train_from_log'):
                new_agent.train_from_log(data, num_organic_offline_users)
            else:
                _train_from_dataframe(new_agent, data)
        elif hasattr(new_agent, 'train_from_log') and getattr(train_env, 'agent', None) is None:
            # the offline protocol shows the agent exactly the rows of generate_logs(offline users) under the
            # env's own uniform policy (bench_agents.py:168-190): produce that log on the device in one go
            cnt, sim = train_env.simulate(num_offline_users, None, num_organic_offline_users, first_user_id=base)
            new_agent.train_from_log(sim.log_columns(), num_organic_offline_users)
            sim.close()
        elif _learns(new_agent) and (getattr(new_agent, 'needs_training', False) or device_policy_of(new_agent) is None):
            _train(train_env, new_agent, num_offline_users, num_organic_offline_users, first_user_id=base)
        s, f = evaluate_counts(eval_env, new_agent, num_online_users, first_user_id=eval_first)
        successes += s
        failures += f
    return (beta.ppf(0.500, successes + 1, failures + 1),
            beta.ppf(0.025, successes + 1, failures + 1),
            beta.ppf(0.975, successes + 1, failures + 1))


test_agent.__test__ = False      # not a pytest test
