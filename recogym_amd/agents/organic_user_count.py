"""OrganicUserEventCounterAgent — reference: recogym/agents/organic_user_count.py:45-96 acting on
the per-user view counts of ViewsFeaturesProvider (recogym/agents/abstract.py:347-395).

On the device the counts live as a sorted (product, count) history per user (rg_config.ouc_*);
this class is the policy descriptor plus the host form of the same arithmetic.
"""
import numpy as np

from .. import _abi, rng
from ..envs.configuration import Configuration
from .abstract import Agent

organic_user_count_args = {
    'num_products': 10,
    'random_seed': np.random.randint(2 ** 31 - 1),
    'select_randomly': True,      # sample proportionally to the counts (else argmax)
    'epsilon': .0,
    'exploit_explore': True,
    'reverse_pop': False,
    'weight_history_function': None,
    'with_ps_all': False,
}


def _icdf(p, u):
    """RandomState.choice(P, p=p) given its uniform (choice works on a float64 copy of p)."""
    cdf = np.asarray(p, dtype=np.float64).cumsum()
    cdf /= cdf[-1]
    return int(cdf.searchsorted(u, side='right'))


class OrganicUserEventCounterAgent(Agent):
    def __init__(self, config=Configuration(organic_user_count_args)):
        super().__init__(config)
        # weight_history_function (organic_user_count.py:19, agents/abstract.py:343-382): time-weighted views instead of
        # counts — float features per event: the per-user host path (no device policy)
        self.history = None
        if getattr(config, 'weight_history_function', None) is not None:
            from .views_history import ViewsHistory
            self.history = ViewsHistory(config.num_products, config.weight_history_function)
        self.views = np.zeros(config.num_products, dtype=np.int64)

    def device_policy(self):
        c = self.config
        if self.history is not None:
            return None
        pol = dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=c.random_seed,
                   ouc=dict(select_randomly=c.select_randomly, epsilon=c.epsilon,
                            exploit_explore=c.exploit_explore,
                            reverse_pop=getattr(c, 'reverse_pop', False)))
        if getattr(c, 'with_ps_all', False):
            # the device logs (a, ps); the whole distribution of every bandit row (`ps-a`, organic_user_count.py:77-94)
            # is a function of the user's views so far, which the log itself holds: rebuilt at materialisation
            pol['ps_all'] = self.ps_all_from_log
        return pol

    def ps_all_from_log(self, df):
        """`ps-a` of every bandit row of a log this agent produced (with_ps_all=True): the distribution act() sampled
        from, recomputed from the organic rows that precede the row (same arithmetic as act(); the explore flip is
        the addressed policy draw of (user, t)).  O(bandit rows x P) on the host, like the reference's own column."""
        c = self.config
        P = c.num_products
        out = np.empty(len(df), dtype=object)
        views = np.zeros(P, dtype=np.int64)
        cur = None
        u_col = df['u'].to_numpy(dtype=np.int64)
        is_b = (df['z'] == 'bandit').to_numpy()
        v_col = df['v'].to_numpy(dtype=np.float64, na_value=np.nan)
        a_col = df['a'].to_numpy(dtype=np.float64, na_value=np.nan)
        eps = c.epsilon
        idx = 0
        for i in range(len(df)):
            if u_col[i] != cur:
                cur = u_col[i]
                views[:] = 0
                idx = 0
            # the policy draw is keyed by the user's EVENT INDEX (= the row's position within the user; the phantom row
            # takes the index after the last real event), not by the `t` column: with a NormalTimeGenerator `t` is the
            # float clock, and the device / act() key the draw by the index (DefaultContext.draw_key)
            ev = idx
            idx += 1
            if not is_b[i]:
                views[int(v_col[i])] += 1
                out[i] = None
                continue
            f = views.astype(np.float64)
            if c.exploit_explore:
                _, u0, _ = rng.policy_uniforms(c.random_seed, int(u_col[i]), ev)
                if not (eps / (eps + (1 - eps)) <= u0):
                    f = (views == 0).astype(np.float64)
                p = f / np.sum(f)
            else:
                f = eps + f
                p = f / np.sum(f)
                if getattr(c, 'reverse_pop', False):
                    p = 1 - p
                    p = p / np.sum(p)
            if c.select_randomly:
                out[i] = p
            else:
                one = np.zeros(P)
                one[int(a_col[i])] = 1.0
                out[i] = one
        return out

    def reset(self):
        self.views[:] = 0
        if self.history is not None:
            self.history.reset()

    def act(self, observation, reward, done):
        c = self.config
        for s in observation.sessions():
            self.views[int(s['v'])] += 1
        ctx = observation.context()
        _, u0, u1 = rng.policy_uniforms(c.random_seed, *(ctx.draw_key() if hasattr(ctx, 'draw_key') else (ctx.user(), ctx.time())))
        eps = c.epsilon
        if self.history is not None:
            # float32 weighted views, exactly as the reference's feature provider builds them; the arithmetic below then runs
            # in float32 where the reference's does (features / sum, eps * p[a])
            self.history.observe(observation)
            f = self.history.features(ctx.time()).flatten()
        else:
            f = self.views.astype(np.float64)
        explore = False
        if c.exploit_explore:
            explore = not (eps / (eps + (1 - eps)) <= u0)
            if explore:
                f = (f == 0).astype(f.dtype)
            p = f / np.sum(f)
        else:
            f = eps + f
            p = f / np.sum(f)
            if getattr(c, 'reverse_pop', False):
                p = 1 - p
                p = p / np.sum(p)
        if c.select_randomly:
            a = _icdf(p, u1)
            ps = ((eps if explore else 1 - eps) * p[a]) if c.exploit_explore else p[a]
            ps_all = p if getattr(c, 'with_ps_all', False) else ()
        else:
            a = int(np.argmax(p))
            ps = 1.0
            ps_all = ()
            if getattr(c, 'with_ps_all', False):
                ps_all = np.zeros(c.num_products)
                ps_all[a] = 1.0
        return {**super().act(observation, reward, done), 'a': a, 'ps': ps, 'ps-a': ps_all}
