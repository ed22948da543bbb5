"""Runs the UNMODIFIED reference (/root/reference, read-only) in this container to pin the
oracle: (a) on its native sequential MT19937 stream, (b) with the counter RNG of
include/recogym_rng.h injected through the duck-typed `env.rng`.

/root/reference does not exist on the GPU box: this module is imported only by
tests/make_golden.py (the committed generator of tests/golden/*) and by tests that skip when the
reference is absent.  Nothing under recogym_amd/ imports it.
"""
import os
import sys

import numpy as np

REFERENCE_ROOT = '/root/reference'
SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_shims')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'recogym'))


def import_reference():
    """import recogym (the reference) with the gym/numba shims and two numpy/scipy aliases."""
    import scipy
    if not hasattr(scipy, 'rand'):
        scipy.rand = np.random.rand           # bayesian_poly_vb.py:24
    if not hasattr(np, 'float'):
        np.float = float                      # agents/abstract.py:274
    for p in (SHIMS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import recogym
    import recogym.envs.abstract as ab
    ab.trange = lambda n, **k: range(n)       # silence tqdm
    return recogym


# ------------------------------------------------------------------------------------------
# numpy restatement of include/recogym_rng.h (python ints: exact)
# ------------------------------------------------------------------------------------------
M32 = 0xFFFFFFFF
DRAW_EVENT, DRAW_POLICY, DRAW_DRIFT, DRAW_RESET, DRAW_TIME = 0, 1, 2, 3, 4


def philox4x32_10(c, k):
    c0, c1, c2, c3 = [int(x) & M32 for x in c]
    k0, k1 = [int(x) & M32 for x in k]
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def draw(seed, user, t, slot, purpose):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return philox4x32_10((user, t, slot, purpose), (seed & M32, seed >> 32))


def m53(a, b):
    return ((a >> 5) << 26) | (b >> 6)


def uniform(a, b):
    return m53(a, b) / 9007199254740992.0


def bounded(a, b, n):
    return (m53(a, b) * n) >> 53


def normals(seed, user, t, purpose, K):
    z = np.empty(K)
    for j in range((K + 1) // 2):
        w = draw(seed, user, t, j, purpose)
        u1, u2 = uniform(w[0], w[1]), uniform(w[2], w[3])
        r = np.sqrt(-2.0 * np.log(1.0 - u1))
        th = 6.283185307179586476925286766559 * u2
        z[2 * j] = r * np.cos(th)
        if 2 * j + 1 < K:
            z[2 * j + 1] = r * np.sin(th)
    return z


def numpy_choice_with_p(a, p, u):
    """RandomState.choice(a, p=p) given its uniform: cdf = cumsum(p); cdf /= cdf[-1];
    searchsorted(cdf, u, 'right') — numpy/random/mtrand.pyx (legacy choice)."""
    p = np.asarray(p, dtype=np.float64)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    idx = int(cdf.searchsorted(u, side='right'))
    if isinstance(a, (int, np.integer)):
        return idx
    return a[idx]


class InjectedEnvRng:
    """Duck-typed `env.rng` (exposes .normal and .choice) that serves the counter-RNG draws
    addressed by (seed, user, t, purpose) to the unmodified reference env.

    The call pattern of the reference identifies each draw:
      normal(0, s0, size=(K,1))         reset               -> RG_DRAW_RESET   (reco_env_v1.py:80)
      choice(P, p=...)                  organic product     -> EVENT words 0,1 (reco_env_v1.py:124)
      choice([0, 1], p=...)             click               -> EVENT words 0,1 (reco_env_v1.py:112)
      choice(3, p=...)                  Markov transition   -> EVENT words 2,3 (reco_env_v1.py:87)
      normal(omega, s, size=(K,1))      drift after event t -> RG_DRAW_DRIFT   (reco_env_v1.py:96)
      choice(P)                         agent=None action   -> POLICY words 0,1 (abstract.py:214)
    """

    def __init__(self, env, seed, policy_seed=None):
        self.env = env
        self.seed = seed
        self.policy_seed = seed if policy_seed is None else policy_seed
        self.t = 0
        self.user = 0
        self.p_click_log = []

    def start_user(self, user_id):
        self.user = user_id
        self.t = 0
        self._reset_pending = True

    def normal(self, loc=0.0, scale=1.0, size=None):
        K = self.env.config.K
        assert size == (K, 1)
        if self._reset_pending:
            self._reset_pending = False
            z = normals(self.seed, self.user, 0, DRAW_RESET, K)
        else:
            # drift that follows the transition of event t-1 (t was already advanced)
            z = normals(self.seed, self.user, self.t - 1, DRAW_DRIFT, K)
        return loc + scale * z.reshape(K, 1)

    def binomial(self, n, p):
        """reco-gym-v0's click (reco_env_v0.py:61-63): RandomState.binomial(1, p), i.e. numpy's legacy inversion
        (numpy/random/src/legacy/legacy-distributions.c) fed with the EVENT draw's words 0,1 (restarts: slots 1, 2, ...)."""
        import math
        assert n == 1
        self.p_click_log.append(float(p))
        pe = p if p <= 0.5 else 1.0 - p
        q = 1.0 - pe
        qn = math.exp(n * math.log(q))
        np_ = n * pe
        bound = int(min(n, np_ + 10.0 * math.sqrt(np_ * q + 1)))
        slot = 0
        w = draw(self.seed, self.user, self.t, slot, DRAW_EVENT)
        U = uniform(w[0], w[1])
        X, px = 0, qn
        while U > px:
            X += 1
            if X > bound:
                X, px = 0, qn
                slot += 1
                w = draw(self.seed, self.user, self.t, slot, DRAW_EVENT)
                U = uniform(w[0], w[1])
            else:
                U -= px
                px = ((n - X + 1) * pe * px) / (X * q)
        return X if p <= 0.5 else n - X

    def choice(self, a, p=None):
        if p is None:
            w = draw(self.policy_seed, self.user, self.t, 0, DRAW_POLICY)
            return bounded(w[0], w[1], int(a))
        if self._reset_pending and type(self.env).__name__ == 'RecoEnv0':
            # reco-gym-v0's reset draws the first product view: choice(P, p = initial_product_probs), reco_env_v0.py:52-54
            self._reset_pending = False
            w = draw(self.seed, self.user, 0, 0, DRAW_RESET)
            return numpy_choice_with_p(a, p, uniform(w[0], w[1]))
        w = draw(self.seed, self.user, self.t, 0, DRAW_EVENT)
        if isinstance(a, (int, np.integer)) and a == 3 and len(p) == 3:
            # NB: P == 3 would be ambiguous with the transition draw; fixtures avoid P == 3
            out = numpy_choice_with_p(a, p, uniform(w[2], w[3]))
            self.t += 1
            return out
        if not isinstance(a, (int, np.integer)):
            self.p_click_log.append(float(p[1]))
        return numpy_choice_with_p(a, p, uniform(w[0], w[1]))


class InjectedTimeRng:
    """Duck-typed `env.time_generator.rng` of the reference's NormalTimeGenerator (normal_time_generator.py:21,25):
    its k-th `normal(mu, sigma)` call after a reset serves the RG_DRAW_TIME draw of (user, k) — call 0 is made by
    reset() (the increment that follows event 0), call k + 1 by the update_state of event k."""

    def __init__(self, env_rng):
        self.env_rng = env_rng
        self.calls = 0

    def normal(self, loc=0.0, scale=1.0, size=None):
        assert size is None
        z = normals(self.env_rng.seed, self.env_rng.user, self.calls, DRAW_TIME, 1)[0]
        self.calls += 1
        return loc + scale * z


class InjectedAgentRng:
    """Duck-typed `agent.rng` / `agent.model.rng` serving RG_DRAW_POLICY draws.

      RandomAgent:  choice(P)                          -> POLICY words 0,1 (random_agent.py:26)
      OrganicUserEventCounter: choice([True, False], p) -> words 0,1 (organic_user_count.py:48)
                               choice(P, p=...)         -> words 2,3 (organic_user_count.py:66)
    (user, t) are read from the env's injected rng, which tracks them.
    """

    def __init__(self, env_rng, policy_seed):
        self.env_rng = env_rng
        self.policy_seed = policy_seed

    def choice(self, a, p=None):
        w = draw(self.policy_seed, self.env_rng.user, self.env_rng.t, 0, DRAW_POLICY)
        if p is None:
            return bounded(w[0], w[1], int(a))
        if not isinstance(a, (int, np.integer)):
            return numpy_choice_with_p(a, p, uniform(w[0], w[1]))
        return numpy_choice_with_p(a, p, uniform(w[2], w[3]))


def make_reference_env(args, env_id='reco-gym-v1'):
    recogym = import_reference()
    import gym
    env = gym.make(env_id)
    env.init_gym(args)
    return env


def inject_counter_rng(env, agent=None, agent_seed=None):
    """Swap the env's (and optionally the agent's) sequential RandomState for the counter RNG.
    No reference file is edited: `env.rng` is a plain attribute and `reset` is wrapped."""
    rng = InjectedEnvRng(env, env.config.random_seed,
                         policy_seed=env.config.random_seed)
    env.rng = rng
    original_reset = env.reset
    trng = None
    if type(env.time_generator).__name__ == 'NormalTimeGenerator':
        trng = InjectedTimeRng(rng)
        env.time_generator.rng = trng

    def reset(user_id=0):
        rng.start_user(user_id)
        if trng is not None:
            trng.calls = 0
        original_reset(user_id)
    env.reset = reset
    if agent is not None:
        arng = InjectedAgentRng(rng, agent_seed)
        if hasattr(agent, 'model_builder'):
            # ModelBasedAgent builds its model (and the model its RandomState) at first act
            original_build = agent.model_builder.build

            def build():
                fp, model = original_build()
                model.rng = arng
                return fp, model
            agent.model_builder.build = build
            if getattr(agent, 'model', None) is not None:
                agent.model.rng = arng
        else:
            agent.rng = arng
    return rng


def log_to_arrays(df, p_click=None):
    """Reference DataFrame -> plain numpy columns (NA -> -1 / NaN) for the fixtures."""
    n = len(df)
    z = (df['z'].values == 'bandit').astype(np.int32)
    out = {
        't': df['t'].values.astype(np.int64),
        'u': df['u'].astype('Int64').fillna(-1).values.astype(np.int64),
        'z': z,
        'v': df['v'].astype('Int64').fillna(-1).values.astype(np.int64),
        'a': df['a'].astype('Int64').fillna(-1).values.astype(np.int64),
        'c': np.where(np.isnan(df['c'].values), -1, df['c'].values).astype(np.int64),
        'ps': np.array([np.nan if x is None else float(x) for x in df['ps'].values],
                       dtype=np.float64),
    }
    assert len(out['t']) == n
    if p_click is not None:
        out['p_click'] = np.asarray(p_click, dtype=np.float64)
    return out
