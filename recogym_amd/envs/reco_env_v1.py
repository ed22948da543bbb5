"""RecoEnv1 — the reference's `reco-gym-v1` environment surface over the HIP step loop.

Reference surface mirrored here (same names, arguments, return values and assertion errors):
    AbstractEnv   recogym/envs/abstract.py:46-327   init_gym, reset_random_seed, reset, step,
                                                    step_offline, generate_logs
    RecoEnv1      recogym/envs/reco_env_v1.py:44-174
    env_args / env_1_args                           abstract.py:20-31, reco_env_v1.py:18-29

What is different, on purpose (DESIGN.md §"drop-in boundary"):
  * every draw is addressed by (random_seed + epoch, user id, t) instead of coming from one
    sequential MT19937 stream, so a user's trajectory does not depend on the users simulated
    before it and `generate_logs` is repeatable without deepcopy;
  * `generate_logs` with `agent=None`, a RandomAgent or an OrganicUserEventCounterAgent runs all
    users concurrently on the GPU; any other agent goes through the per-user path
    (`reset`/`step_offline`, still HIP kernels, batch of one user);
  * user / product ids wider than 16 bits are returned as UInt32 columns (the reference
    overflows, SURVEY.md "facts");
  * there is no CPU fallback: without a GPU, constructing the simulator raises.
"""
from copy import deepcopy

import numpy as np
import pandas as pd
import torch

from .. import _abi, rng
from ..sim import Simulator, decode_rows
from .configuration import Configuration
from .context import DefaultContext
from .features.time import DefaultTimeGenerator, NormalTimeGenerator
from .static_params import time_generator_params
from .observation import Observation
from .session import OrganicSessions
from .static_params import draw_tables

# Markov states — abstract.py:41-43
organic = 0
bandit = 1
stop = 2

# Arguments shared between all environments — abstract.py:20-31
env_args = {
    'num_products': 10,
    'num_users': 100,
    'random_seed': np.random.randint(2 ** 31 - 1),
    'prob_leave_bandit': 0.01,
    'prob_leave_organic': 0.01,
    'prob_bandit_to_organic': 0.05,
    'prob_organic_to_bandit': 0.25,
    'normalize_beta': False,
    'with_ps_all': False,
}

# reco_env_v1.py:18-29 (num_clusters / phi_var leak in from env_0_args in the reference and are
# read by bench_agents' cache key, so they are kept)
env_1_args = {
    **env_args,
    'num_clusters': 2,
    'phi_var': 0.1,
    'K': 5,
    'sigma_omega_initial': 1,
    'sigma_omega': 0.1,
    'number_of_flips': 0,
    'sigma_mu_organic': 3,
    'change_omega_for_bandits': False,
    'normalize_beta': False,
}


class _DiscreteStandIn:
    """gym.spaces.Discrete stand-in (only `.n` is read) where no gym is importable."""

    def __init__(self, n):
        self.n = n


def _gym_surface():
    """(base class, Discrete) — `gym.Env` / `gym.spaces.Discrete` (or gymnasium's) where such a package is importable, the way
    the reference's AbstractEnv derives from gym.Env (recogym/envs/abstract.py:46-57); else plain object and a stand-in."""
    for name in ('gym', 'gymnasium'):
        try:
            mod = __import__(name)
            spaces = __import__(name + '.spaces', fromlist=['Discrete'])
            if isinstance(getattr(mod, 'Env', None), type) and hasattr(spaces, 'Discrete'):
                return mod.Env, spaces.Discrete
        except Exception:       # noqa: BLE001 — not installed, or a stub without the Env / spaces surface
            continue
    return object, _DiscreteStandIn


_EnvBase, Discrete = _gym_surface()


def device_policy_of(agent):
    """-> dict(policy, policy_seed, ouc) if `agent` can run inside the batched device loop."""
    if agent is None:
        return dict(policy=_abi.RG_POLICY_UNIFORM_ENV, policy_seed=None, ouc=None)
    if hasattr(agent, 'device_policy'):
        return agent.device_policy()
    # duck-type the reference's own agent classes
    name = type(agent).__name__
    cfg = getattr(agent, 'config', None)
    if cfg is None or getattr(cfg, 'with_ps_all', False):
        return None
    if name == 'RandomAgent':
        return dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=cfg.random_seed, ouc=None)
    if name == 'OrganicUserEventCounterAgent' and \
            getattr(cfg, 'weight_history_function', None) is None and not getattr(cfg, 'with_ps_all', False):
        return dict(policy=_abi.RG_POLICY_ORGANIC_USER_COUNT, policy_seed=cfg.random_seed,
                    ouc=dict(select_randomly=cfg.select_randomly, epsilon=cfg.epsilon,
                             exploit_explore=cfg.exploit_explore,
                             reverse_pop=getattr(cfg, 'reverse_pop', False)))
    if name == 'LogregMulticlassIpsAgent' \
            and getattr(agent, 'model', None) is not None and hasattr(agent.model, 'logreg'):
        # the reference's own agent object with a built model: run its fitted arrays on the device (select_randomly: sampled
        # there too where the model has a class per product and P <= 1024 — LogregFrozenAgent.device_policy decides)
        from ..agents.logreg_frozen import LogregFrozenAgent
        return LogregFrozenAgent.from_sklearn(cfg, agent.model.logreg).device_policy()
    return None


def rows_to_dataframe(rows, num_products, with_ps_all=False):
    """Decoded device rows -> the DataFrame of generate_logs (abstract.py:256-265,318-327):
    columns t (float32), u/v/a (nullable UInt16, UInt32 beyond the reference's ceiling),
    z ('organic'/'bandit'), c (float32, NaN on organic rows), ps (float64), ps-a (object).
    Vectorised: masked integer arrays are built from (values, mask) directly and the two object
    columns by fancy-indexing a 2-entry lookup (no per-row Python, ~10 M rows/s per core)."""
    is_b = np.asarray(rows['z'] == 1)
    n = len(rows)
    wide_u = n and int(rows['u'].max()) > 65535
    wide_p = num_products > 65535
    u_np, udt = (np.uint32, pd.UInt32Dtype()) if wide_u else (np.uint16, pd.UInt16Dtype())
    p_np, pdt = (np.uint32, pd.UInt32Dtype()) if wide_p else (np.uint16, pd.UInt16Dtype())
    sel = is_b.astype(np.intp)
    lut = np.empty(2, dtype=object)
    lut[0] = None
    lut[1] = (np.ones(num_products) / num_products) if with_ps_all else ()
    z_lut = np.array(['organic', 'bandit'], dtype=object)
    not_b = ~is_b
    IntegerArray = pd.arrays.IntegerArray
    data = {
        't': rows['t'].astype(np.float32),
        'u': IntegerArray(rows['u'].astype(u_np), np.zeros(n, dtype=bool)),
        'z': z_lut[sel],
        'v': IntegerArray(np.where(is_b, 0, rows['v']).astype(p_np), is_b.copy()),
        'a': IntegerArray(np.where(is_b, rows['a'], 0).astype(p_np), not_b),
        'c': np.where(is_b, rows['c'], np.nan).astype(np.float32),
        'ps': np.where(is_b, rows['ps'], np.nan).astype(np.float64),
        'ps-a': lut[sel],
    }
    del udt, pdt
    # Series(copy=False) per column: the dict constructor would re-infer / copy the object columns
    cols = {k: pd.Series(v, dtype=object if k in ('z', 'ps-a') else None, copy=False) for k, v in data.items()}
    return pd.DataFrame(cols, columns=['t', 'u', 'z', 'v', 'a', 'c', 'ps', 'ps-a'], copy=False)


def raw_log_to_dataframe(raw, num_products, uniform_ps=None, with_ps_all=False):
    """Sorted device log ((n, 4) int32 `rg_event` records, host memory) -> the DataFrame of
    generate_logs, column by column (the log materialisation step of SURVEY.md §8f-2).  Same result
    as rows_to_dataframe(decode_rows(raw)), without the structured intermediate."""
    from .. import _abi
    raw = np.ascontiguousarray(raw).view(np.uint32).reshape(-1, 4)
    n = raw.shape[0]
    u = raw[:, 0].copy()
    code = raw[:, 2].copy()
    is_b = (code & np.uint32(_abi.RG_EV_BANDIT)) != 0
    not_b = ~is_b
    idx = code & np.uint32(_abi.RG_EV_INDEX_MASK)
    wide_u = n and int(u.max()) > 65535
    wide_p = num_products > 65535
    u_np = np.uint32 if wide_u else np.uint16
    p_np = np.uint32 if wide_p else np.uint16
    ps = np.full(n, np.nan)
    if uniform_ps is not None:
        ps[is_b] = uniform_ps                                   # exact 1/P of the uniform policies
    else:
        ps[is_b] = raw[:, 3].copy().view(np.float32)[is_b]
    c = np.full(n, np.nan, dtype=np.float32)
    c[is_b] = ((code & np.uint32(_abi.RG_EV_CLICK)) != 0)[is_b]
    sel = is_b.astype(np.intp)
    lut = np.empty(2, dtype=object)
    lut[0] = None
    lut[1] = (np.ones(num_products) / num_products) if with_ps_all else ()
    IntegerArray = pd.arrays.IntegerArray
    v = idx.astype(p_np)
    a = v.copy()
    v[is_b] = 0
    a[not_b] = 0
    data = {
        't': raw[:, 1].astype(np.float32),
        'u': IntegerArray(u.astype(u_np), np.zeros(n, dtype=bool)),
        'z': np.array(['organic', 'bandit'], dtype=object)[sel],
        'v': IntegerArray(v, is_b),
        'a': IntegerArray(a, not_b),
        'c': c,
        'ps': ps,
        'ps-a': lut[sel],
    }
    cols = {k: pd.Series(x, dtype=object if k in ('z', 'ps-a') else None, copy=False) for k, x in data.items()}
    return pd.DataFrame(cols, columns=['t', 'u', 'z', 'v', 'a', 'c', 'ps', 'ps-a'], copy=False)


def columns_to_dataframe(cols, num_products, with_ps_all=False):
    """Simulator.log_columns() -> the DataFrame of generate_logs: only the wrapping is left to the
    host (nullable integer arrays from (values, mask), the two object columns from a 2-entry
    lookup), everything row-wise was done on the device."""
    is_b = cols['is_bandit']
    n = len(is_b)
    # user ids above the reference's UInt16 ceiling arrive as wrapped int32: view as unsigned first
    u = cols['u'].view(np.uint32)
    wide_u = n and int(u.max()) > 65535
    wide_p = num_products > 65535
    u_np = np.uint32 if wide_u else np.uint16
    p_np = np.uint32 if wide_p else np.uint16
    sel = is_b.astype(np.intp)
    lut = np.empty(2, dtype=object)
    lut[0] = None
    lut[1] = (np.ones(num_products) / num_products) if with_ps_all else ()
    IntegerArray = pd.arrays.IntegerArray
    data = {
        't': cols['t'],
        'u': IntegerArray(u.astype(u_np), np.zeros(n, dtype=bool)),
        'z': np.array(['organic', 'bandit'], dtype=object)[sel],
        'v': IntegerArray(cols['v'].astype(p_np), is_b),
        'a': IntegerArray(cols['a'].astype(p_np), ~is_b),
        'c': cols['c'],
        'ps': cols['ps'],
        'ps-a': lut[sel],
    }
    out = {k: pd.Series(x, dtype=object if k in ('z', 'ps-a') else None, copy=False) for k, x in data.items()}
    return pd.DataFrame(out, columns=['t', 'u', 'z', 'v', 'a', 'c', 'ps', 'ps-a'], copy=False)


_RNG_TYPES = None


def _rng_types():
    global _RNG_TYPES
    if _RNG_TYPES is None:
        import random as _random
        t = [np.random.RandomState, np.random.Generator, np.random.BitGenerator, _random.Random]
        try:
            t.append(torch.Generator)
        except Exception:
            pass
        _RNG_TYPES = tuple(t)
    return _RNG_TYPES


def _walk_state(obj, depth=6, seen=None):
    """The objects reachable from `obj` through instance attributes, lists, tuples and dict values (bounded depth)."""
    seen = set() if seen is None else seen
    if id(obj) in seen or depth < 0:
        return
    seen.add(id(obj))
    yield obj
    if isinstance(obj, (str, bytes, int, float, bool, type(None), np.ndarray, torch.Tensor)):
        return
    kids = []
    if isinstance(obj, dict):
        kids = list(obj.values())
    elif isinstance(obj, (list, tuple, set, frozenset)):
        kids = list(obj)
    elif hasattr(obj, '__dict__'):
        kids = list(vars(obj).values())
    for k in kids:
        yield from _walk_state(k, depth - 1, seen)


def batch_safe(agent):
    """May `generate_logs` advance this agent B users at a time with a copy of it per user slot (_generate_logs_batched)?
    The copies of an agent that owns a sequential random stream (EpsilonGreedy, NnIpsAgent, RandomAgent with its RandomState —
    `random_agent.py:14-20`, `epsilon_greedy.py`) would all draw the SAME sequence: such agents keep the reference's one-user-
    at-a-time loop, whose single stream is consumed sequentially as in `abstract.py:292-316`.  `agent.batch_safe = True / False`
    overrides the inspection (an agent whose draws are addressed by user, or that does not care)."""
    flag = getattr(agent, 'batch_safe', None)
    if flag is not None:
        return bool(flag)
    rng = _rng_types()
    for o in _walk_state(agent):
        if isinstance(o, rng):
            return False
    return True


def _shared_state_memo(agent):
    """deepcopy memo that maps the agent's large read-only model state onto itself (shared by the per-slot copies)."""
    memo = {}
    for o in _walk_state(agent):
        big_array = isinstance(o, np.ndarray) and o.nbytes >= (64 << 10)
        big_tensor = isinstance(o, torch.Tensor) and o.numel() * o.element_size() >= (64 << 10)
        model = isinstance(o, torch.nn.Module) or (hasattr(o, 'predict') and hasattr(o, 'get_params'))     # torch / sklearn estimators
        if big_array or big_tensor or model:
            memo[id(o)] = o
    return memo


def _approx_owned_bytes(agent, shared):
    n = 256
    for o in _walk_state(agent):
        if id(o) in shared:
            continue
        if isinstance(o, np.ndarray):
            n += o.nbytes
        elif isinstance(o, torch.Tensor):
            n += o.numel() * o.element_size()
        else:
            n += 64
    return n


class RecoEnv1(_EnvBase):
    """Drop-in for the object `gym.make('reco-gym-v1')` returns: a `gym.Env` subclass wherever gym is importable (like the
    reference's AbstractEnv, abstract.py:46-57), with `action_space = Discrete(num_products)` after init_gym (abstract.py:69)
    and — what gym's checkers ask of an Env — an `observation_space`: Discrete(num_products), the product ids an observation's
    organic sessions carry."""

    metadata = {}

    def __init__(self):
        if _EnvBase is not object:
            _EnvBase.__init__(self)
        self.first_step = True
        self.config = None
        self.state = None
        self.current_user_id = None
        self.current_time = None
        self.empty_sessions = OrganicSessions()
        self.agent = None
        self._epoch = 0
        self._tables = None
        self._seq = None          # 1-user simulator behind reset()/step()
        self._device = None
        self._time_mode = 0       # 1: NormalTimeGenerator (the device keeps the clocks)
        self._event_index = 0

    # -- construction -------------------------------------------------------------------------
    def init_gym(self, args):
        self.config = Configuration(args)
        self.action_space = Discrete(self.config.num_products)
        self.observation_space = Discrete(self.config.num_products)
        # abstract.py:72-76: the default generator unless one is passed in.  A NormalTimeGenerator (this package's or the
        # reference's own object) only contributes mu / sigma: the device keeps the per-user clocks
        self._time_mode = time_generator_params(self.config)[0]
        self.time_generator = args['time_generator'] if 'time_generator' in args else DefaultTimeGenerator(self.config)
        self.agent = args['agent'] if 'agent' in args else None
        self.reset_random_seed()
        self._tables = draw_tables(self.config)      # set_static_params, bit-identical draws

    def reset_random_seed(self, epoch=0):
        assert (self.config.random_seed is not None)
        self._epoch = epoch
        if self._seq is not None:
            self._seq.reseed(self.config.random_seed + epoch, self.config.random_seed + epoch)

    @property
    def seed(self):
        return self.config.random_seed + self._epoch

    # lazily materialised numpy views of the static tables, for debugging / notebooks
    @property
    def Gamma(self):
        return self._tables[0]

    @property
    def mu_organic(self):
        return self._tables[1].reshape(-1, 1)

    @property
    def beta(self):
        return self._tables[2]

    @property
    def mu_bandit(self):
        return self._tables[3].reshape(-1, 1)

    @property
    def omega(self):
        return self._seq.omega().cpu().numpy().reshape(-1, 1) if self._seq is not None else None

    def __deepcopy__(self, memo):
        other = RecoEnv1()
        other.config = self.config
        other.action_space = getattr(self, 'action_space', None)
        other.observation_space = getattr(self, 'observation_space', None)
        other.time_generator = (DefaultTimeGenerator(self.config) if not getattr(self, '_time_mode', 0)
                                else self.time_generator) if self.config else None
        other._time_mode = getattr(self, '_time_mode', 0)
        other.agent = deepcopy(self.agent, memo)
        other._epoch = self._epoch
        other._tables = self._tables               # read-only
        other._device = self._device
        return other

    def __getstate__(self):
        st = dict(self.__dict__)
        st['_seq'] = None                          # device state is re-derived from the seed
        st['_bat'] = None
        return st

    # -- the batched device path ----------------------------------------------------------------
    def make_simulator(self, n_users, agent=None, log=True, device=None, policy=None):
        pol = policy if policy is not None else device_policy_of(agent)
        if pol is None:
            raise ValueError(f'{type(agent).__name__} cannot run inside the device step loop')
        pol = {k: v for k, v in pol.items() if k != 'ps_all'}
        if pol.get('policy_seed') is None:
            pol = dict(pol, policy_seed=self.seed)      # agent=None draws from the ENV stream
        return Simulator(self.config, n_users, epoch=self._epoch, tables=self._tables,
                         log_capacity=None if log else 0, device=device or self._device, **pol)

    def simulate(self, num_users, agent=None, num_organic_users=0, first_user_id=0, log=True,
                 device=None):
        """All users at once on the GPU -> (counters dict, Simulator).  User ids are
        first_user_id .. ; the first `num_organic_users` ids are organic-only warm-up users."""
        total = num_users + num_organic_users
        pol = device_policy_of(agent)
        sim = self.make_simulator(total, agent, log=log, device=device, policy=pol)
        log_retries = 0
        while True:
            sim.reset_users(first_user_id, total,
                            organic_only_below=first_user_id + num_organic_users)
            sim.run()
            cnt = sim.counters()
            if cnt['exact_overflow']:
                raise _abi.RecoGymHipError(f"{cnt['exact_overflow']} organic draws exceeded the float64 resolve scratch")
            if cnt['hist_overflow']:
                # a user viewed more distinct products than its history row holds (default 255): run again with
                # rows twice as long, the way a log overflow is retried with a larger buffer
                cap = 2 * max(int((pol.get('ouc') or {}).get('history_cap', 0) or pol.get('history_cap', 0) or 255), 255) + 1
                if cap > 4 * self.config.num_products + 512:
                    raise _abi.RecoGymHipError('per-user view history overflowed ouc_history_cap')
                sim.close()
                pol = dict(pol, ouc=dict(pol.get('ouc') or {}, history_cap=cap))
                sim = self.make_simulator(total, agent, log=log, device=device, policy=pol)
                continue
            if cnt['log_dropped'] == 0:
                return cnt, sim
            # the walk reserves rows per wave in chunks and leaves holes whose number depends on the scheduling: retry
            # with a proportional margin, not with the last run's exact need; a bounded number of times
            log_retries += 1
            if log_retries > 3:
                raise _abi.RecoGymHipError(f"the log overflowed {log_retries} times ({cnt['log_dropped']} rows dropped at "
                                           f"capacity {cnt['log_rows']})")
            from ..sim import default_log_capacity
            need = cnt['log_rows'] + cnt['log_dropped']
            sim.set_log_capacity(max(default_log_capacity(self.config, total), int(1.10 * need) + 65536 * log_retries))

    # -- gym.Env episode API (one user at a time; batch of 1 on the device) -----------------------
    def _seq_sim(self):
        if self._seq is None:
            self._seq = Simulator(self.config, 1, policy=_abi.RG_POLICY_EXTERNAL, epoch=self._epoch,
                                  tables=self._tables, log_capacity=1 << 16, device=self._device)
            self._act = torch.zeros(1, dtype=torch.int32, device=self._seq.device)
        return self._seq

    def reset(self, user_id=0):
        self.first_step = True
        self.state = organic
        self.time_generator.reset()
        if self.agent:
            self.agent.reset()
        self.current_time = 0 if self._time_mode else self.time_generator.new_time()
        self._event_index = 0
        self.current_user_id = user_id
        sim = self._seq_sim()
        sim.reseed(self.seed, self.seed)
        sim.reset_users(user_id, 1)
        self._rows_read = 0

    def _advance(self, action=None):
        """One Markov transition of the current user on the device -> the emitted row.  One launch sequence and ONE
        pinned-memory read-back of (row, state, clock) per event (rg_sim_step_user)."""
        row, self.state, clock = self._seq.step_user(action)
        self._rows_read += 1
        self._event_index += 1
        self.current_time = clock if self._time_mode else self.time_generator.new_time()
        return row

    def _context(self, t, u, n):
        return DefaultContext(t, u, n if self._time_mode else None)

    def generate_organic_sessions(self):
        session = OrganicSessions()
        while self.state == organic:
            t, u, n = self.current_time, self.current_user_id, self._event_index
            row = self._advance()
            session.next(self._context(t, u, n), int(row['v']))
        return session

    def step(self, action_id):
        info = {}
        if self.first_step:
            assert (action_id is None)
            self.first_step = False
            sessions = self.generate_organic_sessions()
            return (Observation(self._context(self.current_time, self.current_user_id, self._event_index), sessions),
                    None, self.state == stop, info)
        assert (action_id is not None)
        if not 0 <= int(action_id) < self.config.num_products:
            # the reference indexes ctr[action] (reco_env_v1.py:113): IndexError; the device would read out of bounds
            raise IndexError(f'action {action_id} is out of bounds for {self.config.num_products} products')
        row = self._advance(action_id)
        reward = int(row['c'])
        sessions = self.generate_organic_sessions() if self.state == organic \
            else self.empty_sessions
        return (Observation(self._context(self.current_time, self.current_user_id, self._event_index), sessions),
                reward, self.state == stop, info)

    def step_offline(self, observation, reward, done):
        if self.first_step:
            action = None
        else:
            assert (hasattr(self, 'agent'))
            assert (observation is not None)
            if self.agent:
                action = self.agent.act(observation, reward, done)
            else:
                P = self.config.num_products
                ctx = observation.context()
                w = rng.draw(self.seed, *ctx.draw_key(), 0, rng.DRAW_POLICY)
                action = {
                    't': ctx.time(), 'u': ctx.user(), 'a': rng.bounded(w[0], w[1], P),
                    'ps': 1.0 / P,
                    'ps-a': np.ones(P) / P if self.config.with_ps_all else (),
                }
        if done:
            return (action,
                    Observation(self._context(self.current_time, self.current_user_id, self._event_index),
                                self.empty_sessions),
                    0, done, None)
        observation, reward, done, info = self.step(action['a'] if action is not None else None)
        return action, observation, reward, done, info

    # -- generate_logs ----------------------------------------------------------------------------
    def generate_logs(self, num_offline_users, agent=None, num_organic_offline_users=0, first_user_id=0):
        """Logs of `agent` (None = uniform random actions) over the given number of users.  `first_user_id`
        (an extension; the reference always starts at 0) selects which addressed users are simulated."""
        use = agent if agent else self.agent
        pol = device_policy_of(use)
        if pol is not None:
            cnt, sim = self.simulate(num_offline_users, use, num_organic_offline_users, first_user_id=first_user_id)
            cols = sim.log_columns()
            sim.close()
            with_all = bool(getattr(self.config, 'with_ps_all', False)) and use is None
            df = columns_to_dataframe(cols, self.config.num_products, with_all)
            if pol.get('ps_all') is not None:              # the agent's whole distribution per bandit row
                df['ps-a'] = pd.Series(pol['ps_all'](df), dtype=object, copy=False)
            return df
        if self._time_mode or getattr(use, 'per_user_path', False) or getattr(self.config, 'per_user_path', False) or \
                not batch_safe(use):
            return self._generate_logs_per_user(num_offline_users, use, num_organic_offline_users, first_user_id)
        return self._generate_logs_batched(num_offline_users, use, num_organic_offline_users, first_user_id)

    def close(self):
        """Release the device simulators this environment keeps between calls (the per-user one and the batched one)."""
        for name in ('_bat', '_seq'):
            sim = getattr(self, name, None)
            if sim is not None:
                try:
                    sim.close()
                except Exception:
                    pass
                setattr(self, name, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- any Python agent, B users per launch ------------------------------------------------------
    def _batch_sim(self, n):
        """The B-user simulator with the external policy behind _generate_logs_batched (kept across calls)."""
        sim = getattr(self, '_bat', None)
        if sim is None or sim.n_users < n:
            if sim is not None:
                sim.close()
            sim = self._bat = self._external_sim(n)
        return sim

    def _external_sim(self, n):
        from ..sim import default_log_capacity
        return Simulator(self.config, n, policy=_abi.RG_POLICY_EXTERNAL, epoch=self._epoch, tables=self._tables,
                         log_capacity=default_log_capacity(self.config, n), device=self._device)

    def _generate_logs_batched(self, num_offline_users, agent, num_organic_offline_users, first_user_id=0, batch=4096):
        """The reference's loop (abstract.py:292-316, step_offline :199-239) for ANY Python agent, B users at a time: one
        `rg_sim_step` launch sequence and one read-back (the step's rows + the users' new states) per Markov transition of all B
        users, `agent.act` called on the host for every user that needs an action — a copy of the agent per user slot
        (`deepcopy`, the way the reference's harness copies agents: bench_agents.py:79,85), reset per user like `env.reset` does.
        Rows are those of the one-user-at-a-time path (`_generate_logs_per_user`) for every agent whose `act` depends on its
        own user's observations only.  `generate_logs` takes this route only for agents that pass `batch_safe()`: an agent that
        owns a sequential random stream (its per-slot copies would all replay the SAME draws) keeps the one-user-at-a-time
        path, and so does any agent with `per_user_path = True` / `batch_safe = False` (a model it trains while acting).  The
        copies share large read-only model state (`_shared_state_memo`); what a copy still owns bounds the batch."""
        P = self.config.num_products
        total = num_offline_users + num_organic_offline_users
        org_below = first_user_id + num_organic_offline_users
        B = int(min(batch, max(total, 1)))
        # a copy of the agent per user slot; large read-only model state (arrays, tensors, torch modules, fitted estimators) is
        # SHARED between the copies, and what a copy still owns bounds the batch (64 MB of agent state per batch at most)
        shared = _shared_state_memo(agent)
        own = _approx_owned_bytes(agent, shared)
        B = int(max(1, min(B, (64 << 20) // max(own, 1))))
        sim = self._batch_sim(B)
        sim.reseed(self.seed, self.seed)
        agents = [deepcopy(agent, dict(shared)) for _ in range(B)]
        acts_host = torch.zeros(sim.n_users, dtype=torch.int32).pin_memory()
        acts_dev = torch.zeros(sim.n_users, dtype=torch.int32, device=sim.device)
        acts_np = acts_host.numpy()
        frames = []
        EV_B, EV_C, EV_MASK = np.uint32(_abi.RG_EV_BANDIT), np.uint32(_abi.RG_EV_CLICK), np.uint32(_abi.RG_EV_INDEX_MASK)
        empty = self.empty_sessions
        ctx_of = self._context
        const_clock = self._clock_is_constant()
        for b0 in range(0, total, B):
            n = min(B, total - b0)
            first = first_user_id + b0
            sim.reset_users(first, n, organic_only_below=org_below)
            for ag in agents[:n]:
                ag.reset()
            live = np.ones(n, dtype=bool)
            sessions = [None] * n                     # organic rows since the user's last act
            last_reward = [None] * n
            pend = [None] * n                         # the action dict whose bandit event is in flight
            o_u, o_t, o_v = [], [], []                # organic rows, a chunk per step
            b_rows = []                               # (u, event index, t as logged, a, c, ps, ps-a) of bandit rows incl. the trailing one
            base, t = 0, 0
            while True:
                n_live = int(live.sum())
                if n_live == 0:
                    break
                if t:
                    acts_dev.copy_(acts_host, non_blocking=True)
                sim.step(acts_dev)
                rows = sim.log[base:base + n_live].cpu().numpy().view(np.uint32).reshape(-1, 4)
                st = sim.states().cpu().numpy()[:n]
                base += n_live
                uu = (rows[:, 0].astype(np.int64) - first)
                code = rows[:, 2]
                is_b = (code & EV_B) != 0
                idx = (code & EV_MASK).astype(np.int64)
                tt = 0 if const_clock else t
                uo = uu[~is_b]
                if uo.size:                           # organic rows of this step
                    vo = idx[~is_b]
                    o_u.append(uo + first); o_t.append(np.full(uo.size, t, dtype=np.int64)); o_v.append(vo)
                    for u, v in zip(uo.tolist(), vo.tolist()):
                        ss = sessions[u]
                        if ss is None:
                            ss = sessions[u] = OrganicSessions()
                        ss.append({'t': tt, 'u': u + first, 'z': 'pageview', 'v': v})
                ub = uu[is_b]
                if ub.size:                           # bandit rows: the click of the action sent down
                    cb = ((code[is_b] & EV_C) != 0).astype(np.int64)
                    for u, c in zip(ub.tolist(), cb.tolist()):
                        a = pend[u]
                        b_rows.append((u + first, t, a['t'], a['a'], c, a['ps'], a['ps-a'] if 'ps-a' in a else ()))
                        last_reward[u] = c
                # who acts now: every live user that is not organic after this transition
                t1 = t + 1
                t1c = 0 if const_clock else t1
                for u in np.flatnonzero(live & (st != organic)).tolist():
                    done = bool(st[u] == stop)
                    uid = u + first
                    if done and uid < org_below:      # warm-up users: organic rows only (abstract.py:293-297)
                        live[u] = False
                        continue
                    ss = sessions[u]
                    sessions[u] = None
                    a = agents[u].act(Observation(ctx_of(t1c, uid, t1), ss if ss is not None else empty), last_reward[u], done)
                    if done:                          # the trailing row: one more act, reward 0 (abstract.py:311-316)
                        b_rows.append((uid, t1, a['t'], a['a'], 0, a['ps'], a['ps-a'] if 'ps-a' in a else ()))
                        live[u] = False
                    else:
                        ai = a['a']
                        if not 0 <= int(ai) < P:
                            raise IndexError(f'action {ai} is out of bounds for {P} products')
                        acts_np[u] = ai
                        pend[u] = a
                t = t1
            frames.append(self._batched_frame(o_u, o_t, o_v, b_rows, const_clock))
        df = pd.concat(frames, ignore_index=True) if len(frames) > 1 else frames[0]
        wide_u = first_user_id + total > 65536
        wide_p = P > 65535
        df['u'] = pd.array(df['u'].tolist(), dtype=pd.UInt32Dtype() if wide_u else pd.UInt16Dtype())
        df['v'] = pd.array(df['v'].tolist(), dtype=pd.UInt32Dtype() if wide_p else pd.UInt16Dtype())
        df['a'] = pd.array(df['a'].tolist(), dtype=pd.UInt32Dtype() if wide_p else pd.UInt16Dtype())
        return df

    def _clock_is_constant(self):
        return False

    @staticmethod
    def _batched_frame(o_u, o_t, o_v, b_rows, const_clock):
        """The rows of one batch in the reference's order: by user, inside a user by event index (every event index of a user
        carries exactly one row: an organic view, a bandit event, or the trailing undrawn one)."""
        nu = np.concatenate(o_u) if o_u else np.zeros(0, dtype=np.int64)
        nt = np.concatenate(o_t) if o_t else np.zeros(0, dtype=np.int64)
        nv = np.concatenate(o_v) if o_v else np.zeros(0, dtype=np.int64)
        nb = len(b_rows)
        bu = np.fromiter((r[0] for r in b_rows), dtype=np.int64, count=nb)
        be = np.fromiter((r[1] for r in b_rows), dtype=np.int64, count=nb)
        order = np.lexsort((np.concatenate([nt, be]), np.concatenate([nu, bu])))
        n = order.size
        is_b = order >= nu.size
        src = np.where(is_b, order - nu.size, order)
        t_col = np.zeros(n, dtype=np.float32)
        v_col, a_col, ps_col, psa_col = [None] * n, [None] * n, [None] * n, [None] * n
        c_col = np.full(n, np.nan, dtype=np.float32)
        ob = np.flatnonzero(~is_b)
        if not const_clock:
            t_col[ob] = nt[src[ob]]
        for i, v in zip(ob.tolist(), nv[src[ob]].tolist()):
            v_col[i] = v
        for i, j in zip(np.flatnonzero(is_b).tolist(), src[is_b].tolist()):
            r = b_rows[j]
            t_col[i] = r[2]; a_col[i] = r[3]; c_col[i] = r[4]; ps_col[i] = r[5]; psa_col[i] = r[6]
        return pd.DataFrame({'t': t_col, 'u': np.concatenate([nu, bu])[order], 'z': np.where(is_b, 'bandit', 'organic').astype(object),
                             'v': v_col, 'a': a_col, 'c': c_col, 'ps': ps_col, 'ps-a': psa_col})

    def _generate_logs_per_user(self, num_offline_users, agent, num_organic_offline_users, first_user_id=0):
        """Any Python agent: the reference's loop (abstract.py:292-316), one user at a time."""
        old_agent, self.agent = self.agent, agent
        cols = {k: [] for k in ('t', 'u', 'z', 'v', 'a', 'c', 'ps', 'ps-a')}

        def put(t, u, z, v, a, c, ps, ps_a):
            for k, x in zip(cols, (t, u, z, v, a, c, ps, ps_a)):
                cols[k].append(x)

        def store_organic(obs):
            for s in obs.sessions():
                put(s['t'], s['u'], 'organic', s['v'], None, None, None, None)

        def store_bandit(action, reward):
            if action:
                put(action['t'], action['u'], 'bandit', None, action['a'], reward, action['ps'],
                    action['ps-a'] if 'ps-a' in action else ())

        uid = first_user_id
        for _ in range(num_organic_offline_users):
            self.reset(uid)
            uid += 1
            obs, _, _, _ = self.step(None)
            store_organic(obs)
        for _ in range(num_offline_users):
            self.reset(uid)
            uid += 1
            obs, reward, done, _ = self.step(None)
            while not done:
                store_organic(obs)
                action, obs, reward, done, _ = self.step_offline(obs, reward, done)
                store_bandit(action, reward)
            store_organic(obs)
            action, _, reward, done, _ = self.step_offline(obs, reward, done)
            assert done, 'Done must not be changed!'
            store_bandit(action, reward)
        self.agent = old_agent
        wide_u = uid > 65536
        wide_p = self.config.num_products > 65535
        udt = pd.UInt32Dtype() if wide_u else pd.UInt16Dtype()
        pdt = pd.UInt32Dtype() if wide_p else pd.UInt16Dtype()
        cols['t'] = np.array(cols['t'], dtype=np.float32)
        cols['u'] = pd.array(cols['u'], dtype=udt)
        cols['v'] = pd.array(cols['v'], dtype=pdt)
        cols['a'] = pd.array(cols['a'], dtype=pdt)
        cols['c'] = np.array(cols['c'], dtype=np.float32)
        return pd.DataFrame().from_dict(cols)
