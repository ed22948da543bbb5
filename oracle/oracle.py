"""ctypes binding of oracle/librecogym_oracle.so — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module, and
only as the checker.  See recogym_oracle.c for what the oracle is and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from recogym_amd import _abi
from recogym_amd.envs.static_params import draw_tables, make_rg_config

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'librecogym_oracle.so')

RNG_PHILOX, RNG_MT = 0, 1

ROW_DTYPE = np.dtype([('u', np.uint32), ('t', np.uint32), ('z', np.int32), ('v', np.int32),
                      ('a', np.int32), ('c', np.int32), ('phantom', np.int32),
                      ('pad', np.int32), ('ps', np.float64), ('p_click', np.float64), ('time', np.float64)])


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(HERE, 'recogym_oracle.c')
    deps = [src, os.path.join(HERE, '..', 'include', 'recogym_hip.h'),
            os.path.join(HERE, '..', 'include', 'recogym_rng.h')]
    if (not force and os.path.exists(LIB)
            and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)):
        return LIB
    subprocess.check_call(['make', '-C', HERE, '-B', 'librecogym_oracle.so'],
                          stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.rgo_env_create.restype = C.c_void_p
        L.rgo_env_create.argtypes = [C.POINTER(_abi.RgConfig), C.c_int] + [C.c_void_p] * 4
        L.rgo_env_destroy.argtypes = [C.c_void_p]
        L.rgo_env_set_policy_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rgo_env_set_env0.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.rgo_env_set_logreg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.rgo_env_reseed.argtypes = [C.c_void_p, C.c_uint64]
        L.rgo_env_reseed_policy.argtypes = [C.c_void_p, C.c_uint64]
        L.rgo_env_reset.argtypes = [C.c_void_p, C.c_uint32]
        L.rgo_env_step.restype = C.c_int
        L.rgo_env_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
        L.rgo_env_time.restype = C.c_uint32
        L.rgo_env_time.argtypes = [C.c_void_p]
        L.rgo_env_clock.restype = C.c_double
        L.rgo_env_clock.argtypes = [C.c_void_p]
        L.rgo_env_state.restype = C.c_int
        L.rgo_env_state.argtypes = [C.c_void_p]
        L.rgo_env_omega.argtypes = [C.c_void_p, C.c_void_p]
        L.rgo_env_last_p_click.restype = C.c_double
        L.rgo_env_last_p_click.argtypes = [C.c_void_p]
        L.rgo_env_policy_act.restype = C.c_int32
        L.rgo_env_policy_act.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.rgo_env_generate_logs.restype = C.c_int64
        L.rgo_env_generate_logs.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                            C.c_void_p, C.c_uint64]
        L.rgo_env_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.rgo_philox.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
        L.rgo_uniform.restype = C.c_double
        L.rgo_uniform.argtypes = [C.c_uint32, C.c_uint32]
        L.rgo_bounded.restype = C.c_uint32
        L.rgo_bounded.argtypes = [C.c_uint32] * 3
        L.rgo_ff.restype = C.c_double
        L.rgo_ff.argtypes = [C.c_double]
        L.rgo_mt_probe.argtypes = [C.c_uint32, C.c_uint32, C.c_int] + [C.c_void_p] * 3
        _lib = L
    return _lib


class OracleEnv:
    """The reference's RecoEnv1, restated: reset / step / generate_logs over the C library."""

    def __init__(self, config, rng_mode=RNG_PHILOX, policy=_abi.RG_POLICY_UNIFORM_ENV,
                 policy_seed=None, ouc=None, epoch=0, tables=None, policy_table=None, policy_ps=None,
                 logreg=None, env0=None):
        self.config = config
        self.rg_config = make_rg_config(config, config.random_seed + epoch, policy, policy_seed,
                                        ouc, env_kind=1 if env0 is not None else 0,
                                        lr_select_randomly=bool(logreg and logreg.get('select_randomly')))
        if env0 is not None:       # reco-gym-v0: dict of recogym_amd.envs.static_params.draw_env0_tables
            self.tables = []
            self._ptrs = [None] * 4
            self._h = lib().rgo_env_create(C.byref(self.rg_config), rng_mode, *self._ptrs)
            self._click_p = np.ascontiguousarray(env0['click_probs'], dtype=np.float64)
            lib().rgo_env_set_env0(self._h, self._click_p.ctypes.data_as(C.c_void_p), int(env0['cluster_size']))
        else:
            self.tables = tables if tables is not None else draw_tables(config)
            self._ptrs = [t.ctypes.data_as(C.c_void_p) for t in self.tables]
            self._h = lib().rgo_env_create(C.byref(self.rg_config), rng_mode, *self._ptrs)
        self._rows = np.zeros(4096, dtype=ROW_DTYPE)
        if policy_table is not None:
            self._pt = np.ascontiguousarray(policy_table, dtype=np.int32)
            self._pp = None if policy_ps is None else np.ascontiguousarray(policy_ps, dtype=np.float64)
            lib().rgo_env_set_policy_table(self._h, self._pt.ctypes.data_as(C.c_void_p),
                                           None if self._pp is None else self._pp.ctypes.data_as(C.c_void_p))

        if logreg is not None:     # dict(coef_t (P, C) float64, intercept (C,), classes (C,) int32)
            self._lr = (np.ascontiguousarray(logreg['coef_t'], dtype=np.float64),
                        np.ascontiguousarray(logreg['intercept'], dtype=np.float64),
                        np.ascontiguousarray(logreg['classes'], dtype=np.int32))
            lib().rgo_env_set_logreg(self._h, *[x.ctypes.data_as(C.c_void_p) for x in self._lr],
                                     len(self._lr[2]))

    def __del__(self):
        if getattr(self, '_h', None):
            lib().rgo_env_destroy(self._h)
            self._h = None

    def reseed(self, seed):
        lib().rgo_env_reseed(self._h, seed)

    def reseed_policy(self, seed):
        lib().rgo_env_reseed_policy(self._h, seed)

    def reset(self, user_id=0):
        lib().rgo_env_reset(self._h, user_id)

    def step(self, action):
        """-> (organic rows of the new observation, reward or None, done)"""
        n = C.c_uint64(0)
        done = C.c_int32(0)
        r = lib().rgo_env_step(self._h, -1 if action is None else int(action),
                               self._rows.ctypes.data_as(C.c_void_p), len(self._rows),
                               C.byref(n), C.byref(done))
        if r <= -100:
            raise AssertionError('first step must be None / later steps need an action')
        assert n.value <= len(self._rows)
        return self._rows[:n.value].copy(), (None if r < 0 else r), bool(done.value)

    def policy_act(self):
        ps = C.c_double(0)
        a = lib().rgo_env_policy_act(self._h, C.byref(ps))
        return a, ps.value

    @property
    def time(self):
        return lib().rgo_env_time(self._h)

    @property
    def clock(self):
        return lib().rgo_env_clock(self._h)

    @property
    def state(self):
        return lib().rgo_env_state(self._h)

    @property
    def omega(self):
        out = np.zeros(self.config.K)
        lib().rgo_env_omega(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    @property
    def last_p_click(self):
        return lib().rgo_env_last_p_click(self._h)

    def generate_logs(self, n_users, n_organic_users=0, first_user_id=0, capacity=None):
        # (each user lives ~Geometric(prob_leave_organic) events plus its phantom row: the stop column of both rows of the
        # transition matrix, reco_env_v1.py:54-61)
        mean = 1.0 / max(float(getattr(self.config, 'prob_leave_organic', 0.01)), 1e-6) + 2.0
        cap = capacity or int((n_users + n_organic_users) * max(140.0, 1.4 * mean) + 12.0 * mean * (n_users + n_organic_users) ** 0.5 + 4096)
        while True:
            rows = np.zeros(cap, dtype=ROW_DTYPE)
            n = lib().rgo_env_generate_logs(self._h, first_user_id, n_users, n_organic_users,
                                            rows.ctypes.data_as(C.c_void_p), cap)
            if n <= cap:
                return rows[:n]
            raise RuntimeError(f'oracle log capacity {cap} < {n} rows; pass capacity=')

    def counters(self):
        out = np.zeros(4, dtype=np.int64)
        lib().rgo_env_counters(self._h, out.ctypes.data_as(C.c_void_p))
        return dict(organic=int(out[0]), bandit=int(out[1]), clicks=int(out[2]),
                    phantom=int(out[3]))


def philox(c, k):
    out = np.zeros(4, dtype=np.uint32)
    lib().rgo_philox(*[int(x) for x in c], *[int(x) for x in k], out.ctypes.data_as(C.c_void_p))
    return out
