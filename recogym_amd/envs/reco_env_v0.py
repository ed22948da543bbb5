"""RecoEnv0 — the reference's `reco-gym-v0` environment (the cluster toy model) over the same HIP step loop.

Reference: recogym/envs/reco_env_v0.py:1-69 — users move inside clusters of products (a block-diagonal organic transition
matrix), clicks come from a P x P matrix `f(P / 5 (T + T') + phi)`, the Markov organic / bandit / stop loop is AbstractEnv's
(abstract.py:105-239).  There is no user embedding: every draw is a table look-up, so the device path is the lock-step
loop of RecoEnv1 with its own draw kernel (`rg_config.env_kind = 1`: k_draw_env0, the click of k_advance), behind the same
class surface — `generate_logs`, `reset` / `step` / `step_offline`, `test_agent`, `verify_agents`.

As in the reference, `update_state` of this env never asks the time generator for a new time (reco_env_v0.py:56-58): the
`t` of every row of a user is the time `reset` set (0).  The device rows keep the event index (it orders the log and
addresses the draws); the DataFrame's `t` column and the contexts' `time()` are the reference's constant.
"""
import numpy as np

from .. import _abi
from ..sim import Simulator
from .configuration import Configuration
from .context import DefaultContext
from .features.time import DefaultTimeGenerator
from .reco_env_v1 import Discrete, RecoEnv1, env_args, organic
from .static_params import draw_env0_tables

# reco_env_v0.py:7-13
env_0_args = {
    **env_args,
    'num_clusters': 2,
    'phi_var': 0.1,
}


class RecoEnv0(RecoEnv1):

    def init_gym(self, args):
        self.config = Configuration(args)
        if self.config.num_products % self.config.num_clusters:
            # (the reference's kron of `num_clusters` blocks of int(P / num_clusters) products has the wrong shape then and
            # its division raises, reco_env_v0.py:33-37)
            raise ValueError('operands could not be broadcast together: num_products must be a multiple of num_clusters')
        self.action_space = Discrete(self.config.num_products)
        self.observation_space = Discrete(self.config.num_products)
        self._time_mode = 0
        self.time_generator = args['time_generator'] if 'time_generator' in args else DefaultTimeGenerator(self.config)
        self.agent = args['agent'] if 'agent' in args else None
        self.reset_random_seed()
        self._tables = draw_env0_tables(self.config)      # set_static_params, the reference's own numpy calls

    # the reference's attribute names, for notebooks
    @property
    def product_transition(self):
        return self._tables['product_transition']

    @property
    def click_probs(self):
        return self._tables['click_probs']

    @property
    def phi(self):
        return self._tables['phi']

    Gamma = mu_organic = beta = mu_bandit = omega = property(lambda self: None)

    def __deepcopy__(self, memo):
        from copy import deepcopy
        other = RecoEnv0()
        other.config = self.config
        other.action_space = getattr(self, 'action_space', None)
        other.observation_space = getattr(self, 'observation_space', None)
        other.time_generator = DefaultTimeGenerator(self.config) if self.config else None
        other._time_mode = 0
        other.agent = deepcopy(self.agent, memo)
        other._epoch = self._epoch
        other._tables = self._tables               # read-only
        other._device = self._device
        return other

    def make_simulator(self, n_users, agent=None, log=True, device=None, policy=None):
        from .reco_env_v1 import device_policy_of
        pol = policy if policy is not None else device_policy_of(agent)
        if pol is None or pol.get('policy') in (_abi.RG_POLICY_LAST_VIEW_TABLE, _abi.RG_POLICY_LOGREG_FROZEN):
            raise ValueError(f'{type(agent).__name__} cannot run inside the device step loop of reco-gym-v0')
        pol = {k: v for k, v in pol.items() if k != 'ps_all'}
        if pol.get('policy_seed') is None:
            pol = dict(pol, policy_seed=self.seed)
        return Simulator(self.config, n_users, epoch=self._epoch, env0=self._tables,
                         log_capacity=None if log else 0, device=device or self._device, **pol)

    def _seq_sim(self):
        import torch
        if self._seq is None:
            self._seq = Simulator(self.config, 1, policy=_abi.RG_POLICY_EXTERNAL, epoch=self._epoch,
                                  env0=self._tables, log_capacity=1 << 16, device=self._device)
            self._act = torch.zeros(1, dtype=torch.int32, device=self._seq.device)
        return self._seq

    def _advance(self, action=None):
        row, self.state, _ = self._seq.step_user(action)
        self._rows_read += 1
        self._event_index += 1          # (the clock itself does not move: reco_env_v0.py:56-58)
        return row

    def _context(self, t, u, n):
        return DefaultContext(t, u, n)   # time() = the reference's constant; the draws are addressed by the event index

    def _clock_is_constant(self):
        return True

    def _external_sim(self, n):
        from ..sim import default_log_capacity
        return Simulator(self.config, n, policy=_abi.RG_POLICY_EXTERNAL, epoch=self._epoch, env0=self._tables,
                         log_capacity=default_log_capacity(self.config, n), device=self._device)

    def generate_logs(self, num_offline_users, agent=None, num_organic_offline_users=0, first_user_id=0):
        use = agent if agent else self.agent
        from .reco_env_v1 import device_policy_of
        pol = device_policy_of(use)
        if pol is not None and pol.get('policy') in (_abi.RG_POLICY_LAST_VIEW_TABLE, _abi.RG_POLICY_LOGREG_FROZEN):
            return self._generate_logs_per_user(num_offline_users, use, num_organic_offline_users, first_user_id)
        df = super().generate_logs(num_offline_users, agent, num_organic_offline_users, first_user_id)
        if pol is not None:
            df['t'] = np.zeros(len(df), dtype=np.float32)       # every row of a user carries reset()'s time
        return df
