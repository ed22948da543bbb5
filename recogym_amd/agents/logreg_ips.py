"""LogregMulticlassIpsAgent — reference: recogym/agents/logreg_ips.py (select_randomly = False).

Same protocol as the reference's ModelBasedAgent (agents/abstract.py:284-314): `train` collects
the rows it is shown, the first `act` builds the model, then every act is
`classes_[argmax(views @ coef_.T + intercept_)]`.  What differs is where the work happens:

* training set: `train_data_from_log` (vectorised `AbstractFeatureProvider.train_data`) instead of
  the per-user `iterrows` loop; `train_from_log(log)` takes a whole log at once — `test_agent`
  uses it with a log the device produced, which holds exactly the rows the reference's per-user
  offline protocol would have shown to `train` (bench_agents.py:168-190: the env's own uniform
  policy acts; trajectories are keyed by (seed, user id));
* fit: sklearn's LogisticRegression with the reference's arguments (logreg_ips.py:89-99:
  weights = deltas / pss, lbfgs, max_iter, multinomial, random_state = config.random_seed);
* inference: frozen into the device step loop (LogregFrozenAgent / RG_POLICY_LOGREG_FROZEN).
"""
import warnings

import numpy as np

from ..envs.configuration import Configuration
from .abstract import Agent
from .feature_feed import train_data_from_log
from .logreg_frozen import LogregFrozenAgent

logreg_multiclass_ips_args = {
    'num_products': 10,
    'number_of_flips': 1,
    'random_seed': np.random.randint(2 ** 31 - 1),
    'select_randomly': False,
    'poly_degree': 2,
    'solver': 'lbfgs',
    'max_iter': 5000,
    'with_ps_all': False,
}


class LogregMulticlassIpsAgent(Agent):
    needs_training = True     # test_agent: run the offline protocol before asking for device_policy()

    def __init__(self, config=Configuration(logreg_multiclass_ips_args)):
        super().__init__(config)
        self._rows = {k: [] for k in ('t', 'u', 'is_bandit', 'v', 'a', 'c', 'ps')}
        self._log = None
        self.logreg = None
        self.frozen = None

    # -- training data ----------------------------------------------------------------------
    def train(self, observation, action, reward, done=False):
        """ModelBuilder.train (agents/abstract.py:55-83): the organic rows of the observation, then the action."""
        r = self._rows
        for s in observation.sessions():
            r['t'].append(s['t']); r['u'].append(s['u']); r['is_bandit'].append(False); r['v'].append(s['v'])
            r['a'].append(0); r['c'].append(np.nan); r['ps'].append(np.nan)
        if action:
            r['t'].append(action['t']); r['u'].append(action['u']); r['is_bandit'].append(True); r['v'].append(0)
            r['a'].append(action['a']); r['c'].append(reward); r['ps'].append(action['ps'])

    def train_from_log(self, log, num_organic_users=0):
        """A whole log (DataFrame of generate_logs or Simulator.log_columns()) instead of row-by-row train calls."""
        self._log = log

    def _training_set(self):
        wf = getattr(self.config, 'weight_history_function', None)     # time-weighted views (agents/abstract.py:216-263)
        if self._log is not None:
            return train_data_from_log(self._log, self.config.num_products, weight_history_function=wf)
        r = self._rows
        cols = dict(t=np.asarray(r['t'], dtype=np.float64),
                    u=np.asarray(r['u'], dtype=np.int64).astype(np.uint32).view(np.int32),
                    is_bandit=np.asarray(r['is_bandit'], dtype=bool),
                    v=np.asarray(r['v'], dtype=np.int32), a=np.asarray(r['a'], dtype=np.int32),
                    c=np.asarray(r['c'], dtype=np.float32), ps=np.asarray(r['ps'], dtype=np.float64))
        return train_data_from_log(cols, self.config.num_products, weight_history_function=wf)

    def build(self):
        """logreg_ips.py:89-99."""
        from sklearn.linear_model import LogisticRegression
        feats, actions, deltas, pss = self._training_set()
        c = self.config
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')          # sklearn deprecates the spelled-out 'multinomial'
            kw = dict(solver=getattr(c, 'solver', 'lbfgs'), max_iter=getattr(c, 'max_iter', 5000),
                      random_state=c.random_seed)
            try:       # the reference asks for the softmax model explicitly (logreg_ips.py:93): with only two
                # distinct actions in the log sklearn's default would fit a binary/OvR model instead
                model = LogisticRegression(multi_class='multinomial', **kw)
            except TypeError:                        # a scikit-learn without the argument: multinomial is its only form
                model = LogisticRegression(**kw)
            self.logreg = model.fit(feats, actions, deltas / pss)
        self.frozen = LogregFrozenAgent.from_sklearn(c, self.logreg)
        return self.frozen

    # -- acting -----------------------------------------------------------------------------
    def _ready(self):
        if self.frozen is None:
            self.build()
        return self.frozen

    def device_policy(self):
        return self._ready().device_policy()

    def act(self, observation, reward, done):
        return self._ready().act(observation, reward, done)

    def reset(self):
        if self.frozen is not None:
            self.frozen.reset()
