"""Frozen multinomial logistic-regression policy — the inference side of the reference's
LogregMulticlassIpsAgent (recogym/agents/logreg_ips.py:60-87, `select_randomly = False`): with a
fitted sklearn model the agent's act is `classes_[argmax(views @ coef_.T + intercept_)]` over the
user's cumulative view counts (ViewsFeaturesProvider, agents/abstract.py:316-409), `ps = 1`.

Training stays outside the step loop: `agents.feature_feed.train_data_from_log` builds the
reference's `(features, actions, deltas, pss)` from a log, sklearn fits on it (the reference's own
`build()`: weights = deltas / pss, logreg_ips.py:89-99), and this class carries the fitted arrays
into the device step loop (RG_POLICY_LOGREG_FROZEN).  Scores are accumulated exactly like scipy's
CSR x dense product (viewed products ascending, multiply then add, intercept last), so the action
is sklearn's predict() bit for bit — on the device, in the oracle and in `act` below."""
import numpy as np

from .. import _abi, rng
from .abstract import Agent


class LogregFrozenAgent(Agent):
    def __init__(self, config, coef, intercept, classes):
        """coef (n_classes, P) / intercept (n_classes,) / classes (n_classes,) as sklearn stores them;
        a two-class model (coef of one row) is expanded to two rows (zero first row).
        config.select_randomly = True (logreg_ips.py:61-66): the action is sampled from predict_proba — softmax of the
        same scores, one addressed policy draw — on the device where the model has a class per product and P <= 1024
        (rg_config.lr_select_randomly, k_logreg_sample), else on the host paths (needs every product as a class either way,
        like the reference's rng.choice(num_products, p=proba))."""
        super().__init__(config)
        self.select_randomly = bool(getattr(config, 'select_randomly', False))
        coef = np.atleast_2d(np.asarray(coef, dtype=np.float64))
        intercept = np.atleast_1d(np.asarray(intercept, dtype=np.float64))
        classes = np.asarray(classes, dtype=np.int32)
        if coef.shape[0] == 1 and len(classes) == 2:       # sklearn's binary form: score > 0 -> classes_[1]
            coef = np.vstack([np.zeros_like(coef), coef])
            intercept = np.r_[0.0, intercept]
        assert coef.shape == (len(classes), config.num_products) and intercept.shape == (len(classes),)
        self.coef_t = np.ascontiguousarray(coef.T)          # (P, n_classes): what the ABI takes
        self.intercept = np.ascontiguousarray(intercept)
        self.classes = np.ascontiguousarray(classes)
        # weight_history_function (ViewsFeaturesProvider with history, agents/abstract.py:343-382): time-weighted float features
        # instead of view counts — host path only
        self.history = None
        if getattr(config, 'weight_history_function', None) is not None:
            from .views_history import ViewsHistory
            self.history = ViewsHistory(config.num_products, config.weight_history_function, is_sparse=True)
        self.reset()

    @classmethod
    def from_sklearn(cls, config, logreg):
        return cls(config, logreg.coef_, logreg.intercept_, logreg.classes_)

    def device_policy(self):
        if getattr(self.config, 'with_ps_all', False) or self.history is not None:
            return None
        if self.select_randomly:
            # sampled from predict_proba on the device (k_logreg_sample: a wave per act) where the model has a class per product
            # — what the reference's rng.choice(num_products, p = proba) needs — and P <= 1024; else the host paths
            P = self.config.num_products
            if P > 1024 or len(self.classes) != P or not np.array_equal(self.classes, np.arange(P)):
                return None
            return dict(policy=_abi.RG_POLICY_LOGREG_FROZEN, policy_seed=self.config.random_seed, ouc=None,
                        logreg=dict(coef_t=self.coef_t, intercept=self.intercept, classes=self.classes, select_randomly=True))
        return dict(policy=_abi.RG_POLICY_LOGREG_FROZEN, policy_seed=0, ouc=None,
                    logreg=dict(coef_t=self.coef_t, intercept=self.intercept, classes=self.classes))

    def reset(self):
        self.views = np.zeros(self.config.num_products, dtype=np.int64)
        if getattr(self, 'history', None) is not None:
            self.history.reset()

    def act(self, observation, reward, done):
        for session in observation.sessions():
            self.views[int(session['v'])] += 1
        score = np.zeros(len(self.classes))
        if self.history is not None:
            # the weighted features as the reference's provider builds them (float32 sums per viewed product), through the same
            # CSR x dense product order: stored entries ascending, each widened to float64, multiply then add
            self.history.observe(observation)
            f = self.history.features(observation.context().time()).tocsr()
            f.sort_indices()
            for p, w in zip(f.indices, f.data):
                score = score + np.float64(w) * self.coef_t[p]
        else:
            for p in np.flatnonzero(self.views):             # ascending, multiply then add
                score = score + np.float64(self.views[p]) * self.coef_t[p]
        score = score + self.intercept
        if self.select_randomly:
            # sklearn's multinomial predict_proba: softmax(decision_function) (sklearn.utils.extmath.softmax)
            e = np.exp(score - score.max())
            proba = e / e.sum()
            assert len(proba) == self.config.num_products, 'select_randomly needs every product as a class'
            ctx = observation.context()
            key = ctx.draw_key() if hasattr(ctx, 'draw_key') else (ctx.user(), ctx.time())
            _, _, u1 = rng.policy_uniforms(self.config.random_seed, *key)
            cdf = proba.cumsum()
            cdf /= cdf[-1]
            a = int(cdf.searchsorted(u1, side='right'))
            return {**super().act(observation, reward, done), 'a': a, 'ps': float(proba[a]),
                    'ps-a': proba if getattr(self.config, 'with_ps_all', False) else ()}
        a = int(self.classes[int(np.argmax(score))])
        ps_all = ()
        if getattr(self.config, 'with_ps_all', False):       # logreg_ips.py:73-80: a one-hot vector over the products
            ps_all = np.zeros(self.config.num_products)
            ps_all[a] = 1.0
        return {**super().act(observation, reward, done), 'a': a, 'ps': 1.0, 'ps-a': ps_all}
