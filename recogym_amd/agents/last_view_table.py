"""Frozen "last product viewed -> action" policies — the inference side of the reference's
BanditMFSquare (recogym/agents/bandit_mf.py:44-87): with trained embeddings the agent's act is
`argmax_a <E_p[a], E_u[last_product_viewed]>`, a function of the last viewed product only, i.e. a
P-entry table.  Training stays the reference's code (out of scope); this class takes
the trained embeddings (or any table) and runs inside the device step loop."""
import numpy as np

from .. import _abi
from .abstract import Agent


class LastViewTableAgent(Agent):
    def __init__(self, config, table, ps=None):
        super().__init__(config)
        self.table = np.ascontiguousarray(table, dtype=np.int32)
        assert self.table.shape == (config.num_products,)
        assert self.table.min() >= 0 and self.table.max() < config.num_products
        self.ps = None if ps is None else np.ascontiguousarray(ps, dtype=np.float64)
        self.last_product_viewed = None

    @classmethod
    def from_bandit_mf(cls, config, product_embedding, user_embedding):
        """product_embedding, user_embedding: (P, embed_dim) arrays (the two nn.Embedding weights
        of a trained BanditMFSquare).  `ps` reproduces the reference's quirk of logging the
        winning LOGIT as `ps` (bandit_mf.py:84)."""
        # float32 like the reference's torch forward (bandit_mf.py:44-52)
        Ep = np.asarray(product_embedding, dtype=np.float32)
        Eu = np.asarray(user_embedding, dtype=np.float32)
        # [last viewed product][action] logits, a block of rows at a time: the full P x P x E broadcast is 2 GB at
        # P = 10^4 (BASELINE config 5) and impossible at 10^5.  Every row keeps numpy's own float32 summation over E.
        P, E = Ep.shape
        table = np.empty(P, dtype=np.int64)
        win = np.empty(P, dtype=np.float32)
        rows = max(1, min(P, (64 << 20) // max(1, P * E * 4)))
        for lo in range(0, P, rows):
            logits = (Eu[lo:lo + rows, None, :] * Ep[None, :, :]).sum(axis=2)
            table[lo:lo + rows] = logits.argmax(axis=1)
            win[lo:lo + rows] = logits[np.arange(logits.shape[0]), table[lo:lo + rows]]
        return cls(config, table, win)

    def device_policy(self):
        if getattr(self.config, 'with_ps_all', False):
            return None
        return dict(policy=_abi.RG_POLICY_LAST_VIEW_TABLE, policy_seed=0, ouc=None,
                    policy_table=self.table, policy_ps=self.ps)

    def act(self, observation, reward, done):
        if observation.sessions():
            self.last_product_viewed = int(observation.sessions()[-1]['v'])
        a = int(self.table[self.last_product_viewed])
        return {**super().act(observation, reward, done), 'a': a,
                'ps': 1.0 if self.ps is None else float(self.ps[self.last_product_viewed]),
                'ps-a': ()}
