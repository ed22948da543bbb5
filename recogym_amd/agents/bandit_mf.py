"""BanditMFSquareAgent — reference: recogym/agents/bandit_mf.py.

Matrix-factorisation bandit: logit(a | last product viewed l) = <E_p[a], E_u[l]>, trained online
with mini-batches through `train`, acting by argmax.  Here:

* inference is the frozen per-product table of LastViewTableAgent inside the device step loop;
* training keeps the reference's arithmetic (torch nn.Embedding x 2, BCEWithLogitsLoss, RMSprop,
  one optimiser step every `mini_batch_size` train calls on the samples gathered since the last one
  — the call that triggers the step contributes no sample, bandit_mf.py:110-126) but is fed from a
  whole log at once: `train_from_log` rebuilds the (last product viewed, action, reward) triple of
  every train call with array operations (forward fill of the organic views) and replays the
  optimiser steps.  `train` (observation by observation) is kept and produces the same weights.
"""
import numpy as np

from ..envs.configuration import Configuration
from .abstract import Agent
from .last_view_table import LastViewTableAgent

bandit_mf_square_args = {
    'num_products': 10,
    'embed_dim': 5,
    'mini_batch_size': 32,
    'learning_rate': 0.01,
    'with_ps_all': False,
}


class BanditMFSquareAgent(Agent):
    needs_training = True

    def __init__(self, config=Configuration(bandit_mf_square_args), product_embedding=None, user_embedding=None):
        """Initial embeddings: given arrays (P, embed_dim), else torch's nn.Embedding default init under
        the current torch seed (product table first, then user table, like the reference's __init__)."""
        import torch
        from torch import nn, optim
        super().__init__(config)
        P, E = config.num_products, config.embed_dim
        self.product_embedding = nn.Embedding(P, E)
        self.user_embedding = nn.Embedding(P, E)
        with torch.no_grad():
            if product_embedding is not None:
                self.product_embedding.weight.copy_(torch.as_tensor(np.asarray(product_embedding, dtype=np.float32)))
            if user_embedding is not None:
                self.user_embedding.weight.copy_(torch.as_tensor(np.asarray(user_embedding, dtype=np.float32)))
        params = list(self.product_embedding.parameters()) + list(self.user_embedding.parameters())
        self.optimizer = optim.RMSprop(params, lr=getattr(config, 'learning_rate', 0.01))
        self.loss = nn.BCEWithLogitsLoss()
        self.batch = getattr(config, 'mini_batch_size', 32)
        self.last_product_viewed = None
        self.curr_step = 0
        self.train_data = ([], [], [])
        self._table = None

    # -- the reference's train, call by call ------------------------------------------------------
    def _update(self, lpvs, actions, rewards):
        import torch
        if len(lpvs) == 0:
            return
        self.optimizer.zero_grad()
        a = self.product_embedding(torch.as_tensor(np.asarray(actions, dtype=np.int64)))
        b = self.user_embedding(torch.as_tensor(np.asarray(lpvs, dtype=np.int64)))
        logit = torch.sum(a * b, dim=1)
        loss = self.loss(logit, torch.as_tensor(np.asarray(rewards, dtype=np.float32)))
        loss.backward()
        self.optimizer.step()
        self._table = None

    def train(self, observation, action, reward, done=False):
        if observation.sessions():
            self.last_product_viewed = observation.sessions()[-1]['v']
        self.curr_step += 1
        if self.curr_step % self.batch == 0:
            self._update(*self.train_data)
            self.train_data = ([], [], [])
        elif action is not None and reward is not None:
            self.train_data[0].append(self.last_product_viewed)
            self.train_data[1].append(action['a'])
            self.train_data[2].append(reward)

    # -- the same, from a whole log -----------------------------------------------------------------
    def train_from_log(self, log, num_organic_users=0):
        """`log`: DataFrame of generate_logs or Simulator.log_columns() in the reference's row order (the
        first `num_organic_users` users are organic-only: the offline protocol calls train once for
        each of them with action None, bench_agents.py:168-175).  Equivalent to the train calls the
        reference's offline protocol makes, in the same order."""
        from .feature_feed import _columns_of
        u, is_b, v, a, c, _ = _columns_of(log)
        n = len(u)
        # last product viewed at every row = forward fill of the organic views (never empty: a user
        # starts with an organic row); it deliberately carries over user boundaries like the reference
        idx = np.where(~is_b, np.arange(n), -1)
        np.maximum.accumulate(idx, out=idx)
        lpv_all = np.where(idx >= 0, v[np.maximum(idx, 0)], -1)
        # the train calls: one per organic-only user (after its rows), one per bandit row
        first_main = 0
        call_pos, call_has = [], []
        if num_organic_users:
            users = u[np.r_[True, u[1:] != u[:-1]]]
            org_users = users[:num_organic_users]
            last_row = np.flatnonzero(np.r_[u[1:] != u[:-1], True])
            call_pos.append(last_row[:num_organic_users])
            call_has.append(np.zeros(num_organic_users, dtype=bool))
            first_main = int(last_row[num_organic_users - 1]) + 1
            del org_users
        b_rows = np.flatnonzero(is_b)
        b_rows = b_rows[b_rows >= first_main]
        call_pos.append(b_rows)
        call_has.append(np.ones(len(b_rows), dtype=bool))
        pos = np.concatenate(call_pos)
        has = np.concatenate(call_has)
        k = self.curr_step + 1 + np.arange(len(pos))               # 1-based call numbers
        lpv, act, rew = lpv_all[pos], a[pos], np.nan_to_num(c[pos])
        if self.last_product_viewed is not None:                     # (a call before any view of this log)
            lpv = np.where(lpv < 0, self.last_product_viewed, lpv)
        step_calls = np.flatnonzero(k % self.batch == 0)
        start = 0
        pend = tuple(list(x) for x in self.train_data)
        for sc in step_calls:
            sel = np.flatnonzero(has[start:sc]) + start                # samples since the last step; call sc itself adds none
            self._update(pend[0] + lpv[sel].tolist(), pend[1] + act[sel].tolist(), pend[2] + rew[sel].tolist())
            pend = ([], [], [])
            start = sc + 1
        sel = np.flatnonzero(has[start:]) + start
        self.train_data = (pend[0] + lpv[sel].tolist(), pend[1] + act[sel].tolist(), pend[2] + rew[sel].tolist())
        self.curr_step += len(pos)
        if len(pos) and lpv_all[pos[-1]] >= 0:
            self.last_product_viewed = int(lpv_all[n - 1]) if lpv_all[n - 1] >= 0 else self.last_product_viewed

    # -- acting ---------------------------------------------------------------------------------------
    def frozen(self):
        if self._table is None:
            self._table = LastViewTableAgent.from_bandit_mf(
                self.config, self.product_embedding.weight.detach().numpy(),
                self.user_embedding.weight.detach().numpy())
        return self._table

    def device_policy(self):
        return self.frozen().device_policy()

    def act(self, observation, reward, done):
        return self.frozen().act(observation, reward, done)

    def reset(self):
        pass
