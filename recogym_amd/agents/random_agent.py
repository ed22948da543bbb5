"""RandomAgent — reference: recogym/agents/random_agent.py:22-33 (uniform action, ps = 1/P)."""
import numpy as np

from .. import _abi, rng
from ..envs.configuration import Configuration
from .abstract import Agent

random_args = {
    'num_products': 10,
    'random_seed': np.random.randint(2 ** 31 - 1),
    'with_ps_all': False,
}


class RandomAgent(Agent):
    def __init__(self, config=Configuration(random_args)):
        super().__init__(config)

    def device_policy(self):
        if getattr(self.config, 'with_ps_all', False):
            return None
        return dict(policy=_abi.RG_POLICY_RANDOM_AGENT, policy_seed=self.config.random_seed,
                    ouc=None)

    def act(self, observation, reward, done):
        P = self.config.num_products
        ctx = observation.context()
        user, t = ctx.draw_key() if hasattr(ctx, 'draw_key') else (ctx.user(), ctx.time())
        w = rng.draw(self.config.random_seed, user, t, 0, rng.DRAW_POLICY)
        return {
            **super().act(observation, reward, done),
            'a': rng.bounded(w[0], w[1], P),
            'ps': 1.0 / float(P),
            'ps-a': np.ones(P) / P if getattr(self.config, 'with_ps_all', False) else (),
        }
