"""Time generators — reference: recogym/envs/features/time/*.py.

DefaultTimeGenerator: t = per-user event index.  NormalTimeGenerator: the clock advances by |N(mu, sigma)| per event
and the omega drift is scaled by the time delta (reco_env_v1.py:89-98); on the device the increments are addressed
draws (RG_DRAW_TIME of (user, event index)) and the clock is per-user state, so this class only carries mu / sigma.
"""


class TimeGenerator:
    def __init__(self, config):
        self.config = config

    def new_time(self):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError


class DefaultTimeGenerator(TimeGenerator):
    """0, 1, 2, ... per user; reset() rewinds to 0 (default_time_generator.py:10-16)."""

    def __init__(self, config):
        super().__init__(config)
        self.current_time = 0

    def new_time(self):
        now = self.current_time
        self.current_time = now + 1
        return now

    def reset(self):
        self.current_time = 0


class NormalTimeGenerator(TimeGenerator):
    """normal_time_generator.py:7-31 — `normal_time_mu` / `normal_time_sigma` from the config (0 / 1 when absent).
    The device step loop keeps every user's clock (RecoEnv1 reads it back); new_time() is therefore not a host call."""

    def __init__(self, config):
        super().__init__(config)
        self.current_time = 0
        self.normal_time_mu = getattr(config, 'normal_time_mu', 0)
        self.normal_time_sigma = getattr(config, 'normal_time_sigma', 1)

    def new_time(self):
        raise NotImplementedError('the device step loop keeps the clock of a NormalTimeGenerator')

    def reset(self):
        self.current_time = 0
