"""Time generators — reference: recogym/envs/features/time/*.py.

Only DefaultTimeGenerator (t = per-user event index) is supported by the device step loop;
NormalTimeGenerator is listed as "next" in SURVEY.md §8f-4 and init_gym rejects it loudly.
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
