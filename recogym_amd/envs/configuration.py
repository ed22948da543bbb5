"""Immutable attribute bag built from an args dict (reference: recogym/envs/configuration.py).

Callers read `env.config.<key>` and `agent.config.<key>`; assignment after construction is
silently ignored and deepcopy returns the same object, exactly like the reference's class —
the evaluation harness deep-copies envs and agents freely and relies on both behaviours.
"""


class Configuration:
    def __init__(self, args):
        for key, value in dict(args).items():
            object.__setattr__(self, key, value)
        object.__setattr__(self, '_keys', tuple(args))

    def __setattr__(self, key, value):
        return None          # frozen after __init__

    def __deepcopy__(self, memo=None):
        return self

    def __contains__(self, key):
        return key in self._keys

    def as_dict(self):
        return {k: getattr(self, k) for k in self._keys}

    def __repr__(self):
        return f'Configuration({self.as_dict()!r})'
