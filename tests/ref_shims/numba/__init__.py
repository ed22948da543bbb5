"""Identity stand-in for numba (not installed here): njit/jit return the function unchanged,
bare or with keyword arguments.  Only sig/ff on the hot path use it (reco_env_v1.py:32-41)."""


def _identity(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


njit = jit = _identity
