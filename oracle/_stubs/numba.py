"""Stand-in for numba — TEST INFRASTRUCTURE ONLY (deterministic.py:11)."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


prange = range
