"""Stand-in for numba so that the *reference* package can be imported un-jitted
in the build container (test infrastructure only; contains no reference code).
``njit``/``jit`` return the decorated function unchanged."""
__version__ = "0.0.stub"


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(func):
        return func
    return wrap


njit = _identity_decorator
jit = _identity_decorator
