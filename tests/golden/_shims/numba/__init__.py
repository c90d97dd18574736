"""Import stand-in (test-infra only): ark.utils.data_utils decorates helpers this repository never calls
with numba.njit; the decorator passes the function through."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


jit = njit


def prange(*a):
    return range(*a)


class _Namespace:
    def __getattr__(self, name):      # nb.typed.typeddict, nb.types.int32 ... appear in annotations only
        return _Namespace()

    def __call__(self, *a, **k):
        return _Namespace()


typed = _Namespace()
types = _Namespace()
