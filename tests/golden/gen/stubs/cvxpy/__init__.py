"""Import-only stand-in for cvxpy (golden-vector generation): every expression is an inert object, so the reference's
problem-construction code (ConvexSetFinder.cvx_mvie_socp*, executed in its constructor) runs without effect."""
CLARABEL = "CLARABEL"


class _Any:
    __array_priority__ = 1000.0      # numpy defers its binary operators to this class
    __array_ufunc__ = None

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, n):
        if n.startswith("__") and n.endswith("__"):
            raise AttributeError(n)
        return _Any()

    def __call__(self, *a, **k):
        return _Any()

    def __getitem__(self, i):
        return _Any()

    def _op(self, *a, **k):
        return _Any()

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __matmul__ = __rmatmul__ = __neg__ = _op
    __truediv__ = __rtruediv__ = __pow__ = __le__ = __ge__ = __lt__ = __gt__ = __eq__ = _op
    __hash__ = object.__hash__


class _Sym(_Any):
    """Variable / Parameter: keeps the declared shape and an assignable value."""

    def __init__(self, shape=(), *a, **k):
        object.__setattr__(self, "shape", (shape,) if isinstance(shape, int) else tuple(shape))
        object.__setattr__(self, "value", None)


Variable = Parameter = _Sym
SOC = Problem = Minimize = Maximize = _Any


def __getattr__(name):      # any other cvxpy function (norm, hstack, ...)
    if name.startswith("__"):
        raise AttributeError(name)
    return _Any()
