"""Import-only stand-in for cvxpy (golden-vector generation)."""
CLARABEL = "CLARABEL"


class _Any:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, n):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


Variable = Parameter = SOC = Problem = Minimize = Maximize = _Any
