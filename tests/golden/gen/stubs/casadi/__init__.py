"""Numeric stand-in for the `casadi` Python module (golden-vector generation only).

Every ``SX.sym`` is a concrete 2-D array, so *building* the reference's optimisation
problem with this module evaluates it at the supplied point.  Semantics follow CasADi
where the reference relies on them: column-major ``X[:]``/``reshape``, single-index
slicing of vectors keeps their orientation, 1x1 broadcasting, comparisons -> 0/1,
``if_else`` as a numeric select.  dtype may be switched to complex for complex-step
derivatives.  Round 4: the pure operations also record how their value was computed (`rec`), so that an expression can be
RE-EVALUATED with some symbols bound to other values: that is what `ca.Function(name, inputs, outputs)(args)` and
`ca.jacobian(expr, x)` (complex step on the re-evaluation; scalar by scalar only) need in the reference's via-point NLP
(optimization_functions.py:227-387).  Matrices that are filled in place (`m[i, j] = ...`) stay leaves: their value at build time is
used as is.  This file is test tooling: it is never imported by the product.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from ca_tape import TapeFunction  # noqa: E402

inf = np.inf
pi = np.pi

DTYPE = [float]          # mutable: [float] or [complex]
SYM_LOG = []             # (index, name, shape)
PROVIDER = [None]        # callable(index, name, shape, k) -> ndarray or None
CAPTURED = {}            # last nlpsol/qpsol problem


def _arr(x):
    if isinstance(x, M):
        return x.a
    a = np.asarray(x, dtype=DTYPE[0] if not np.iscomplexobj(x) else complex)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(-1, 1)
    return a


def _bc(a, b):
    """1x1 broadcast only (CasADi semantics); otherwise shapes must agree."""
    if a.shape == b.shape or a.size == 1 or b.size == 1:
        return a, b
    raise ValueError(f"shape mismatch {a.shape} vs {b.shape}")


def _node(fn, *parents):
    """M with the value fn(parent arrays) that remembers fn and its parents (M objects or plain constants)"""
    m = M(fn(*[q.a if isinstance(q, M) else q for q in parents]))
    m.rec = (fn, parents)
    return m


def _eval(m, subs, memo):
    """value of m (array) with the M objects in `subs` (by id) bound to other arrays"""
    if not isinstance(m, M):
        return m
    if id(m) in subs:
        return subs[id(m)]
    if id(m) in memo:
        return memo[id(m)]
    rec = getattr(m, "rec", None)
    if rec is None:
        v = m.a
    elif rec[0] == "jac":                  # d expr / d x by complex step at the CURRENT binding of x
        _, expr, x = rec
        x0 = np.asarray(_eval(x, subs, memo), dtype=complex)
        s2 = dict(subs); s2[id(x)] = x0 + 1e-30j
        v = np.imag(np.asarray(_eval(expr, s2, {}))) / 1e-30
    else:
        fn, parents = rec
        v = fn(*[_eval(q, subs, memo) if isinstance(q, M) else q for q in parents])
    v = np.asarray(v)
    if v.ndim == 0:
        v = v.reshape(1, 1)
    memo[id(m)] = v
    return v


class M:
    __array_priority__ = 1000

    def __init__(self, a):
        a = np.array(a, dtype=complex if np.iscomplexobj(a) else DTYPE[0])
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        self.a = a

    # -- structure -------------------------------------------------------
    @property
    def shape(self):
        return self.a.shape

    @property
    def T(self):
        return _node(lambda a: a.T, self)

    def size1(self):
        return self.a.shape[0]

    def size2(self):
        return self.a.shape[1]

    def numel(self):
        return self.a.size

    def is_vector(self):
        return 1 in self.a.shape

    def reshape(self, shp):
        r, c = shp
        return _node(lambda a: a.reshape((r, c), order="F"), self)

    def full(self):
        return np.real(self.a).copy() if DTYPE[0] is float else self.a.copy()

    def __len__(self):
        return self.a.shape[0]

    def __float__(self):
        assert self.a.size == 1
        return float(np.real(self.a[0, 0]))

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return f"M({self.a!r})"

    # -- indexing --------------------------------------------------------
    def _single(self, k):
        n = self.a.size
        flat_idx = np.arange(n)[k]
        return flat_idx

    def __getitem__(self, k):
        if isinstance(k, tuple):
            i, j = k
            ii, jj = self._ax(i, 0), self._ax(j, 1)
            return _node(lambda a: a[ii][:, jj], self)
        # single index: column-major linear indexing
        row = self.a.shape[0] == 1 and self.a.shape[1] != 1

        def pick(a):
            sel = a.reshape(-1, order="F")[k]
            if np.ndim(sel) == 0:
                return np.asarray(sel).reshape(1, 1)
            return np.asarray(sel).reshape(1, -1) if row else np.asarray(sel).reshape(-1, 1)
        return _node(pick, self)

    def _ax(self, i, axis):
        n = self.a.shape[axis]
        if isinstance(i, slice):
            return np.arange(n)[i]
        if isinstance(i, M):
            i = int(np.real(i.a[0, 0]))
        return np.array([np.arange(n)[i]])

    def __setitem__(self, k, v):
        v = _arr(v)
        if isinstance(k, tuple):
            i, j = k
            ii, jj = self._ax(i, 0), self._ax(j, 1)
            tgt = (len(ii), len(jj))
            if v.size == 1:
                v = np.full(tgt, v[0, 0])
            elif v.shape != tgt:
                v = v.reshape(tgt, order="F")
            if np.iscomplexobj(v) and not np.iscomplexobj(self.a):
                self.a = self.a.astype(complex)
            self.a[np.ix_(ii, jj)] = v
            return
        flat_idx = np.arange(self.a.size)[k]
        flat_idx = np.atleast_1d(flat_idx)
        vals = v.reshape(-1, order="F")
        if vals.size == 1:
            vals = np.full(flat_idx.size, vals[0])
        assert vals.size == flat_idx.size
        if np.iscomplexobj(vals) and not np.iscomplexobj(self.a):
            self.a = self.a.astype(complex)
        r = flat_idx % self.a.shape[0]
        c = flat_idx // self.a.shape[0]
        self.a[r, c] = vals

    # -- arithmetic ------------------------------------------------------
    def _bin(self, o, f, rev=False):
        _bc(self.a, _arr(o))              # shape check
        other = o if isinstance(o, M) else _arr(o)
        return _node((lambda a, b: f(b, a)) if rev else (lambda a, b: f(a, b)), self, other)

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __pow__(self, o): return self._bin(o, np.power)
    def __neg__(self): return _node(lambda a: -a, self)
    def __pos__(self): return self

    def __matmul__(self, o):
        return _node(_mm, self, o if isinstance(o, M) else _arr(o))

    def __rmatmul__(self, o):
        return _node(_mm, o if isinstance(o, M) else _arr(o), self)

    # comparisons act on real parts and give 0/1
    def _cmp(self, o, f):
        a, b = _bc(np.real(self.a), np.real(_arr(o)))
        return M(f(a, b).astype(float))

    def __lt__(self, o): return self._cmp(o, np.less)
    def __le__(self, o): return self._cmp(o, np.less_equal)
    def __gt__(self, o): return self._cmp(o, np.greater)
    def __ge__(self, o): return self._cmp(o, np.greater_equal)
    def __eq__(self, o): return self._cmp(o, np.equal)
    def __ne__(self, o): return self._cmp(o, np.not_equal)

    def __bool__(self):
        assert self.a.size == 1
        return bool(np.real(self.a[0, 0]) != 0)


class _SymFactory:
    """SX / MX namespace."""

    @staticmethod
    def sym(name, n=1, m=1, k=None):
        def one(kk):
            idx = len(SYM_LOG)
            SYM_LOG.append((idx, name, (n, m), kk))
            val = None
            if PROVIDER[0] is not None:
                val = PROVIDER[0](idx, name, (n, m), kk)
            if val is None:
                val = np.zeros((n, m))
            val = np.asarray(val)
            assert val.shape == (n, m), (name, val.shape, (n, m))
            return M(val)

        if k is None:
            return one(None)
        return [one(kk) for kk in range(k)]

    @staticmethod
    def zeros(*shape):
        if len(shape) == 1 and isinstance(shape[0], tuple):
            shape = shape[0]
        if len(shape) == 1:
            shape = (shape[0], 1)
        return M(np.zeros(shape, dtype=DTYPE[0]))

    @staticmethod
    def eye(n):
        return M(np.eye(n, dtype=DTYPE[0]))


SX = _SymFactory
MX = _SymFactory


class DM(M):
    pass


def _mm(a, b):
    return a * b if (a.size == 1 or b.size == 1) else a @ b


def _lift(x):
    return x if isinstance(x, M) else _arr(x)


def vertcat(*args):
    parts = [_lift(a) for a in args if _arr(a).size > 0]
    return _node(lambda *p: np.vstack(p), *parts)


def horzcat(*args):
    parts = [_lift(a) for a in args if _arr(a).size > 0]
    return _node(lambda *p: np.hstack(p), *parts)


def sumsqr(x):
    return _node(lambda a: np.sum(a * a), _lift(x))          # NOT |a|^2: analytic for complex step


def dot(x, y):
    return _node(lambda a, b: np.sum(a.reshape(-1, order="F") * b.reshape(-1, order="F")), _lift(x), _lift(y))


def norm_2(x):
    return _node(lambda a: np.sqrt(np.sum(a * a)), _lift(x))


def _elem(fn, x):
    # like CasADi on plain numbers: numeric in, numeric out (the host-side numpy code of the reference
    # calls ca.sin / ca.cos on floats, optimization_functions.py:101-103)
    if isinstance(x, M):
        return _node(fn, x)
    return fn(np.asarray(x, dtype=float)) if np.ndim(x) else float(fn(x))


def exp(x): return _elem(np.exp, x)
def sqrt(x): return _elem(np.sqrt, x)
def sin(x): return _elem(np.sin, x)
def cos(x): return _elem(np.cos, x)


def if_else(c, a, b):
    cv = bool(np.real(_arr(c)).reshape(-1)[0] != 0)
    return a if cv else b


def jacobian(expr, x, *a, **k):
    """d expr / d x for a scalar expression and a scalar symbol: complex step on the re-evaluated expression (see _eval)."""
    if not (isinstance(expr, M) and isinstance(x, M) and expr.a.size == 1 and x.a.size == 1):
        return None
    m = M(np.zeros((1, 1)))
    m.rec = ("jac", expr, x)
    m.a = np.asarray(_eval(m, {}, {}), dtype=DTYPE[0] if DTYPE[0] is float else complex)
    return m


class _TapeWrap:
    def __init__(self, path, shape):
        self.f = TapeFunction(path, shape)

    def __call__(self, q):
        return M(self.f(_arr(q).reshape(-1, order="F")))


class Function:
    _shapes = {"fk_pos": (3, 1), "hom_trans": (4, 4), "jacobian": (6, 7)}

    def __init__(self, *a, **k):
        self.args = a
        # Function(name, [input symbols], [output expressions]): callable by re-evaluation
        self.inputs = list(a[1]) if len(a) >= 3 and isinstance(a[1], (list, tuple)) else None
        self.outputs = list(a[2]) if len(a) >= 3 and isinstance(a[2], (list, tuple)) else None

    def __call__(self, *a, **k):
        if self.inputs is None or len(a) != len(self.inputs) or not all(isinstance(i, M) for i in self.inputs):
            raise RuntimeError("this Function call is not supported by the numeric shim")
        subs = {id(sym): np.asarray(_arr(val)) for sym, val in zip(self.inputs, a)}
        outs = [M(_eval(o, subs, {})) for o in self.outputs]
        return outs[0] if len(outs) == 1 else outs

    @staticmethod
    def load(path):
        base = os.path.basename(path).replace(".ca", "")
        shape = Function._shapes.get(base, (3, 1))
        return _TapeWrap(path, shape)


class _Solver:
    def __init__(self, kind, name, plugin, prob, opts):
        self.kind, self.prob, self.opts = kind, prob, opts
        CAPTURED[kind] = prob
        CAPTURED["last"] = prob

    def __call__(self, **kw):
        raise RuntimeError("the numeric shim cannot solve; it only captures the problem")

    def stats(self):
        return {}

    def generate_dependencies(self, *a, **k):
        pass


def nlpsol(name, plugin, prob, opts=None):
    return _Solver("nlpsol", name, plugin, prob, opts)


def qpsol(name, plugin, prob, opts=None):
    return _Solver("qpsol", name, plugin, prob, opts)
