"""NIST StRD problems of the reference's test/nonlinearfitting.jl as Python callables, built from the committed data
fixture tests/golden/nist.json (made by tests/golden/make_nist.py).  Residuals follow the reference's ff!
(nonlinearfitting.jl:1448-1452): fcur[i] = y[i] - model(x[i], beta).  The analytic Jacobian is derived from the
model expression with sympy; the finite-difference one is the host-side central difference of the API mirror."""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    return json.load(open(os.path.join(_HERE, "golden", "nist.json")))


class Problem:
    def __init__(self, rec):
        import sympy as sy
        self.name = rec["name"]
        self.y = np.array(rec["y"], dtype=np.float64)
        self.x = np.array(rec["x"], dtype=np.float64)
        self.starts = [np.array(s, dtype=np.float64) for s in rec["starts"]]
        self.certified = np.array(rec["certified"], dtype=np.float64)
        self.m, self.n = len(self.y), len(self.certified)
        xs = sy.Symbol("x")
        bs = sy.symbols("b0:%d" % self.n)
        expr = eval(rec["model"], {"exp": sy.exp, "x": xs, "b": bs})
        self._model = sy.lambdify((xs,) + bs, expr, "numpy")
        self._grads = [sy.lambdify((xs,) + bs, sy.diff(expr, bk), "numpy") for bk in bs]

    def f(self, out, beta):
        """f!(fcur, beta)"""
        with np.errstate(all="ignore"):
            out[:] = self.y - self._model(self.x, *beta)

    def g(self, J, beta):
        """g!(J, beta) on an (m, n) array view (analytic)"""
        with np.errstate(all="ignore"):
            for k, gk in enumerate(self._grads):
                J[:, k] = -np.broadcast_to(gk(self.x, *beta), self.x.shape)

    def g_flat(self, Jval, beta):
        """g! on the flat column-major buffer the oracle hands out"""
        self.g(Jval.reshape((self.m, self.n), order="F"), beta)


def central_difference_g(f, m, n):
    """FiniteDiff-style central differences (the reference's default Jacobian, types.jl:55-58); the same formula as
    leastsquaresoptim.jl_amd/api.py:_central_difference_jacobian, restated here so the oracle can be driven with it."""
    eps3 = np.finfo(float).eps ** (1.0 / 3.0)

    def g_flat(Jval, x):
        J = Jval.reshape((m, n), order="F")
        fp, fm = np.zeros(m), np.zeros(m)
        xp = np.array(x, dtype=np.float64)
        for j in range(n):
            h = max(eps3 * abs(x[j]), eps3)
            xj = xp[j]
            xp[j] = xj + h
            f(fp, xp)
            xp[j] = xj - h
            f(fm, xp)
            xp[j] = xj
            J[:, j] = (fp - fm) / (2 * h)

    return g_flat


def problems():
    return [Problem(r) for r in load()["problems"]]
