"""Test problems restated from the reference's test-suite (problem DEFINITIONS, i.e. data):

* the 14 MINPACK hybrj functions / 21 instances  (test/nonlinearsolvers.jl:7-501, list :512-522)
* the rank-deficient 9x6 factor model            (test/nonlinearleastsquares.jl:7-95)
* the three box-constraint problems              (test/bounds.jl:7-37)
* README Rosenbrock                              (README.md:67-80, test/runtests.jl:19-41)

Each problem is (name, f(out, x), g(J, x), x0) with J an (m, n) Fortran-ordered numpy view that
g fills in place -- the calling convention of the reference's f!/g!.
"""
import math

import numpy as np


def rosenbrock():
    def f(o, x):
        o[0] = 1 - x[0]
        o[1] = 10 * (x[1] - x[0] ** 2)

    def g(J, x):
        J[0, 0] = -1
        J[0, 1] = 0
        J[1, 0] = -20 * x[0]
        J[1, 1] = 10

    return "rosenbrock", f, g, np.array([-1.2, 1.0])


def powell_singular():
    s5, s10 = math.sqrt(5), math.sqrt(10)

    def f(o, x):
        o[0] = x[0] + 10 * x[1]
        o[1] = s5 * (x[2] - x[3])
        o[2] = (x[1] - 2 * x[2]) ** 2
        o[3] = s10 * (x[0] - x[3]) ** 2

    def g(J, x):
        J[:] = 0
        J[0, 0] = 1
        J[0, 1] = 10
        J[1, 2] = s5
        J[1, 3] = -s5
        J[2, 1] = 2 * (x[1] - 2 * x[2])
        J[2, 2] = -2 * J[2, 1]
        J[3, 0] = 2 * s10 * (x[0] - x[3])
        J[3, 3] = -J[3, 0]

    return "powell_singular", f, g, np.array([3.0, -1.0, 0.0, 1.0])


def powell_badly_scaled():
    c1, c2 = 1e4, 1.0001

    def f(o, x):
        o[0] = c1 * x[0] * x[1] - 1
        o[1] = math.exp(-x[0]) + math.exp(-x[1]) - c2

    def g(J, x):
        J[0, 0] = c1 * x[1]
        J[0, 1] = c1 * x[0]
        J[1, 0] = -math.exp(-x[0])
        J[1, 1] = -math.exp(-x[1])

    return "powell_badly_scaled", f, g, np.array([0.0, 1.0])


def wood():
    c3, c4, c5, c6 = 2e2, 2.02e1, 1.98e1, 1.8e2

    def f(o, x):
        t1 = x[1] - x[0] ** 2
        t2 = x[3] - x[2] ** 2
        o[0] = -c3 * x[0] * t1 - (1 - x[0])
        o[1] = c3 * t1 + c4 * (x[1] - 1) + c5 * (x[3] - 1)
        o[2] = -c6 * x[2] * t2 - (1 - x[2])
        o[3] = c6 * t2 + c4 * (x[3] - 1) + c5 * (x[1] - 1)

    def g(J, x):
        J[:] = 0
        t1 = x[1] - 3 * x[0] ** 2
        t2 = x[3] - 3 * x[2] ** 2
        J[0, 0] = -c3 * t1 + 1
        J[0, 1] = -c3 * x[0]
        J[1, 0] = -2 * c3 * x[0]
        J[1, 1] = c3 + c4
        J[1, 3] = c5
        J[2, 2] = -c6 * t2 + 1
        J[2, 3] = -c6 * x[2]
        J[3, 1] = c5
        J[3, 2] = -2 * c6 * x[2]
        J[3, 3] = c6 + c4

    return "wood", f, g, np.array([-3.0, -1.0, -3.0, -1.0])


def helical_valley():
    tpi = 8 * math.atan(1)

    def f(o, x):
        if x[0] > 0:
            t1 = math.atan(x[1] / x[0]) / tpi
        elif x[0] < 0:
            t1 = math.atan(x[1] / x[0]) / tpi + 0.5
        else:
            t1 = 0.25 * np.sign(x[1])
        t2 = math.sqrt(x[0] ** 2 + x[1] ** 2)
        o[0] = 10 * (x[2] - 10 * t1)
        o[1] = 10 * (t2 - 1)
        o[2] = x[2]

    def g(J, x):
        t = x[0] ** 2 + x[1] ** 2
        t1 = tpi * t
        t2 = math.sqrt(t)
        J[0, 0] = 100 * x[1] / t1
        J[0, 1] = -100 * x[0] / t1
        J[0, 2] = 10
        J[1, 0] = 10 * x[0] / t2
        J[1, 1] = 10 * x[1] / t2
        J[1, 2] = 0
        J[2, 0] = 0
        J[2, 1] = 0
        J[2, 2] = 1

    return "helical_valley", f, g, np.array([-1.0, 0.0, 0.0])


def watson(n):
    def f(o, x):
        o[:] = 0
        for i in range(1, 30):
            ti = i / 29.0
            s1, t = 0.0, 1.0
            for j in range(1, n):
                s1 += j * t * x[j]
                t *= ti
            s2, t = 0.0, 1.0
            for j in range(n):
                s2 += t * x[j]
                t *= ti
            t1 = s1 - s2 ** 2 - 1
            t2 = 2 * ti * s2
            t = 1 / ti
            for k in range(n):
                o[k] += t * (k - t2) * t1
                t *= ti
        t = x[1] - x[0] ** 2 - 1
        o[0] += x[0] * (1 - 2 * t)
        o[1] += t

    def g(J, x):
        J[:] = 0
        for i in range(1, 30):
            ti = i / 29.0
            s1, t = 0.0, 1.0
            for j in range(1, n):
                s1 += j * t * x[j]
                t *= ti
            s2, t = 0.0, 1.0
            for j in range(n):
                s2 += t * x[j]
                t *= ti
            t1 = 2 * (s1 - s2 ** 2 - 1)
            t2 = 2 * s2
            t = ti ** 2
            tk = 1.0
            for k in range(n):
                tj = tk
                for j in range(k, n):
                    J[k, j] += tj * ((k / ti - t2) * (j / ti - t2) - t1)
                    tj *= ti
                tk *= t
        J[0, 0] += 6 * x[0] ** 2 - 2 * x[1] + 3
        J[0, 1] -= 2 * x[0]
        J[1, 1] += 1
        for k in range(n):
            for j in range(k, n):
                J[j, k] = J[k, j]

    return "watson", f, g, np.zeros(n)


def chebyquad(n):
    tk = 1.0 / n

    def f(o, x):
        o[:] = 0
        for j in range(n):
            t1 = 1.0
            t2 = 2 * x[j] - 1
            t = 2 * t2
            for i in range(n):
                o[i] += t2
                ti = t * t2 - t1
                t1 = t2
                t2 = ti
        iev = -1.0
        for k in range(n):
            o[k] *= tk
            if iev > 0:
                o[k] += 1.0 / ((k + 1) ** 2 - 1)
            iev = -iev

    def g(J, x):
        for j in range(n):
            t1 = 1.0
            t2 = 2 * x[j] - 1
            t = 2 * t2
            t3 = 0.0
            t4 = 2.0
            for k in range(n):
                J[k, j] = tk * t4
                ti = 4 * t2 + t * t4 - t3
                t3 = t4
                t4 = ti
                ti = t * t2 - t1
                t1 = t2
                t2 = ti

    return "chebyquad", f, g, np.arange(1, n + 1) / (n + 1.0)


def brown_almost_linear(n):
    def f(o, x):
        s = np.sum(x) - (n + 1)
        for k in range(n - 1):
            o[k] = x[k] + s
        o[n - 1] = np.prod(x) - 1

    def g(J, x):
        J[:] = 1
        for k in range(n):
            J[k, k] = 2
        prd = np.prod(x)
        for j in range(n):
            if x[j] == 0.0:
                v = 1.0
                for k in range(n):
                    if k != j:
                        v *= x[k]
                J[n - 1, j] = v
            else:
                J[n - 1, j] = prd / x[j]

    return "brown_almost_linear", f, g, 0.5 * np.ones(n)


def discrete_boundary_value(n):
    h = 1.0 / (n + 1)

    def f(o, x):
        for k in range(n):
            t = (x[k] + (k + 1) * h + 1) ** 3
            t1 = x[k - 1] if k != 0 else 0.0
            t2 = x[k + 1] if k != n - 1 else 0.0
            o[k] = 2 * x[k] - t1 - t2 + t * h ** 2 / 2

    def g(J, x):
        for k in range(n):
            t = 3 * (x[k] + (k + 1) * h + 1) ** 2
            J[k, :] = 0
            J[k, k] = 2 + t * h ** 2 / 2
            if k != 0:
                J[k, k - 1] = -1
            if k != n - 1:
                J[k, k + 1] = -1

    x = np.arange(1, n + 1) * h
    return "discrete_boundary_value", f, g, x * (x - 1)


def discrete_integral_equation(n):
    h = 1.0 / (n + 1)

    def f(o, x):
        for k in range(n):
            tk = (k + 1) * h
            s1 = 0.0
            for j in range(k + 1):
                tj = (j + 1) * h
                s1 += tj * (x[j] + tj + 1) ** 3
            s2 = 0.0
            for j in range(k + 1, n):
                tj = (j + 1) * h
                s2 += (1 - tj) * (x[j] + tj + 1) ** 3
            o[k] = x[k] + h * ((1 - tk) * s1 + tk * s2) / 2

    def g(J, x):
        for k in range(n):
            tk = (k + 1) * h
            for j in range(n):
                tj = (j + 1) * h
                J[k, j] = h * min(tj * (1 - tk), tk * (1 - tj)) * 3 * (x[j] + tj + 1) ** 2 / 2
            J[k, k] += 1

    x = np.arange(1, n + 1) * h
    return "discrete_integral_equation", f, g, x * (x - 1)


def trigonometric(n):
    def f(o, x):
        for j in range(n):
            o[j] = math.cos(x[j])
        s = np.sum(o)
        for k in range(n):
            o[k] = n + (k + 1) - math.sin(x[k]) - s - (k + 1) * o[k]

    def g(J, x):
        for j in range(n):
            t = math.sin(x[j])
            J[:, j] = t
            J[j, j] = (j + 2) * t - math.cos(x[j])

    return "trigonometric", f, g, np.ones(n) / n


def variably_dimensioned(n):
    def f(o, x):
        s = 0.0
        for j in range(n):
            s += (j + 1) * (x[j] - 1)
        t = s * (1 + 2 * s ** 2)
        for k in range(n):
            o[k] = x[k] - 1 + (k + 1) * t

    def g(J, x):
        s = 0.0
        for j in range(n):
            s += (j + 1) * (x[j] - 1)
        t = 1 + 6 * s ** 2
        for k in range(n):
            for j in range(k, n):
                J[k, j] = (k + 1) * (j + 1) * t
                J[j, k] = J[k, j]
            J[k, k] += 1

    return "variably_dimensioned", f, g, np.arange(1, n + 1) / float(n)


def broyden_tridiagonal(n):
    def f(o, x):
        for k in range(n):
            t = (3 - 2 * x[k]) * x[k]
            t1 = x[k - 1] if k != 0 else 0.0
            t2 = x[k + 1] if k != n - 1 else 0.0
            o[k] = t - t1 - 2 * t2 + 1

    def g(J, x):
        J[:] = 0
        for k in range(n):
            J[k, k] = 3 - 4 * x[k]
            if k != 0:
                J[k, k - 1] = -1
            if k != n - 1:
                J[k, k + 1] = -2

    return "broyden_tridiagonal", f, g, -np.ones(n)


def broyden_banded(n):
    ml, mu = 5, 1

    def f(o, x):
        for k in range(n):
            k1 = max(0, k - ml)
            k2 = min(k + mu, n - 1)
            t = 0.0
            for j in range(k1, k2 + 1):
                if j != k:
                    t += x[j] * (1 + x[j])
            o[k] = x[k] * (2 + 5 * x[k] ** 2) + 1 - t

    def g(J, x):
        J[:] = 0
        for k in range(n):
            k1 = max(0, k - ml)
            k2 = min(k + mu, n - 1)
            for j in range(k1, k2 + 1):
                if j != k:
                    J[k, j] = -(1 + 2 * x[j])
            J[k, k] = 2 + 15 * x[k] ** 2

    return "broyden_banded", f, g, -np.ones(n)


def minpack_all():
    """The 21 instances of test/nonlinearsolvers.jl:512-522, in that order."""
    return [rosenbrock(), powell_singular(), powell_badly_scaled(), wood(), helical_valley(),
            watson(6), watson(9), chebyquad(5), chebyquad(6), chebyquad(7), chebyquad(9),
            brown_almost_linear(10), brown_almost_linear(30), brown_almost_linear(40),
            discrete_boundary_value(10), discrete_integral_equation(1),
            discrete_integral_equation(10), trigonometric(10), variably_dimensioned(10),
            broyden_tridiagonal(10), broyden_banded(10)]


def minpack_cholesky():
    """The 18 instances of the dense-Cholesky block, test/nonlinearsolvers.jl:573-583."""
    return [rosenbrock(), powell_singular(), powell_badly_scaled(), wood(), helical_valley(),
            watson(6), chebyquad(5), chebyquad(6), chebyquad(7), chebyquad(9),
            brown_almost_linear(10), discrete_boundary_value(10), discrete_integral_equation(1),
            discrete_integral_equation(10), trigonometric(10), variably_dimensioned(10),
            broyden_tridiagonal(10), broyden_banded(10)]


def label(p):
    return "%s(%d)" % (p[0], len(p[3]))


# --- rank-deficient factor model (test/nonlinearleastsquares.jl:7-95) -------------------------
FACTOR_TARGETS = [3.0, 2.0, 5.0, 4.5, 3.2, 2.0, 5.0, 1.3, 1.5]


def factor_dense():
    def f(o, x):
        k = 0
        for a in range(3):
            for b in range(3):
                o[k] = FACTOR_TARGETS[k] - x[a] * x[3 + b]
                k += 1

    def g(J, x):
        J[:] = 0
        k = 0
        for a in range(3):
            for b in range(3):
                J[k, a] = -x[3 + b]
                J[k, 3 + b] = -x[a]
                k += 1

    return "factor", f, g, np.ones(6)


def factor_sparse():
    """Fixed 18-entry pattern; g writes nzval in CSC order (nonlinearleastsquares.jl:47-86)."""
    name, f, gd, x0 = factor_dense()
    pattern = np.zeros((9, 6), dtype=bool)
    k = 0
    for a in range(3):
        for b in range(3):
            pattern[k, a] = True
            pattern[k, 3 + b] = True
            k += 1
    rows, cols = np.nonzero(pattern.T)  # iterate column-major
    colidx, rowidx = rows, cols
    colptr = np.zeros(7, dtype=np.int32)
    for c in colidx:
        colptr[c + 1] += 1
    colptr = np.cumsum(colptr).astype(np.int32)
    rowval = rowidx.astype(np.int32)

    def g(nz, x):
        J = np.zeros((9, 6))
        gd(J, x)
        nz[:] = J[rowidx, colidx]

    return name, f, g, x0, (9, 6, colptr, rowval)


# --- box-constraint problems (test/bounds.jl) -------------------------------------------------
def readme_rosenbrock():
    def f(o, x):
        o[0] = 1 - x[0]
        o[1] = 100 * (x[1] - x[0] ** 2)

    def g(J, x):
        J[0, 0] = -1
        J[0, 1] = 0
        J[1, 0] = -200 * x[0]
        J[1, 1] = 100

    return "readme_rosenbrock", f, g, np.zeros(2)


def bound_lower_active():
    def f(o, x):
        o[0] = x[0] - 0.5
        o[1] = x[1] ** 2 - 9

    def g(J, x):
        J[0, 0] = 1
        J[0, 1] = 0
        J[1, 0] = 0
        J[1, 1] = 2 * x[1]

    return "bound_lower", f, g, np.array([2.0, 1.0])


def bound_upper_active():
    def f(o, x):
        o[0] = x[0] - 5
        o[1] = x[1] ** 2 - 4

    def g(J, x):
        J[0, 0] = 1
        J[0, 1] = 0
        J[1, 0] = 0
        J[1, 1] = 2 * x[1]

    return "bound_upper", f, g, np.array([0.0, 1.0])


def full_csc_pattern(m, n):
    """Every entry structural (what `sparse(ones(n, n))` gives): nzval == column-major dense."""
    colptr = (np.arange(n + 1) * m).astype(np.int32)
    rowval = np.tile(np.arange(m, dtype=np.int32), n)
    return m, n, colptr, rowval


def wrap_dense(f, g, m, n):
    """Adapt g(J2d, x) to the flat column-major buffer handed over by the C side."""

    def g_flat(jflat, x):
        g(jflat.reshape((m, n), order="F"), x)

    return f, g_flat
