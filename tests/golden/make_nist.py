#!/usr/bin/env python3
"""Generates tests/golden/nist.json: the 16 NIST StRD nonlinear-regression problems the reference's own test
suite holds (/root/reference/test/nonlinearfitting.jl:6-1445; problem list :1455; tolerances :1465) -- DATA ONLY:
observations, the starting points (every column of `parameters`), and the CERTIFIED parameter values.  These certified values are the
one set of reference-held numbers that pins results beyond `ssr <= 1e-3`.

Run in the build container (where /root/reference exists); the JSON is committed and is all that travels.

Each model's formula is restated here as a Python expression (`model` key, evaluated with numpy names on the
vector x of regressor values and the parameter vector b, 0-based); the analytic Jacobians the tests need are
derived symbolically from that expression by tests/nist.py (sympy), not taken from the reference, which
differentiates numerically (FiniteDiff central differences, types.jl:55-58).

Column convention of the reference's `data` matrices: column 1 = observed y, column 2 = regressor x
(`fcur[i] = data[i, 1] - f(data[i, 2], x)`, nonlinearfitting.jl:1448-1452).
"""
import json
import os
import re

SRC = "/root/reference/test/nonlinearfitting.jl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nist.json")

# name of the Julia function in the reference -> (reported name, line of the formula, restated model)
MODELS = {
    "misra1a": ("Misra1a", 28, "b[0]*(1-exp(-b[1]*x))"),
    "Chwirut2": ("Chwirut2", 98, "exp(-b[0]*x)/(b[1]+b[2]*x)"),
    "Chwitrut1": ("Chwirut1", 327, "exp(-b[0]*x)/(b[1]+b[2]*x)"),
    "Lanczos3": ("Lanczos3", 369, "b[0]*exp(-b[1]*x) + b[2]*exp(-b[3]*x) + b[4]*exp(-b[5]*x)"),
    "Gauss1": ("Gauss1", 658, "b[0]*exp(-b[1]*x) + b[2]*exp(-(x-b[3])**2/b[4]**2) + b[5]*exp(-(x-b[6])**2/b[7]**2)"),
    "Gauss2": ("Gauss2", 928, "b[0]*exp(-b[1]*x) + b[2]*exp(-(x-b[3])**2/b[4]**2) + b[5]*exp(-(x-b[6])**2/b[7]**2)"),
    "DanWood": ("DanWood", 958, "b[0]*x**b[1]"),
    "Misra1b": ("Misra1b", 989, "b[0]*(1-(1/(1+b[1]*x/2)**2))"),
    "MGH09": ("MGH09", 1019, "b[0]*(x**2+x*b[1])/(x**2+x*b[2]+b[3])"),
    "Thurber": ("Thurber", 1081, "(b[0]+b[1]*x+b[2]*x**2+b[3]*x**3)/(1+b[4]*x+b[5]*x**2+b[6]*x**3)"),
    "BoxBOD": ("BoxBOD", 1112, "b[0]*(1-exp(-b[1]*x))"),
    "Rat42": ("Rat42", 1140, "b[0]/(1+exp(b[1]-b[2]*x))"),
    "MGH10": ("MGH10", 1173, "b[0]*exp(b[1]/(x+b[2]))"),
    "Eckerle4": ("Eckerle4", 1227, "b[0]/b[1]*exp(-(x-b[2])**2/(2*b[1]**2))"),
    "Rat43": ("Rat43", 1263, "b[0]/(1+exp(b[1]-b[2]*x))**(1/b[3])"),
    "Bennett5": ("Bennett5", 1442, "b[0]*(b[1]+x)**(-1/b[2])"),
}
ORDER = ["misra1a", "Chwirut2", "Chwitrut1", "Lanczos3", "Gauss1", "Gauss2", "DanWood", "Misra1b", "MGH09", "Thurber",
         "BoxBOD", "Rat42", "MGH10", "Eckerle4", "Rat43", "Bennett5"]          # nonlinearfitting.jl:1455

NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def matrix(block, key):
    """`key = [ ... ]` as a list of rows (rows end at `;` or at a newline)."""
    m = re.search(r"\b%s\s*=\s*\[(.*?)\]" % key, block, flags=re.S)
    assert m, key
    rows = []
    for line in re.split(r"[;\n]", m.group(1)):
        vals = re.findall(NUM, line.split("#")[0])
        if vals:
            rows.append([float(v) for v in vals])
    return rows


def main():
    txt = open(SRC).read()
    out = {"source": "test/nonlinearfitting.jl of matthieugomez/LeastSquaresOptim.jl (NIST StRD nonlinear regression, "
                     "https://www.itl.nist.gov/div898/strd/nls/nls_main.shtml)",
           "reference_settings": {"optimizers": ["Dogleg(QR())", "LevenbergMarquardt(QR())"], "x_tol": 1e-50,
                                  "f_tol": 1e-36, "g_tol": 1e-50, "success": "norm(minimizer - certified) <= 1e-3",
                                  "asserted_by_reference": "!isnan(mean(minimizer)); the success count is printed, "
                                                           "not asserted (nonlinearfitting.jl:1465-1470)"},
           "problems": []}
    for fn in ORDER:
        a = txt.index("function %s()" % fn)
        b = txt.index("\nend", a)
        block = txt[a:b]
        name, line, model = MODELS[fn]
        data = matrix(block, "data")
        par = matrix(block, "parameters")
        sol = [v for row in matrix(block, "solution") for v in row]
        p = len(par)
        assert all(len(r) == 2 for r in data), (fn, [r for r in data if len(r) != 2][:3])
        ncol = len(par[0])     # 2 or 3 columns: the NIST starts, for some problems followed by the certified values
        assert all(len(r) == ncol for r in par) and len(sol) == p, (fn, par, sol)
        out["problems"].append({
            "name": name, "reference_function": fn, "formula_at": "test/nonlinearfitting.jl:%d" % line, "model": model,
            "y": [r[0] for r in data], "x": [r[1] for r in data],
            "starts": [[r[j] for r in par] for j in range(ncol)],   # every column of `parameters` (:1462)
            "certified": sol})
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", OUT, "with", len(out["problems"]), "problems;", sum(len(p["y"]) for p in out["problems"]), "observations")


if __name__ == "__main__":
    main()
