#!/usr/bin/env python
"""An INDEPENDENT second restatement of sdlp::linprog<d> (Seidel's LP as shipped in the reference) — written straight from
the reference's text in plain Python floats WITHOUT reading oracle/ — whose optimum, optimal point and status conventions
on seeded problems are committed as tests/golden/lp_independent.json; the CPU tests hold the C++ oracle
(oracle/lp_oracle.cpp, `orc_linprog_perm`) to them (tests/test_lp_independent.py).  It does not pin the oracle to the
REFERENCE (Eigen absent), it makes two separately written readings agree.

Restated, block by block:  plan_manager/include/sfc_gen/sdlp.hpp (= traj_utils/include/traj_utils/sdlp.hpp)
  dot2, cross2, unit2, unit<d>       :50-94          lp_no_con<d>          :97-131
  move_to_front                      :134-152        lp_min_lin_rat        :154-258
  wedge                              :260-370        lp_base_case          :373-441
  findimax, vector_up, vector_down, plane_down  :444-516
  linfracprog<d>, linfracprog<1>     :518-680        linprog<d>            :704-787
The random permutation (rand_permutation :682-702, std::mt19937_64 + libstdc++'s uniform_int_distribution) is an INPUT
here: the fixture stores the permutation of every problem.  Where the text leaves a summation order to Eigen (the column
norms of `halves.colwise().normalize()`, `c.dot(x)`) it is written left to right.
Run from the repo root:   python tests/golden/make_lp_fixture.py
"""
import json
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EPS = 1.0e-12
MINIMUM, INFEASIBLE, UNBOUNDED, AMBIGUOUS = 0, 1, 2, 3
INF = float("inf")


def dot2(a, b):
    return a[0] * b[0] + a[1] * b[1]


def cross2(a, b):
    return a[0] * b[1] - a[1] * b[0]


def unit2(a, b):
    """writes the unit vector of a into b; True if a is (near) zero"""
    mag = math.sqrt(a[0] * a[0] + a[1] * a[1])
    if mag < 2.0 * EPS:
        return True
    b0 = a[0] / mag
    b1 = a[1] / mag
    b[0] = b0
    b[1] = b1
    return False


def unit(d, a):
    mag = 0.0
    for i in range(d + 1):
        mag += a[i] * a[i]
    if mag < (d + 1) * EPS * EPS:
        return True
    mag = 1.0 / math.sqrt(mag)
    for i in range(d + 1):
        a[i] *= mag
    return False


def lp_no_con(d, n_vec, d_vec, opt):
    n_dot_d = 0.0
    d_dot_d = 0.0
    for i in range(d + 1):
        n_dot_d += n_vec[i] * d_vec[i]
        d_dot_d += d_vec[i] * d_vec[i]
    if d_dot_d < EPS * EPS:
        n_dot_d = 0.0
        d_dot_d = 1.0
    for i in range(d + 1):
        opt[i] = -n_vec[i] + d_vec[i] * n_dot_d / d_dot_d
    if unit(d, opt):
        opt[d] = 1.0
        return AMBIGUOUS
    return MINIMUM


def move_to_front(i, nxt, prv):
    if i == 0 or i == nxt[0]:
        return i
    previ = prv[i]
    nxt[prv[i]] = nxt[i]
    prv[nxt[i]] = prv[i]
    nxt[i] = nxt[0]
    prv[i] = 0
    prv[nxt[i]] = i
    nxt[0] = i
    return previ


def fdiv(a, b):
    """C's double division: inf / nan instead of an exception"""
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0.0 or a != a:
            return float("nan")
        return math.copysign(INF, a) * math.copysign(1.0, b)


def wedge(H, m, nxt, prv, cw, ccw):
    """H[i] = half line i (2 numbers); returns (status, degen)"""
    degen = False
    i = 0
    while i != m:
        if not unit2(H[i], ccw):
            cw[0] = ccw[1]
            cw[1] = -ccw[0]
            ccw[0] = -cw[0]
            ccw[1] = -cw[1]
            break
        i = nxt[i]
    if i == m:
        return UNBOUNDED, degen
    i = 0
    while i != m:
        offensive = False
        d_cw = dot2(cw, H[i])
        d_ccw = dot2(ccw, H[i])
        if d_ccw >= 2.0 * EPS:
            if d_cw <= -2.0 * EPS:
                cw[0] = H[i][1]
                cw[1] = -H[i][0]
                unit2(cw, cw)
                offensive = True
        elif d_cw >= 2.0 * EPS:
            if d_ccw <= -2.0 * EPS:
                ccw[0] = -H[i][1]
                ccw[1] = H[i][0]
                unit2(ccw, ccw)
                offensive = True
        elif d_ccw <= -2.0 * EPS and d_cw <= -2.0 * EPS:
            return INFEASIBLE, degen
        elif d_cw <= -2.0 * EPS or d_ccw <= -2.0 * EPS or cross2(cw, H[i]) < 0.0:
            if d_cw <= -2.0 * EPS:
                unit2(ccw, cw)
            elif d_ccw <= -2.0 * EPS:
                unit2(cw, ccw)
            degen = True
            offensive = True
        if offensive:
            i = move_to_front(i, nxt, prv)
        i = nxt[i]
        if degen:
            break
    if degen:
        while i != m:
            d_cw = dot2(cw, H[i])
            d_ccw = dot2(ccw, H[i])
            if d_cw < -2.0 * EPS:
                if d_ccw < -2.0 * EPS:
                    return INFEASIBLE, degen
                cw[0] = ccw[0]
                cw[1] = ccw[1]
            elif d_ccw < -2.0 * EPS:
                ccw[0] = cw[0]
                ccw[1] = cw[1]
            i = nxt[i]
    return MINIMUM, degen


def lp_base_case(H, m, n_vec, d_vec, opt, nxt, prv):
    cw, ccw = [0.0, 0.0], [0.0, 0.0]
    status, degen = wedge(H, m, nxt, prv, cw, ccw)
    if status == INFEASIBLE:
        return status
    if status == UNBOUNDED:
        return lp_no_con(1, n_vec, d_vec, opt)
    if abs(cross2(n_vec, d_vec)) < 2.0 * EPS * EPS:
        if dot2(n_vec, n_vec) < 2.0 * EPS * EPS or dot2(d_vec, d_vec) > 2.0 * EPS * EPS:
            opt[0] = cw[0]
            opt[1] = cw[1]
            status = AMBIGUOUS
        else:
            if (not degen) and cross2(cw, n_vec) <= 0.0 and cross2(n_vec, ccw) <= 0.0:
                opt[0] = -n_vec[0]
                opt[1] = -n_vec[1]
            elif dot2(n_vec, cw) > dot2(n_vec, ccw):
                opt[0] = ccw[0]
                opt[1] = ccw[1]
            else:
                opt[0] = cw[0]
                opt[1] = cw[1]
            status = MINIMUM
    else:
        lp_min_lin_rat(degen, cw, ccw, n_vec, d_vec, opt)
        status = MINIMUM
    return status


def lp_min_lin_rat(degen, cw, ccw, n_vec, d_vec, opt):
    """lp_min_lin_rat with C's division semantics (a zero denominator in the degenerate branch gives inf / nan, and the
    comparison is then simply false or true as IEEE says)"""
    d_cw = dot2(cw, d_vec)
    d_ccw = dot2(ccw, d_vec)
    n_cw = dot2(cw, n_vec)
    n_ccw = dot2(ccw, n_vec)

    def take(v):
        opt[0] = v[0]
        opt[1] = v[1]

    if degen:
        take(cw if fdiv(n_cw, d_cw) < fdiv(n_ccw, d_ccw) else ccw)
    elif abs(d_cw) > 2.0 * EPS and abs(d_ccw) > 2.0 * EPS:
        if d_cw * d_ccw > 0.0:
            take(cw if n_cw / d_cw < n_ccw / d_ccw else ccw)
        elif d_cw > 0.0:
            opt[0] = -d_vec[1]
            opt[1] = d_vec[0]
        else:
            opt[0] = d_vec[1]
            opt[1] = -d_vec[0]
    elif abs(d_cw) > 2.0 * EPS:
        take(cw if n_ccw * d_cw > 0.0 else ccw)
    elif abs(d_ccw) > 2.0 * EPS:
        take(ccw if n_cw * d_ccw > 2.0 * EPS else cw)
    else:
        take(cw if cross2(d_vec, n_vec) > 0.0 else ccw)


def findimax(d, pln):
    imax = 0
    rmax = abs(pln[0])
    for i in range(1, d + 1):
        ab = abs(pln[i])
        if ab > rmax:
            imax = i
            rmax = ab
    return imax


def vector_up(d, eq, ivar, low, vec):
    vec[ivar] = 0.0
    for i in range(d + 1):
        if i != ivar:
            j = i if i < ivar else i - 1
            vec[i] = low[j]
            vec[ivar] -= eq[i] * low[j]
    vec[ivar] /= eq[ivar]


def vector_down(d, elim, ivar, old):
    ve = 0.0
    ee = 0.0
    for i in range(d + 1):
        ve += old[i] * elim[i]
        ee += elim[i] * elim[i]
    fac = ve / ee
    new = [0.0] * d
    for i in range(d + 1):
        if i != ivar:
            new[i if i < ivar else i - 1] = old[i] - elim[i] * fac
    return new


def plane_down(d, elim, ivar, old):
    crit = old[ivar] / elim[ivar]
    new = [0.0] * d
    for i in range(d + 1):
        if i != ivar:
            new[i if i < ivar else i - 1] = old[i] - elim[i] * crit
    return new


def linfracprog(d, H, max_size, m, n_vec, d_vec, opt, nxt, prv):
    """H: list of planes (d + 1 numbers each), indexed like the reference's `halves`"""
    if d == 1:
        if m > 0:
            return lp_base_case(H, m, n_vec, d_vec, opt, nxt, prv)
        return lp_no_con(1, n_vec, d_vec, opt)
    val = 0.0
    for j in range(d + 1):
        val += d_vec[j] * d_vec[j]
    d_vec_zero = val < (d + 1) * EPS * EPS
    status = lp_no_con(d, n_vec, d_vec, opt)
    if m <= 0:
        return status
    new_opt = [0.0] * d
    new_H = [[0.0] * d for _ in range(max_size)]
    i = 0
    while i != m:
        plane_i = H[i]
        val = 0.0
        for j in range(d + 1):
            val += opt[j] * plane_i[j]
        if val < -(d + 1) * EPS:
            imax = findimax(d, plane_i)
            if i != 0:
                fac = 1.0 / plane_i[imax]
                j = 0
                while j != i:
                    old = H[j]
                    crit = old[imax] * fac
                    new = new_H[j]
                    for k in range(d + 1):
                        if k != imax:
                            new[k if k < imax else k - 1] = old[k] - plane_i[k] * crit
                    j = nxt[j]
            if d_vec_zero:
                new_n = vector_down(d, plane_i, imax, n_vec)
                new_d = [0.0] * d
            else:
                new_n = plane_down(d, plane_i, imax, n_vec)
                new_d = plane_down(d, plane_i, imax, d_vec)
            status = linfracprog(d - 1, new_H, max_size, i, new_n, new_d, new_opt, nxt, prv)
            if status != INFEASIBLE:
                vector_up(d, plane_i, imax, new_opt, opt)
                mag = 0.0
                for j in range(d + 1):
                    mag += opt[j] * opt[j]
                mag = 1.0 / math.sqrt(mag)
                for j in range(d + 1):
                    opt[j] *= mag
            else:
                return status
            i = move_to_front(i, nxt, prv)
        i = nxt[i]
    return status


def linprog(c, A, b, perm):
    """min c.x  s.t.  A x <= b; returns (minimum, x, status); perm = the order rand_permutation would have drawn"""
    d = len(c)
    m = len(b) + 1
    x = [0.0] * d
    if m <= 1:
        return (-INF if max(abs(v) for v in c) > 0.0 else 0.0), x, None
    H = [[0.0] * d + [1.0]]
    for r in range(m - 1):
        col = [-A[r][k] for k in range(d)] + [b[r]]
        H.append(col)
    for col in H:                                  # halves.colwise().normalize()
        s = 0.0
        for v in col:
            s += v * v
        nrm = math.sqrt(s)
        for k in range(d + 1):
            col[k] = col[k] / nrm
    n_vec = list(c) + [0.0]
    d_vec = [0.0] * d + [1.0]
    opt = [0.0] * (d + 1)
    nxt = [0] * m
    prv = [0] * (m + 1)
    prv[0] = 0
    nxt[0] = perm[0] + 1
    prv[perm[0] + 1] = 0
    for i in range(m - 2):
        nxt[perm[i] + 1] = perm[i + 1] + 1
        prv[perm[i + 1] + 1] = perm[i] + 1
    nxt[perm[m - 2] + 1] = m
    status = linfracprog(d, H, m, m, n_vec, d_vec, opt, nxt, prv)
    minimum = INF
    if status != INFEASIBLE:
        if opt[d] != 0.0 and status != UNBOUNDED:
            x = [opt[k] / opt[d] for k in range(d)]
            minimum = 0.0
            for k in range(d):
                minimum += c[k] * x[k]
        if opt[d] == 0.0 or status == UNBOUNDED:
            x = [opt[k] for k in range(d)]
            minimum = -INF
    return minimum, x, status


def library_permutation(n):
    """NOT from the reference: the insertion order the LIBRARY uses in place of sdlp's process-global mt19937_64 (DESIGN.md
    section 4, deviation 2; `fixed_permutation` in csrc/sogm_corridor.hip) — an LCG Fisher-Yates, a function of n only"""
    p = list(range(n))
    s = 0x9E3779B97F4A7C15
    for i in range(n - 1, 0, -1):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        j = (s >> 33) % (i + 1)
        p[i], p[j] = p[j], p[i]
    return p


def make_problems(rng):
    """generic bounded LPs (random polytopes around a point), the shapes the path uses (d = 3: the deconfliction's
    separating plane; d = 4: the deepest interior point of a polytope, `firi.hpp:142-170`), and the corner cases:
    infeasible, unbounded, zero objective, duplicated and parallel rows, a single row"""
    out = []

    def add(kind, c, A, b):
        perm = rng.permutation(len(b)).tolist()
        out.append({"kind": kind, "c": [float(v) for v in c], "A": np.asarray(A, float).tolist(),
                    "b": [float(v) for v in b], "perm": perm})

    for d in (3, 4):
        for n in (d + 1, 8, 20, 60, 152 - 2 * d):     # + the 2 d box rows = the kernel's capacity of 152 rows
            for _ in range(6):
                N = rng.normal(size=(n, d))
                N /= np.linalg.norm(N, axis=1, keepdims=True)
                x0 = rng.uniform(-2, 2, d)
                bb = N @ x0 + rng.uniform(0.2, 2.0, n)
                # make sure it is bounded: add a box
                A = np.concatenate([N, np.eye(d), -np.eye(d)])
                b2 = np.concatenate([bb, x0 + 6.0, -(x0 - 6.0)])
                add("bounded", rng.normal(size=d), A, b2)
        for _ in range(6):       # interior-point shape: max t s.t. n.x + t <= b (d = 4 uses c = (0,0,0,-1))
            n = 12
            N = rng.normal(size=(n, d - 1))
            N /= np.linalg.norm(N, axis=1, keepdims=True)
            A = np.concatenate([N, np.ones((n, 1))], axis=1)
            add("interior", [0.0] * (d - 1) + [-1.0], A, rng.uniform(0.5, 2.0, n))
        for _ in range(6):       # infeasible: x_0 <= -1 and -x_0 <= -1 among random rows
            N = rng.normal(size=(10, d))
            e = np.zeros(d)
            e[0] = 1.0
            A = np.concatenate([N, [e], [-e]])
            add("infeasible", rng.normal(size=d), A, np.concatenate([rng.uniform(0.5, 2, 10), [-1.0, -1.0]]))
        for _ in range(6):       # unbounded: a cone of rows that leaves the objective direction open
            N = rng.normal(size=(6, d))
            c = -np.abs(rng.normal(size=d)) - 0.1
            N = -np.abs(N)
            add("unbounded", c, N, rng.uniform(0.5, 2, 6))
        for _ in range(4):       # zero objective
            N = rng.normal(size=(12, d))
            add("zero_objective", np.zeros(d), np.concatenate([N, np.eye(d), -np.eye(d)]),
                np.concatenate([rng.uniform(0.5, 2, 12), np.full(2 * d, 3.0)]))
        for _ in range(4):       # duplicated / parallel rows (degenerate vertices)
            N = rng.normal(size=(8, d))
            bb = rng.uniform(0.5, 2, 8)
            A = np.concatenate([N, N[:4], 2.0 * N[4:], np.eye(d), -np.eye(d)])
            b2 = np.concatenate([bb, bb[:4], 2.0 * bb[4:], np.full(2 * d, 4.0)])
            add("degenerate", rng.normal(size=d), A, b2)
        add("single_row", rng.normal(size=d), rng.normal(size=(1, d)), [1.0])
        # objective along a face normal: the optimum is a whole face
        N = np.concatenate([np.eye(d), -np.eye(d)])
        add("face_optimum", -np.eye(d)[0], N, np.full(2 * d, 1.0))
    return out


def main():
    rng = np.random.default_rng(0x5D19)
    probs = make_problems(rng)
    counts = {}
    for p in probs:
        v, x, st = linprog(p["c"], p["A"], p["b"], p["perm"])
        p["minimum"] = "inf" if v == INF else "-inf" if v == -INF else float(v)
        p["x"] = [float(t) for t in x]
        p["status"] = st
        v2, x2, _ = linprog(p["c"], p["A"], p["b"], library_permutation(len(p["b"])))   # what the HIP kernel must return
        p["minimum_library_order"] = "inf" if v2 == INF else "-inf" if v2 == -INF else float(v2)
        p["x_library_order"] = [float(t) for t in x2]
        counts[(p["kind"], st)] = counts.get((p["kind"], st), 0) + 1
        if p["kind"] == "bounded":      # sanity of the restatement itself: feasible and not beaten by a random feasible point
            A, b = np.asarray(p["A"]), np.asarray(p["b"])
            assert np.all(A @ np.asarray(x) <= b + 1e-7), (p["kind"], (A @ np.asarray(x) - b).max())
    print(sorted(counts.items()))
    out = {"what": "sdlp::linprog<d> restated independently in Python (tests/golden/make_lp_fixture.py); status: 0 minimum, "
                   "1 infeasible, 2 unbounded, 3 ambiguous (linfracprog's return); minimum +-inf as strings",
           "problems": probs}
    with open(os.path.join(ROOT, "tests", "golden", "lp_independent.json"), "w") as f:
        json.dump(out, f)
    print(f"written tests/golden/lp_independent.json ({len(probs)} problems)")


if __name__ == "__main__":
    main()
