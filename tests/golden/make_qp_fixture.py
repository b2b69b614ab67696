#!/usr/bin/env python
"""An INDEPENDENT second restatement of BezierOpt's QP assembly — written straight from the reference's text in numpy WITHOUT
reading oracle/ — whose cost matrix, constraint matrix and bounds on seeded problems are committed as
tests/golden/qp_assembly_independent.json; a CPU test holds the C++ oracle (`orc_qp_assemble`) to them entry by entry
(tests/test_qp_independent.py; the kernel's block-structured assembly is held to that oracle through the solves of
tests/test_qp_gpu.py: same status, iteration count and coefficients).  It does not pin the oracle to the
REFERENCE (Eigen / OSQP absent), it makes two separately written readings of the assembly agree.

Restated, block by block:  traj_opt/src/bezier_optimizer.cpp
  setup                      :27-52      (init_ / goal_ = 3 x 3, rows position, velocity, acceleration; DM_ = DIM M (N + 1))
  calcCtrlPtsCvtMat          :62-82      p2v_, v2a_, a2j_: forward differences of the control points times N, N - 1, N - 2
  calcMinJerkCost            :90-105     QM = p2j' [I/3 I/6; I/6 I/3] p2j on every segment's diagonal block (NOT time-scaled)
  addConstraints             :107-130    row counts and the order continuity, dynamical, safety
  addContinuityConstraints   :136-216    position / velocity / acceleration: initial, M - 1 junctions, final
  addDynamicalConstraints    :218-250    |p2v block| <= vmax t per velocity control point, |p2a block| <= amax t^2
  addSafetyConstraints       :252-270    per segment, per face, per control point: n . p <= -d, lower bound -OSQP_INFTY
N = Bernstein::ORDER = 4, DIM = 3 (include/bernstein/bezier_optimizer.hpp:15,39).  -OSQP_INFTY is stored as null (the
constant lives in OSQP's headers, which the reference does not ship).
Run from the repo root:   python tests/golden/make_qp_fixture.py
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N, DIM = 4, 3


def cvt_mats():
    I = np.eye(DIM)
    p2v = np.zeros((DIM * N, DIM * (N + 1)))
    v2a = np.zeros((DIM * (N - 1), DIM * N))
    a2j = np.zeros((DIM * (N - 2), DIM * (N - 1)))
    for i in range(N):
        p2v[i * DIM:(i + 1) * DIM, i * DIM:(i + 1) * DIM] = -N * I
        p2v[i * DIM:(i + 1) * DIM, (i + 1) * DIM:(i + 2) * DIM] = N * I
    for i in range(N - 1):
        v2a[i * DIM:(i + 1) * DIM, i * DIM:(i + 1) * DIM] = -(N - 1) * I
        v2a[i * DIM:(i + 1) * DIM, (i + 1) * DIM:(i + 2) * DIM] = (N - 1) * I
    for i in range(N - 2):
        a2j[i * DIM:(i + 1) * DIM, i * DIM:(i + 1) * DIM] = -(N - 2) * I
        a2j[i * DIM:(i + 1) * DIM, (i + 1) * DIM:(i + 2) * DIM] = (N - 2) * I
    return p2v, v2a, a2j


def assemble(start, goal, t, polys, vmax, amax):
    """start, goal: 3 x 3 (rows p, v, a); t: M durations; polys: list of [F_i][4] half-spaces n . x + d <= 0.
    Returns Q [DM, DM], A [m, DM], lb [m] (None = -OSQP_INFTY), ub [m]."""
    start, goal, t = np.asarray(start, float), np.asarray(goal, float), [float(x) for x in t]
    M = len(t)
    W = DIM * (N + 1)      # variables of one segment
    DM = M * W
    p2v, v2a, a2j = cvt_mats()
    I = np.eye(DIM)
    # cost
    p2j = a2j @ v2a @ p2v
    P = np.block([[I / 3, I / 6], [I / 6, I / 3]])
    QM = p2j.T @ P @ p2j
    Q = np.zeros((DM, DM))
    for i in range(M):
        Q[i * W:(i + 1) * W, i * W:(i + 1) * W] = QM
    n_safe = sum(len(c) for c in polys) * (N + 1)
    n_cont = (1 + M) * DIM * 3
    n_dyn = M * (DIM * N + DIM * (N - 1))
    m = n_safe + n_cont + n_dyn
    A, ub, lb = np.zeros((m, DM)), np.zeros(m), [0.0] * m
    idx = 0

    def eq(rows):
        nonlocal idx
        for k in range(DIM):
            ub[idx + k] = rows[k]
            lb[idx + k] = float(rows[k])
        idx += DIM

    t0, tM = t[0], t[M - 1]
    # position
    A[idx:idx + DIM, 0:DIM] = I
    eq(start[0])
    for i in range(1, M):
        A[idx:idx + DIM, i * W:i * W + DIM] = I
        A[idx:idx + DIM, i * W - DIM:i * W] = -I
        eq(np.zeros(3))
    A[idx:idx + DIM, M * W - DIM:M * W] = I
    eq(goal[0])
    # velocity
    A[idx:idx + DIM, 0:DIM] = -N * I
    A[idx:idx + DIM, DIM:2 * DIM] = N * I
    eq(start[1] * t0)
    for i in range(1, M):
        t1, t1_ = t[i], t[i - 1]
        A[idx:idx + DIM, i * W:i * W + DIM] = -N * I / t1
        A[idx:idx + DIM, i * W + DIM:i * W + 2 * DIM] = N * I / t1
        A[idx:idx + DIM, i * W - DIM:i * W] = -N * I / t1_
        A[idx:idx + DIM, i * W - 2 * DIM:i * W - DIM] = N * I / t1_
        eq(np.zeros(3))
    A[idx:idx + DIM, M * W - 2 * DIM:M * W - DIM] = -N * I
    A[idx:idx + DIM, M * W - DIM:M * W] = N * I
    eq(goal[1] * tM)
    # acceleration
    p2a = (v2a @ p2v)[0:DIM, 0:3 * DIM]
    A[idx:idx + DIM, 0:3 * DIM] = p2a
    eq(start[2] * t0 * t0)
    for i in range(1, M):
        t2, t2_ = t[i] ** 2, t[i - 1] ** 2
        A[idx:idx + DIM, i * W:i * W + 3 * DIM] = p2a / t2
        A[idx:idx + DIM, i * W - 3 * DIM:i * W] = -p2a / t2_
        eq(np.zeros(3))
    A[idx:idx + DIM, M * W - 3 * DIM:M * W] = p2a
    eq(goal[2] * tM * tM)
    assert idx == n_cont
    # dynamics
    p2v_b = p2v[0:DIM, 0:2 * DIM]
    for i in range(M):
        for j in range(N):
            A[idx:idx + DIM, i * W + j * DIM:i * W + j * DIM + 2 * DIM] = p2v_b
            for k in range(DIM):
                ub[idx + k] = vmax * 1.0 * t[i]
                lb[idx + k] = -vmax * 1.0 * t[i]
            idx += DIM
    for i in range(M):
        for j in range(N - 1):
            A[idx:idx + DIM, i * W + j * DIM:i * W + j * DIM + 3 * DIM] = p2a
            for k in range(DIM):
                ub[idx + k] = amax * 1.0 * t[i] * t[i]
                lb[idx + k] = -amax * 1.0 * t[i] * t[i]
            idx += DIM
    assert idx == n_cont + n_dyn
    # safety
    for i in range(M):
        for face in polys[i]:
            for n in range(N + 1):
                A[idx, i * W + n * DIM:i * W + (n + 1) * DIM] = face[0:3]
                ub[idx] = -face[3]
                lb[idx] = None
                idx += 1
    assert idx == m
    return Q, A, lb, ub


def problems():
    rng = np.random.RandomState(0x51D)
    out = []
    for M, faces in ((1, [6]), (2, [4, 8]), (3, [6, 6, 6]), (4, [5, 7, 4, 8]), (6, [8, 6, 7, 5, 4, 6]), (8, [6] * 8)):
        t = rng.uniform(0.3, 2.5, M)
        start = np.stack([rng.uniform(-5, 5, 3), rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 3)])
        goal = np.stack([rng.uniform(-5, 5, 3), rng.uniform(-2, 2, 3), np.zeros(3)])
        polys = []
        for F in faces:
            nrm = rng.normal(size=(F, 3))
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            d = -rng.uniform(0.5, 3.0, F) - nrm @ rng.uniform(-4, 4, 3)
            polys.append(np.concatenate([nrm, d[:, None]], axis=1))
        out.append((start, goal, t, polys, float(rng.uniform(1.5, 4.0)), float(rng.uniform(2.0, 6.0))))
    return out


def main():
    cases = []
    for start, goal, t, polys, vmax, amax in problems():
        Q, A, lb, ub = assemble(start, goal, t, polys, vmax, amax)
        assert np.allclose(Q, Q.T) and np.linalg.eigvalsh(Q).min() > -1e-9
        r, c = np.nonzero(A)
        qr, qc = np.nonzero(Q)
        cases.append({"start": start.tolist(), "goal": goal.tolist(), "t": t.tolist(), "polys": [p.tolist() for p in polys],
                      "vmax": vmax, "amax": amax, "n": int(Q.shape[0]), "m": int(A.shape[0]),
                      "Q": [[int(i), int(j), float(Q[i, j])] for i, j in zip(qr, qc)],
                      "A": [[int(i), int(j), float(A[i, j])] for i, j in zip(r, c)],
                      "lb": lb, "ub": [float(x) for x in ub]})
    path = os.path.join(ROOT, "tests", "golden", "qp_assembly_independent.json")
    json.dump({"what": "BezierOpt::setup's matrices from an independent numpy restatement (make_qp_fixture.py); "
                       "lb null = -OSQP_INFTY; Q and A as (row, col, value) of the non-zeros", "cases": cases}, open(path, "w"))
    print(path, len(cases), "problems", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
