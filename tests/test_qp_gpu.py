"""GPU parity: batched Bezier QP (assembly + OSQP-algorithm ADMM) and the full replan chain vs the
CPU oracle.  Trajectory coefficients within 1e-4 (north_star tolerance; observed ~1e-9), same
solver status and iteration count."""
import importlib

import numpy as np
import pytest

from helpers import hard_cases, oracle_grids

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _setup(pop, A, seed, fake=True):
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    spec = pop.config.make_spec("parity")
    sc, pva = hard_cases(pop, A, seed)
    recs = pop.scene.straight_records(sc)
    dev = sogm.upload_scene(sc)
    m = sogm.SogmMap(spec, A)
    m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
    m.addOtherAgents(sogm._dev(recs), A, dev["ego_ids"])
    ap, pp, qs = pop.config.make_astar_params(), pop.config.make_planner_params(fake), pop.config.make_qp_settings()
    P = planner.SogmPlanner(m, ap, pp, qs)
    return sogm, planner, spec, sc, pva, recs, dev, m, ap, pp, qs, P


@pytest.mark.parametrize("A,seed", [(8, 17), (12, 99)])
def test_qp_matches_oracle(pop, orc, A, seed):
    sogm, planner, spec, sc, pva, recs, dev, m, ap, pp, qs, P = _setup(pop, A, seed)
    t_start = sc["stamps"] + 0.05
    d_pva, d_ts = sogm._dev(pva, np.float64), sogm._dev(t_start, np.float64)
    s = P.search(d_pva, sogm._dev(sc["goals"], np.float64), d_ts)
    c = P.generateCorridors(d_pva, d_ts, s["route"], s["route_len"])
    q = P.optimize(d_pva, c["goal"], c["polys"], c["nfaces"], c["npoly"])
    cn = {k: v.cpu().numpy() for k, v in c.items()}
    qn = {k: v.cpu().numpy() for k, v in q.items()}
    worst = 0.0
    for a in range(A):
        M = int(cn["npoly"][a])
        assert M > 0
        goal = np.concatenate([cn["goal"][a], np.zeros(3)])
        st, x, it = orc.qp_solve(pva[a], goal, [pp.corridor_tau] * M, cn["polys"][a], cn["nfaces"][a],
                                 pp.max_faces, pp.opt_max_vel, pp.opt_max_acc, qs)
        assert qn["status"][a] == st and qn["iters"][a] == it, (a, qn["status"][a], st, qn["iters"][a], it)
        got = qn["cpts"][a, :15 * M]
        worst = max(worst, np.abs(got - x).max())
        assert np.allclose(got, x, atol=TOL, rtol=0), f"agent {a}: max diff {np.abs(got - x).max()}"
        assert not qn["cpts"][a, 15 * M:].any()
        # solution quality (reference's own test tolerance 1e-3, test_bezier_opt.cpp:156-185)
        X = got.reshape(-1, 3)
        d = np.full(M, pp.corridor_tau)
        assert np.abs(orc.bezier_eval(d, X, 0.0) - pva[a, :3]).max() < 1e-3 + 1e-3
    print("max |cpts_gpu - cpts_oracle| =", worst)
    P.close()
    m.close()


@pytest.mark.parametrize("A,seed", [(8, 17), (6, 5)])
def test_replan_chain_matches_oracle(pop, orc, A, seed):
    sogm, planner, spec, sc, pva, recs, dev, m, ap, pp, qs, P = _setup(pop, A, seed)
    t_start = sc["stamps"] + 0.05
    rec_d, ok_d = P.replan(sogm._dev(pva, np.float64), sogm._dev(sc["goals"], np.float64),
                           sogm._dev(t_start, np.float64), dev["ego_ids"])
    got = planner.records_from_bytes(rec_d.cpu().numpy())
    ok = ok_d.cpu().numpy()
    grids = oracle_grids(pop, orc, spec, sc, recs)
    n_ok = 0
    for a in range(A):
        w_ok, w, stage = orc.replan(spec, ap, pp, qs, grids[a], sc["poses"][a], sc["stamps"][a], pva[a],
                                    sc["goals"][a], t_start[a], a)
        assert ok[a] == w_ok
        assert got[a].drone_id == a and got[a].n_pieces == w.n_pieces and got[a].time_start == w.time_start
        if w_ok:
            n_ok += 1
            k = w.n_pieces
            assert list(got[a].duration)[:k] == list(w.duration)[:k]
            assert np.allclose(np.array(got[a].cpts[:15 * k]), np.array(w.cpts[:15 * k]), atol=TOL, rtol=0)
    assert n_ok > 0
    P.close()
    m.close()


def _box_chain(M, nf, max_faces, rng, tau_len=0.3):
    """A straight corridor of M overlapping polytopes along +x: the six box faces of segment i plus nf - 6 redundant
    tilted planes that do not cut the box (h.x + d <= 0 inside).  Returns polys [16][max_faces][4], nfaces [16]."""
    polys = np.zeros((16, max_faces, 4))
    nfaces = np.zeros(16, np.int32)
    for i in range(M):
        lo = np.array([i * tau_len - 0.5, -0.6, 0.4]); hi = np.array([(i + 1) * tau_len + 0.5, 0.6, 1.6])
        H = []
        for k in range(3):
            e = np.zeros(3); e[k] = 1.0
            H.append(np.concatenate([e, [-hi[k]]]))
            H.append(np.concatenate([-e, [lo[k]]]))
        c = 0.5 * (lo + hi)
        while len(H) < nf:
            n = rng.normal(size=3); n /= np.linalg.norm(n)
            # support of the box in direction n, pushed out by a margin: redundant but a real row of the QP
            sup = np.sum(np.abs(n) * 0.5 * (hi - lo)) + rng.uniform(0.05, 0.5)
            H.append(np.concatenate([n, [-(n @ c + sup)]]))
        polys[i, :nf] = np.array(H)
        nfaces[i] = nf
    return polys, nfaces


# (pieces, faces per polytope): 12 x 6 -> general path (no block factor), rows in LDS;
# 8 x 30 -> S = 1200 rows: general path, row storage in HBM scratch;
# 8 x 25 -> S = 1000: register-resident iteration with the cold row data in HBM scratch;
# 5 x 40 -> S = 1000 with few pieces; 3 x 6 -> the common small case
@pytest.mark.parametrize("M,nf", [(12, 6), (8, 30), (8, 25), (5, 40), (3, 6)])
def test_qp_solver_paths_match_oracle(pop, orc, M, nf):
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    A = 3
    spec = pop.config.make_spec("parity")
    m = sogm.SogmMap(spec, A)
    ap, pp, qs = pop.config.make_astar_params(), pop.config.make_planner_params(True), pop.config.make_qp_settings()
    P = planner.SogmPlanner(m, ap, pp, qs)
    rng = np.random.default_rng(1000 * M + nf)
    polys = np.zeros((A, 16, pp.max_faces, 4)); nfaces = np.zeros((A, 16), np.int32)
    pva = np.zeros((A, 9)); goal = np.zeros((A, 6))
    for a in range(A):
        polys[a], nfaces[a] = _box_chain(M, nf, pp.max_faces, rng)
        pva[a, :3] = [0.0, rng.uniform(-0.2, 0.2), 1.0 + rng.uniform(-0.2, 0.2)]
        pva[a, 3:6] = [rng.uniform(0.0, 0.8), 0.0, 0.0]
        goal[a, :3] = [M * 0.3, rng.uniform(-0.2, 0.2), 1.0]
    npoly = np.full(A, M, np.int32)
    dev = lambda x, t: torch.as_tensor(np.ascontiguousarray(x, dtype=t), device="cuda")
    q = P.optimize(dev(pva, np.float64), dev(goal, np.float64), dev(polys, np.float64), dev(nfaces, np.int32),
                   dev(npoly, np.int32))
    qn = {k: v.cpu().numpy() for k, v in q.items()}
    worst = 0.0
    for a in range(A):
        st, x, it = orc.qp_solve(pva[a], np.concatenate([goal[a], np.zeros(3)]), [pp.corridor_tau] * M, polys[a],
                                 nfaces[a], pp.max_faces,
                                 pp.opt_max_vel, pp.opt_max_acc, qs)
        assert qn["status"][a] == st and qn["iters"][a] == it, (a, qn["status"][a], st, qn["iters"][a], it)
        if st in (1, 2):
            got = qn["cpts"][a, :15 * M]
            worst = max(worst, np.abs(got - x).max())
            assert np.allclose(got, x, atol=TOL, rtol=0), f"agent {a}: max diff {np.abs(got - x).max()}"
    print(f"M={M} faces={nf}: statuses {qn['status'].tolist()} iters {qn['iters'].tolist()} max diff {worst:.2e}")
    assert (qn["status"] == 1).any(), "the synthetic corridors are meant to be solvable"
    P.close()
    m.close()
