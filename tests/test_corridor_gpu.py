"""GPU parity: corridor stage (obstacle points -> FIRI + MVIE -> shrink -> validity / intersection /
goal LPs) vs the CPU oracle.  Same fp64 operation order on both sides -> polytopes are compared
bit for bit (fallback tolerance 1e-9 documented if the exact check ever has to be relaxed)."""
import importlib

import numpy as np
import pytest

from helpers import hard_cases, oracle_grids

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("A,seed,fake", [(8, 17, True), (12, 99, True), (6, 5, False), (10, 1234, True)])
def test_corridors_match_oracle(pop, orc, A, seed, fake):
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    spec = pop.config.make_spec("parity")
    sc, pva = hard_cases(pop, A, seed)
    recs = pop.scene.straight_records(sc)
    dev = sogm.upload_scene(sc)
    m = sogm.SogmMap(spec, A)
    m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
    m.addOtherAgents(sogm._dev(recs), A, dev["ego_ids"])
    ap = pop.config.make_astar_params()
    pp = pop.config.make_planner_params(fake)
    P = planner.SogmPlanner(m, ap, pp, pop.config.make_qp_settings())
    t_start = sc["stamps"] + 0.05
    d_pva, d_ts = sogm._dev(pva, np.float64), sogm._dev(t_start, np.float64)
    s = P.search(d_pva, sogm._dev(sc["goals"], np.float64), d_ts)
    c = P.generateCorridors(d_pva, d_ts, s["route"], s["route_len"])
    c = {k: v.cpu().numpy() for k, v in c.items()}
    route, rlen = s["route"].cpu().numpy(), s["route_len"].cpu().numpy()
    grids = oracle_grids(pop, orc, spec, sc, recs)
    nonbox = 0
    for a in range(A):
        w = orc.corridor_generate(spec, pp, grids[a], sc["poses"][a], sc["stamps"][a], pva[a], t_start[a],
                                  route[a, :rlen[a]])
        assert c["npoly"][a] == w["npoly"], f"agent {a}: npoly {c['npoly'][a]} vs {w['npoly']}"
        assert np.array_equal(c["nfaces"][a], w["nfaces"]), f"agent {a}: {c['nfaces'][a]} vs {w['nfaces']}"
        for i in range(w["npoly"]):
            nf = w["nfaces"][i]
            got, want = c["polys"][a, i, :nf], w["polys"][i, :nf]
            assert np.array_equal(got, want), f"agent {a} poly {i}: max diff {np.abs(got - want).max()}"
            nonbox += max(nf - 6, 0)
        assert np.array_equal(c["goal"][a], w["goal"])
    assert nonbox > 0, "test scene produced only bounding-box corridors"
    P.close()
    m.close()


def test_corridor_box_with_more_than_4096_points_gives_the_uncapped_answer(pop, orc):
    """The reference grows its obstacle-point vector (baseline.cpp:317 reserves 2000 and push_backs past it); the
    point capacity here (pc_capacity 16384, round 2: 4096) must not show in the result: an agent flies down a canyon
    between two solid walls, every corridor box holds > 4096 obstacle points (three SOGM slices of both walls), and
    the polytopes must be the ones the oracle computes with an effectively unlimited capacity — not "corridor
    invalid", which is what exceeding a capacity means on both sides."""
    import ctypes as C
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    planner = importlib.import_module("pred-occ-planner_amd.planner")
    spec = pop.config.make_spec("parity")
    sc = pop.scene.make_scene(1, 4.95, seed=3, n_cyl=0)
    sc["starts"][0] = (0.0, -3.0, 1.0)
    sc["goals"][0] = (0.0, 3.0, 1.0)
    sc["poses"][0] = (0.0, -3.0, 1.0)
    sc["cylinders"] = np.zeros((0, 5))
    # two solid walls 0.1 m lattice: x in [0.85, 2.2] and [-2.2, -0.85], all y, z in [0, 3)
    xs = np.concatenate([np.arange(0.85, 2.2, 0.1), -np.arange(0.85, 2.2, 0.1)])
    ys, zs = np.arange(-7.9, 7.9, 0.1), np.arange(0.05, 3.0, 0.1)
    gx, gy, gz = np.meshgrid(xs, ys, zs, indexing="ij")
    sc["cloud"] = np.ascontiguousarray(np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1), np.float32)
    dev = sogm.upload_scene(sc)
    m = sogm.SogmMap(spec, 1)
    m.updateMap(dev["cloud"], dev["cloud_range"], dev["cylinders"], dev["n_cyl"], dev["poses"], dev["stamps"])
    ap, pp = pop.config.make_astar_params(), pop.config.make_planner_params(True)
    assert pp.pc_capacity >= 16384
    P = planner.SogmPlanner(m, ap, pp, pop.config.make_qp_settings())
    pva = np.concatenate([sc["starts"], np.zeros((1, 6))], axis=1)
    t_start = sc["stamps"] + 0.02
    d_pva, d_ts = sogm._dev(pva, np.float64), sogm._dev(t_start, np.float64)
    s = P.search(d_pva, sogm._dev(sc["goals"], np.float64), d_ts)
    assert int(s["ret"][0]) != 0 and int(s["route_len"][0]) >= 4
    c = {k: v.cpu().numpy() for k, v in P.generateCorridors(d_pva, d_ts, s["route"], s["route_len"]).items()}
    lib = pop.lib()
    lib.sogm_debug_corridor_stats.argtypes = [C.c_void_p, C.c_void_p]
    dbg = np.zeros((16, 16), np.int64)
    lib.sogm_debug_corridor_stats(P._p, dbg.ctypes.data)
    npts = dbg[dbg[:, 10] > 0, 0]
    assert npts.max() > 4096 and npts.max() < pp.pc_capacity, npts   # past the old limit, inside the new one
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    g = orc.update_gt(spec, sc["cloud"], cyl, 0, sc["poses"][0])
    route = s["route"].cpu().numpy()[0, :int(s["route_len"][0])]
    big = pop.config.make_planner_params(True)
    big.pc_capacity = 1 << 20                                         # the oracle with no practical limit
    w = orc.corridor_generate(spec, big, g, sc["poses"][0], sc["stamps"][0], pva[0], t_start[0], route)
    assert w["npoly"] >= 3 and c["npoly"][0] == w["npoly"]
    assert np.array_equal(c["nfaces"][0], w["nfaces"])
    for i in range(w["npoly"]):
        nf = w["nfaces"][i]
        assert np.array_equal(c["polys"][0, i, :nf], w["polys"][i, :nf]), i
    assert P.counters()["corridor_capacity"] == 0
    P.close()
    m.close()
