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
