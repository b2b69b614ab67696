"""CPU: properties of the map oracle that pin the reference's documented quirks."""
import numpy as np


def test_fp32_truncation_sets_kernel_width(pop, orc):
    # trap 10: 0.45f / 0.15f = 2.9999998f -> inf_step 2 (not 3); 0.3f/0.15f -> 2
    import ctypes as C
    for clearance in (0.45, 0.3):
        s = pop.config.make_spec(clearance=clearance)
        assert orc.lib().orc_inf_step(C.byref(s)) == 2
    assert np.float32(0.45) / np.float32(0.15) < 3.0


def test_ranges_are_int_division_times_res(pop, orc):
    import ctypes as C
    s = pop.config.make_spec("parity")
    r = np.zeros(3, np.float32)
    orc.lib().orc_ranges(C.byref(s), orc.fptr(r))
    assert r[0] == np.float32(33) * np.float32(0.15) == np.float32(4.9500003)
    assert r[2] == np.float32(10) * np.float32(0.15)


def test_body_particle_counts(pop, orc):
    # 3x3x4 = 36 for 0.4x0.4x0.45 and 27 for 0.3x0.3x0.4
    assert len(orc.ego_particles((0.4, 0.4, 0.45))) == 36
    assert len(orc.ego_particles((0.3, 0.3, 0.4))) == 27
    assert np.array_equal(orc.ego_particles((0.4, 0.4, 0.45)), pop.scene.body_particles((0.4, 0.4, 0.45)))


def test_voxel_index_roundtrip_and_strict_range(pop, orc):
    import ctypes as C
    s = pop.config.make_spec("parity")
    L = orc.lib()
    pose = np.array([1.25, -0.5, 1.0], np.float32)
    rng = np.random.default_rng(1)
    for idx in rng.integers(0, 66 * 66 * 20, 200):
        p = np.zeros(3, np.float32)
        L.orc_voxel_position(C.byref(s), orc.fptr(pose), int(idx), orc.fptr(p))
        q = (p - pose) + np.float32(1e-3)  # nudge inside the cell (corner convention)
        assert L.orc_voxel_index_f(C.byref(s), orc.fptr(q)) == idx
    edge = np.array([4.9500003, 0, 0], np.float32)
    assert L.orc_is_in_range_f(C.byref(s), orc.fptr(edge)) == 0  # strict '<'


def test_update_gt_static_and_moving(pop, orc):
    s = pop.config.make_spec("parity")
    sc = pop.scene.make_scene(1, 4.95, seed=0x5067, moving=True)
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    g = orc.update_gt(s, sc["cloud"], cyl, len(sc["cylinders"]), sc["poses"][0])
    assert set(np.unique(g)) <= {0.0, 1.0}
    occ0 = int((g[:, 0] > 0).sum())
    assert occ0 > 0
    # every later slice holds at most as many cells as slice 0 (advected marks may leave the window)
    for k in range(1, s.T):
        assert 0 < (g[:, k] > 0).sum() <= occ0
    # empty cloud -> all-zero grid
    g0 = orc.update_gt(s, np.zeros((0, 3), np.float32), cyl, 0, sc["poses"][0])
    assert not g0.any()


def test_neighbour_overlay_validity_chain(pop, orc):
    """A neighbour whose trajectory has not started at slice 0 is skipped for the WHOLE update
    (getParticlesWithRisk returns false on empty waypoints, particles.cpp:353-356)."""
    s = pop.config.make_spec("parity")
    sc = pop.scene.make_scene(2, 4.95, seed=3, circle_radius=2.0)
    body = pop.scene.received_body_particles()
    V = s.L * s.W * s.H
    recs = pop.scene.straight_records(sc, t_start=sc["stamps"][0] + 0.1)  # starts after slice 0
    g = np.zeros((V, s.T), np.float32)
    orc.project_neighbours(s, g, recs, 2, 0, body, sc["poses"][0], sc["stamps"][0])
    assert not g.any()
    recs = pop.scene.straight_records(sc, t_start=sc["stamps"][0] - 0.05)
    orc.project_neighbours(s, g, recs, 2, 0, body, sc["poses"][0], sc["stamps"][0])
    assert g.sum() > 0 and float(g.max()) >= 1.0
    # ego's own record is never projected
    g2 = np.zeros((V, s.T), np.float32)
    orc.project_neighbours(s, g2, recs, 1, 0, body, sc["poses"][0], sc["stamps"][0])
    assert not g2.any()


def test_query_semantics_fake_vs_riskbase(pop, orc):
    s = pop.config.make_spec("parity")
    V = s.L * s.W * s.H
    g = np.zeros((V, s.T), np.float32)
    pose = np.zeros(3, np.float32)
    pos = np.array([0.0, 0.0, 1.0])
    assert orc.query_clear(s, g, pose, pos, 0.0) == 0
    assert orc.query_clear(s, g, pose, np.array([0, 0, 3.5]), 0.0) == -1      # fake: above ceiling -> -1
    assert orc.query_clear(s, g, pose, np.array([6.0, 0, 1.0]), 0.0) == -1    # outside window
    sr = pop.config.make_spec("parity", map_kind=pop._abi.SOGM_MAP_RISKBASE)
    assert orc.query_clear(sr, g, pose, np.array([0, 0, 3.5]), 0.0) == 1      # RiskBase: -> 1
    # occupied neighbour two cells away in x is inside the 5-wide kernel, three cells is not
    import ctypes as C
    idx = orc.lib().orc_voxel_index_f(C.byref(s), orc.fptr(np.array([0.31, 0.01, 1.0], np.float32)))
    g[idx, 2] = 1.0
    assert orc.query_clear(s, g, pose, pos, 0.45) == 1   # slice floor(0.45/0.2)=2
    assert orc.query_clear(s, g, pose, pos, 0.39) == 0   # slice 1
    assert orc.query_clear(s, g, pose, pos, 99.0) == 0   # clamps to T-1
    idx3 = orc.lib().orc_voxel_index_f(C.byref(s), orc.fptr(np.array([0.46, 0.01, 1.0], np.float32)))
    g[:] = 0
    g[idx3, 0] = 1.0
    assert orc.query_clear(s, g, pose, pos, 0.0) == 0
    # fake map ignores z neighbours (degenerate z loop), RiskBase does not (needs > 1.2 region sum)
    g[:] = 0
    idz = orc.lib().orc_voxel_index_f(C.byref(s), orc.fptr(np.array([0.01, 0.01, 1.16], np.float32)))
    g[idz, 0] = 5.0
    assert orc.query_clear(s, g, pose, pos, 0.0) == 0
    assert orc.query_clear(sr, g, pose, pos, 0.0) == 1


def test_mark_whose_index_leaves_the_array_is_dropped(pop, orc):
    """Reference UB (map.h:169-174): z one ulp below +range passes the strict range test, but "z + rz" and the fp32
    division round up and the z index equals H — voxel index >= V, a write outside risk_maps_ in the reference.  The
    oracle (like the HIP path) drops the mark instead of writing past its array."""
    import ctypes as C
    s = pop.config.make_spec("parity")
    L = orc.lib()
    rz = np.float32(s.H // 2) * np.float32(s.resolution)
    z = np.nextafter(rz, np.float32(0), dtype=np.float32)  # largest float below +rz
    p = np.array([0.0, 0.0, z], np.float32)
    assert L.orc_is_in_range_f(C.byref(s), orc.fptr(p)) == 1
    assert L.orc_voxel_index_f(C.byref(s), orc.fptr(p)) >= s.L * s.W * s.H   # the reference's index leaves the array
    pose = np.zeros(3, np.float32)
    cloud = np.array([[0.0, 0.0, z], [0.3, 0.3, 0.3]], np.float32)
    cyl = pop.scene.cylinders_to_struct(np.zeros((0, 5)))
    g = orc.update_gt(s, cloud, cyl, 0, pose)
    assert int((g[:, 0] > 0).sum()) == 1 and int((g > 0).sum()) == s.T  # only the interior point is marked
