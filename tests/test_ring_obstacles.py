"""Ring obstacles (Cylinder.type == 2, fake_particle_risk_voxel.cpp:137-149): a moving ring advects its voxels
into the future slices through the plane-projection test.  CPU: the oracle's restatement of Eigen's operation
sequence; GPU: k_stamp_cloud bit-exact against it, rings before and after the cylinders in the record list."""
import importlib

import numpy as np
import pytest


def _ring_scene(pop, A, seed, first):
    sc = pop.scene.make_scene(A, 4.95, seed=seed, moving=True, circle_radius=3.0, n_cyl=6)
    rng = np.random.default_rng(seed)
    rings = []
    for a in range(A):  # one ring next to every agent (inside its window), tilted, moving
        p = sc["starts"][a]
        ang = rng.uniform(0, 2 * np.pi)
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        half = rng.uniform(0.0, 0.6)
        q = np.concatenate([[np.cos(half)], np.sin(half) * ax])
        rings.append([p[0] + 1.8 * np.cos(ang), p[1] + 1.8 * np.sin(ang), 1.2, rng.uniform(1.0, 2.0),
                      rng.uniform(-1, 1), rng.uniform(-1, 1), *q])
    # an upright ring (q = 90 deg about x: the ring's plane is vertical) overlapping a cylinder's footprint
    c = sc["cylinders"][0]
    rings.append([c[0], c[1], 1.5, 2.0 * c[2] + 0.3, 0.7, -0.4, np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0])
    arr, n = pop.scene.add_rings(sc, rings, first=first)
    return sc, arr, n, np.asarray(rings)


def test_oracle_ring_velocity_marks_future_slices(pop, orc):
    spec = pop.config.make_spec("parity")
    sc, arr, n, rings = _ring_scene(pop, 2, 3, True)
    static = arr.__class__.from_buffer_copy(arr)
    for i in range(n):
        static[i].vx = static[i].vy = 0.0
    for a in range(2):
        g_mov = orc.update_gt(spec, sc["cloud"], arr, n, sc["poses"][a])
        g_sta = orc.update_gt(spec, sc["cloud"], static, n, sc["poses"][a])
        assert np.array_equal(g_mov[:, 0], g_sta[:, 0])            # slice 0 is the cloud itself
        assert not np.array_equal(g_mov[:, 3], g_sta[:, 3])        # the ring (and cylinders) moved
        # with every record static the future slices repeat slice 0 up to the corner-position round trip
        # (getVoxelPosition -> getVoxelIndex truncates x*res - range + range back to x or x - 1, map.h:153-215)
        assert abs(int(g_sta[:, 0].sum()) - int(g_sta[:, spec.T - 1].sum())) < 0.2 * g_sta[:, 0].sum()
    # only rings moving: the cells that differ from the static map lie near a ring's circle
    only_ring = arr.__class__.from_buffer_copy(arr)
    for i in range(n):
        if only_ring[i].type == 3:
            only_ring[i].vx = only_ring[i].vy = 0.0
    g_ring = orc.update_gt(spec, sc["cloud"], only_ring, n, sc["poses"][0])
    g_sta = orc.update_gt(spec, sc["cloud"], static, n, sc["poses"][0])
    assert (g_ring[:, 1] != g_sta[:, 1]).sum() > 10


@pytest.mark.gpu
@pytest.mark.parametrize("first", [True, False])
def test_ring_stamp_bit_exact(pop, orc, first):
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    A = 4
    spec = pop.config.make_spec("parity")
    sc, arr, n, rings = _ring_scene(pop, A, 17, first)
    dev = sogm.upload_scene(sc)
    m = sogm.SogmMap(spec, A)
    m.updateMap(dev["cloud"], dev["cloud_range"], sogm._dev(arr), n, dev["poses"], dev["stamps"])
    static = arr.__class__.from_buffer_copy(arr)
    for i in range(n):
        if static[i].type == 2:
            static[i].vx = static[i].vy = 0.0
    n_ring_cells = 0
    for a in range(A):
        want = orc.update_gt(spec, sc["cloud"], arr, n, sc["poses"][a])
        got = m.download(a)
        assert np.array_equal(got, want), f"agent {a}: {(got != want).sum()} cells differ"
        n_ring_cells += int((want != orc.update_gt(spec, sc["cloud"], static, n, sc["poses"][a])).sum())
    assert n_ring_cells > 50, "the rings must actually move cells in this scene"
    m.close()
