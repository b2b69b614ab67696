"""Particle-filter SOGM (row a6): HIP path vs the CPU oracle on the same sensor sequences.

Slot contents, voxel indices and particle positions/velocities must be bit-exact (the parallel
ordered-first-fit reproduces the reference's sequential sweep); weights are compared bit-exact as
well (same fp32 summation order); the future-occupancy accumulators are scatter-added with atomics,
so they are compared with the 1e-4 tolerance north_star states for occupancy values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(pop, orc, grid, seeds, n_updates, mutate=None, wall=True):
    import importlib
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    dsp = importlib.import_module("pred-occ-planner_amd.dsp")
    spec = pop.config.make_spec(grid)
    spec.map_kind = pop._abi.SOGM_MAP_RISKBASE
    A = len(seeds)
    P = dsp.make_dsp_params(spec.T)
    tabs = dsp.make_tables(11, n_gauss=1 << 18, n_rand=1 << 12)
    half = (spec.L * 0.15 / 2, spec.W * 0.15 / 2, spec.H * 0.15 / 2)
    seqs = [pop.scene.make_dsp_sequence(s, n_updates, half=half, wall=wall) for s in seeds]
    if mutate:
        mutate(seqs)
    m = sogm.SogmMap(spec, A)
    g = dsp.DspMap(m, P, tabs)
    oracles = [orc.DspOracle(spec, P, tabs) for _ in range(A)]
    stats = np.zeros(16, np.int64)
    for k in range(n_updates):
        pts = np.concatenate([seqs[a][k]["points"] for a in range(A)], axis=0)
        lab = np.concatenate([seqs[a][k]["labels"] for a in range(A)], axis=0)
        ends = np.cumsum([len(seqs[a][k]["points"]) for a in range(A)])
        rng = np.stack([np.concatenate([[0], ends[:-1]]), ends], axis=1).astype(np.int32)
        if len(pts) == 0:
            pts, lab = np.zeros((1, 3), np.float32), np.zeros((1, 4), np.float32)
        pos = np.stack([seqs[a][k]["pos"] for a in range(A)]).astype(np.float32)
        quat = np.stack([seqs[a][k]["quat"] for a in range(A)]).astype(np.float32)
        stamp = np.asarray([seqs[a][k]["stamp"] for a in range(A)], np.float64)
        ok = g.update(sogm._dev(pts, np.float32), sogm._dev(lab, np.float32), sogm._dev(rng, np.int32),
                      sogm._dev(pos, np.float32), sogm._dev(quat, np.float32), sogm._dev(stamp, np.float64))
        ok = ok.cpu().numpy()
        for a in range(A):
            s = seqs[a][k]
            want_ok = oracles[a].update(s["points"], s["labels"], s["pos"], s["quat"], s["stamp"])
            assert ok[a] == want_ok
            ws, wo, wc = oracles[a].state()
            gs, go, gc = g.download_state(a)
            assert gc[10] == 0 and gc[11] == 0 and gc[12] == 0, f"device error counters {gc}"
            if want_ok:
                wn, wp, wm = oracles[a].observations(g.NP)
                gn, gp, gm = g.download_observations(a)
                assert np.array_equal(gn, wn)
                assert np.array_equal(gm, wm)
                OM = P.obs_max_per_pyramid
                mask = np.arange(OM)[None, :] < wn[:, None]
                assert np.array_equal(gp[mask][:, [0, 1, 2, 4]], wp[mask][:, [0, 1, 2, 4]])
                assert np.array_equal(gp[mask][:, 3], wp[mask][:, 3]), "C_k differs"
            # particle store: flags and slots identical, payload bit-exact (slot 8 = update time, not stored)
            assert np.array_equal(gs[:, :, 0], ws[:, :, 0]), f"update {k} agent {a}: slot flags differ"
            live = ws[:, :, 0] > 0.1
            for f in (1, 2, 3, 4, 5, 6):
                assert np.array_equal(gs[:, :, f][live], ws[:, :, f][live]), f"update {k} agent {a} field {f}"
            assert np.array_equal(gs[:, :, 7][live], ws[:, :, 7][live]), f"update {k} agent {a}: weights differ"
            assert np.array_equal(gc[[0, 1, 2, 4, 5, 6]], wc[[0, 1, 2, 4, 5, 6]]), (gc, wc)
            assert np.array_equal(go[:, :4], wo[:, :4])
            np.testing.assert_allclose(go[:, 4:], wo[:, 4:], rtol=1e-4, atol=1e-6)
            stats[:3] = wc[:3]
            stats[3] = max(stats[3], int(live.sum()))
        if k % 3 == 2 or k == n_updates - 1:
            n_occ = g.publish().cpu().numpy()
            for a in range(A):
                want, wn_occ = oracles[a].publish(spec.risk_threshold, int(np.float32(spec.clearance) / np.float32(spec.resolution)))
                got = m.download(a)
                assert n_occ[a] == wn_occ
                np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)
    g.close()
    m.close()
    for o in oracles:
        o.close()
    return stats


def test_dsp_sequence_parity_grid(pop, orc):
    """66x66x20x6 (the reference's compiled size): voxel-full and pyramid-full events both occur."""
    st = _run(pop, orc, "parity", [0x71, 0x72], 14)
    assert st[0] > 0, "no voxel-overflow event exercised"
    assert st[1] > 0, "no pyramid-overflow event exercised"
    assert st[3] > 5000


def test_dsp_small_grid_shallow_pyramids(pop, orc):
    """40x40x20x10: SAFE_PARTICLE_NUM_PYRAMID drops to 10, so pyramid overflow (particles that vanish and
    free their slot for later arrivals) is frequent — the place/pyramid fixed point must still converge."""
    st = _run(pop, orc, "cfg0", [0x81], 12)
    assert st[1] > 20


def test_dsp_rejected_and_empty_updates(pop, orc):
    """Invalid quaternion / time going backwards -> update() returns 0 and nothing changes (:176-203);
    an empty cloud re-uses the previous new-born list (:1488)."""
    def mutate(seqs):
        seqs[0][3]["quat"] = np.asarray([1.5, 0, 0, 0], np.float32)
        seqs[0][5]["stamp"] = seqs[0][4]["stamp"] - 1.0
        seqs[1][4]["points"] = np.zeros((0, 3), np.float32)
        seqs[1][4]["labels"] = np.zeros((0, 4), np.float32)
        seqs[1][6]["pos"] = seqs[1][6]["pos"] + np.asarray([20.0, 0, 0], np.float32)  # > 10 m jump
    _run(pop, orc, "parity", [0x91, 0x92], 8, mutate=mutate)


def test_dsp_fast_sensor_particles_leave_map(pop, orc):
    """A climbing sensor: ground particles leave through the bottom of the map (removeParticle,
    :738-741) while their slots still count as occupied for earlier arrivals of the same sweep."""
    import importlib
    def mutate(seqs):
        for s in seqs:
            for k, u in enumerate(s):
                u["pos"] = (u["pos"] + np.asarray([0.05 * k, 0, 0.12 * k], np.float32)).astype(np.float32)
    st = _run(pop, orc, "parity", [0xA1], 10, mutate=mutate)
    assert st[2] > 0


def test_dsp_full_size_properties_cfg1(pop):
    """BASELINE configs[1] (16 agents, 100^3 x 15, 640x480 depth clouds): no oracle at this size — the
    size-independent invariants of DSPMap: every live particle sits in the voxel its position maps to, vz == 0,
    flags after resampling are 1.0 / 0.6, the voxel weight equals the sum of its particles' weights, the published
    grid equals the accumulated future status, publishing clears the accumulators, no capacity error."""
    import importlib
    import torch
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    dsp = importlib.import_module("pred-occ-planner_amd.dsp")
    A = 16
    spec = pop.config.make_spec("cfg1", map_kind=pop._abi.SOGM_MAP_RISKVOXEL)
    m = sogm.SogmMap(spec, A)
    g = dsp.DspMap(m, dsp.make_dsp_params(spec.T), dsp.make_tables(5))
    cap = 5000
    clouds = [pop.scene.make_depth_cloud(40 + a) for a in range(A)]
    n_pix = len(clouds[0])
    raw = sogm._dev(np.concatenate(clouds, axis=0), np.float32)
    rng = sogm._dev(np.stack([np.arange(A) * n_pix, (np.arange(A) + 1) * n_pix], axis=1), np.int32)
    labels = torch.zeros((A * cap, 4), dtype=torch.float32, device="cuda")
    base = torch.arange(A, dtype=torch.int32, device="cuda") * cap
    quat = sogm._dev(np.tile(np.float32([1, 0, 0, 0]), (A, 1)), np.float32)
    for k in range(6):
        pos = sogm._dev(np.tile(np.float32([0.04 * k, 0.0, 0.0]), (A, 1)), np.float32)
        stamps = sogm._dev(np.full(A, 10.0 + k / 30.0), np.float64)
        pts, cnt = m.filterPointCloud(raw, rng, 0.15, cap)
        ok = g.update(pts.view(-1, 3), labels, torch.stack([base, base + cnt], dim=1).contiguous(), pos, quat, stamps)
        assert bool(ok.all())
    assert int(cnt.min()) > 500
    half = np.float32(spec.resolution) * np.float32([spec.L, spec.W, spec.H]) * np.float32(0.5)
    for a in (0, A - 1):
        st, ob, c = g.download_state(a)
        assert c[10] == 0 and c[11] == 0 and c[12] == 0, c
        live = st[:, :, 0] > 0.1
        assert live.sum() > 5000
        assert set(np.unique(st[:, :, 0][live])) <= {np.float32(1.0), np.float32(0.6)}
        v, p = np.nonzero(live)
        idx = ((st[v, p, 4:7] + half) / np.float32(spec.resolution)).astype(np.int32)
        assert np.array_equal(idx[:, 2] * spec.W * spec.L + idx[:, 1] * spec.L + idx[:, 0], v)
        assert np.all(st[:, :, 3] == 0)
        np.testing.assert_allclose(ob[:, 0], (st[:, :, 7] * live).sum(axis=1), rtol=1e-5, atol=1e-7)
        fut = ob[:, 4:].copy()
        if a == 0:
            g.publish()
            got = m.download(0)
            diff = np.nonzero(got != fut)
            assert np.all(got[diff] == 0) and np.all(diff[1] < 3)      # only the inflate-kernel zeroing differs
            _, ob2, _ = g.download_state(0)
            assert not ob2[:, 4:].any()
    g.close()
    m.close()
