"""Several ranks of one swarm co-simulated by one process (driver.LoopbackHub, the harness of bench.py's configs[3] rank
share, tools/bench_rank_share.py): the same records as one process flying every agent.  Reference behaviour: every agent
reads the swarm's records of the tick before (plan_manager.cpp:364-399 -> particles.cpp:131-191)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_CHILD = r"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["SOGM_REPO"])
pop = importlib.import_module("pred-occ-planner_amd")
driver = importlib.import_module("pred-occ-planner_amd.driver")
WORLD, A_LOC, TICKS = 3, 4, 6
spec = pop.config.make_spec("parity")
scene = pop.scene.make_scene(A_LOC * WORLD, (spec.L // 2) * 0.15, seed=0x5069)
hub = driver.LoopbackHub(WORLD, A_LOC)
sws = [driver.SwarmTick("parity", A_LOC, r, WORLD, 0, scene=scene, moving_world=True, prestamp=False,
                        exchange=hub.exchange(r), grids=None if r == 0 else 1) for r in range(WORLD)]
oks, rows = [], [[] for _ in range(WORLD)]
for k in range(TICKS):
    n = 0
    for sw in sws:
        assert sw.exchange.active and sw.publish and sw.A_tot == A_LOC * WORLD
        n += int(sw.step().sum().item())
    for r, sw in enumerate(sws):
        rows[r].append(sw.own.clone())      # what rank r contributes to the all-gather of tick k
    hub.commit()
    oks.append(n)
torch.cuda.synchronize()
tables = [sw.records_all().cpu().numpy().copy() for sw in sws]
owns = [sw.own.cpu().numpy().copy() for sw in sws]
for sw in sws:
    assert sw.planner.flow_failures()[1] == 0
    sw.close()
# the same flight by ONE process that owns all 12 agents
sw = driver.SwarmTick("parity", A_LOC * WORLD, scene=scene, moving_world=True, prestamp=False)
ref_oks = [int(sw.step().sum().item()) for _ in range(TICKS)]
ref = sw.records_all().cpu().numpy().copy()
sw.close()
for r in range(WORLD):
    assert np.array_equal(tables[r], ref), ("rank table differs from the single-process table", r)
    assert np.array_equal(owns[r], ref[r * A_LOC:(r + 1) * A_LOC]), ("own rows differ", r)
assert oks == ref_oks and sum(ref_oks) >= 3 * TICKS, (oks, ref_oks)
# replay (tools/bench_rank_share.py's timed pass): rank 1 alone, the rows of ranks 0 and 2 replayed tick by tick from the
# co-simulation above — bit-identical to its rows there, every tick
hub2 = driver.LoopbackHub(WORLD, A_LOC)
rep = driver.SwarmTick("parity", A_LOC, 1, WORLD, 0, scene=scene, moving_world=True, prestamp=False, exchange=hub2.exchange(1))
for k in range(TICKS):
    rep.step()
    assert torch.equal(rep.own, rows[1][k]), ("replayed flight differs from the co-simulated one", k)
    hub2.commit({0: rows[0][k], 2: rows[2][k]})
assert np.array_equal(rep.records_all().cpu().numpy(), ref)
rep.close()
# a rank flown WITHOUT its neighbours must differ somewhere (otherwise the comparison above proves nothing about the
# cross-rank rows): rank 1 alone, the others silent
hub1 = driver.LoopbackHub(WORLD, A_LOC)
solo = driver.SwarmTick("parity", A_LOC, 1, WORLD, 0, scene=scene, moving_world=True, prestamp=False, exchange=hub1.exchange(1))
for k in range(TICKS):
    solo.step(); hub1.commit()
torch.cuda.synchronize()
solo_own = solo.own.cpu().numpy().copy()
solo_tab = solo.records_all().cpu().numpy().copy()
solo.close()
assert not solo_tab[:A_LOC].any() and not solo_tab[2 * A_LOC:].any()          # silent ranks: empty rows
print("rank share ok", oks, "solo differs:", not np.array_equal(solo_own, ref[A_LOC:2 * A_LOC]))
"""


@pytest.mark.own_device  # (a child process holding three planners' streams)
def test_three_co_simulated_ranks_match_the_single_process_flight(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOGM_REPO=root)
    r = subprocess.run([sys.executable, "-c", _CHILD], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "rank share ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
