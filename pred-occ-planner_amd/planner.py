"""Host-side mirror of the planner classes for a batch of agents (over the C ABI).

`SogmPlanner` plays the role of FakeBaselinePlanner / BaselinePlanner
(plan_manager/src/baseline_fake.cpp, baseline.cpp): `search` = FakeRiskHybridAstar::search +
getPathWithVel, `generateCorridors` = the FIRI corridor stage, `optimize` = BezierOpt::setup +
optimize, `replan` = the whole BaselinePlanner::replan.  PyTorch only provides device buffers and
the stream; all compute is in libsogm_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._abi import SOGM_MAX_PIECES, check, lib
from .sogm import _dev, _stream


def linprog_batched(c, A_rows, b_rows, row_range):
    """sdlp::linprog<d> (traj_utils/include/traj_utils/sdlp.hpp:709-787) for a batch of LPs  min c.x, A x <= b.
    c [n, d] (d = 3 | 4), A_rows [total, d], b_rows [total], row_range [n, 2] int32 — device tensors.
    Returns (x [n, d], minimum [n]); +inf infeasible, -inf unbounded, NaN over capacity (152 rows)."""
    n, d = c.shape
    x = torch.zeros((n, d), dtype=torch.float64, device=c.device)
    v = torch.zeros((n,), dtype=torch.float64, device=c.device)
    check(lib().sogm_linprog_batched(d, c.data_ptr(), A_rows.data_ptr(), b_rows.data_ptr(), row_range.data_ptr(),
                                     n, x.data_ptr(), v.data_ptr(), _stream()), "sogm_linprog_batched")
    return x, v


def firi_batched(bd, pc, pc_range, a, b, r=None, iterations=2, epsilon=1.0e-6, max_points=4096, max_faces=64):
    """firi::firi (plan_manager/include/sfc_gen/firi.hpp:238-365) for n independent problems.
    bd [n, n_bd, 4], pc [total, 3], pc_range [n, 2] int32, a / b [n, 3] — device float64 tensors; r [n, 3] in/out
    (ones if None).  Returns (hpoly [n, max_faces, 4], nfaces [n], status [n], r [n, 3])."""
    n, n_bd = int(bd.shape[0]), int(bd.shape[1])
    dev = bd.device
    if r is None:
        r = torch.ones((n, 3), dtype=torch.float64, device=dev)
    hp = torch.zeros((n, max_faces, 4), dtype=torch.float64, device=dev)
    nf = torch.zeros((n,), dtype=torch.int32, device=dev)
    st = torch.zeros((n,), dtype=torch.int32, device=dev)
    check(lib().sogm_firi_batched(bd.data_ptr(), n_bd, pc.data_ptr(), pc_range.data_ptr(), a.data_ptr(), b.data_ptr(),
                                  r.data_ptr(), iterations, epsilon, n, max_points, max_faces, hp.data_ptr(),
                                  nf.data_ptr(), st.data_ptr(), _stream()), "sogm_firi_batched")
    return hp, nf, st, r


class SogmPlanner:
    def __init__(self, sogm_map, astar_params, planner_params, qp_settings):
        self.map = sogm_map
        self.ap, self.pp, self.qs = astar_params, planner_params, qp_settings
        self.A = sogm_map.n_agents
        self._p = C.c_void_p()
        check(lib().sogm_planner_create(sogm_map.ctx, C.byref(astar_params), C.byref(planner_params),
                                        C.byref(qp_settings), C.byref(self._p)), "sogm_planner_create")

    def close(self):
        if self._p:
            torch.cuda.synchronize()
            lib().sogm_planner_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def select_agents(self, first=0, count=None):
        """Per-stage entries (search / generateCorridors / optimize / isSafeAfterOpt) then process agents
        [first, first + count) only; replan() always processes all."""
        check(lib().sogm_planner_select_agents(self._p, first, self.A if count is None else count),
              "sogm_planner_select_agents")

    def set_search_mode(self, mode):
        """0 the replan's two-call pattern, 1 / 2 one RiskHybridAstar::search with init_search true / false."""
        check(lib().sogm_planner_set_search_mode(self._p, mode), "sogm_planner_set_search_mode")

    def flow_error(self):
        """0 if the last replan()'s dataflow kernels completed normally (synchronises the device)."""
        return int(lib().sogm_planner_flow_error(self._p))

    def flow_failures(self):
        """(code of the most recent failed tick, number of failed replan() calls) as of the ticks completed on the
        device — no synchronisation (sogm_planner_flow_failures)."""
        out = (C.c_int32 * 2)()
        check(lib().sogm_planner_flow_failures(self._p, out), "sogm_planner_flow_failures")
        return int(out[0]), int(out[1])

    def counters(self, reset=False):
        """Cumulative outcome / capacity counters of replan() (sogm_planner_counters) as a dict."""
        out = (C.c_int64 * len(_abi.COUNTER_NAMES))()
        check(lib().sogm_planner_counters(self._p, out, 1 if reset else 0), "sogm_planner_counters")
        return dict(zip(_abi.COUNTER_NAMES, [int(v) for v in out]))

    # ---- FakeRiskHybridAstar::search + getPathWithVel ----
    def search(self, start_pva, goal, t_start, route_cap=64, trace_cap=0):
        A, dev = self.A, start_pva.device
        out = {
            "ret": torch.zeros((A,), dtype=torch.int32, device=dev),
            "route": torch.zeros((A, route_cap, 6), dtype=torch.float64, device=dev),
            "route_len": torch.zeros((A,), dtype=torch.int32, device=dev),
            "stats": torch.zeros((A, 4), dtype=torch.int32, device=dev),
            "trace": torch.full((A, max(trace_cap, 1)), -1, dtype=torch.int32, device=dev),
        }
        check(lib().sogm_astar_search(self._p, start_pva.data_ptr(), goal.data_ptr(), t_start.data_ptr(),
                                      out["ret"].data_ptr(), out["route"].data_ptr(),
                                      out["route_len"].data_ptr(), route_cap, out["stats"].data_ptr(),
                                      out["trace"].data_ptr() if trace_cap else None, trace_cap, _stream()),
              "sogm_astar_search")
        return out

    # ---- corridor stage ----
    def generateCorridors(self, start_pva, t_start, route, route_len):
        A, dev = self.A, start_pva.device
        mf = self.pp.max_faces
        out = {
            "polys": torch.zeros((A, SOGM_MAX_PIECES, mf, 4), dtype=torch.float64, device=dev),
            "nfaces": torch.zeros((A, SOGM_MAX_PIECES), dtype=torch.int32, device=dev),
            "npoly": torch.zeros((A,), dtype=torch.int32, device=dev),
            "goal": torch.zeros((A, 6), dtype=torch.float64, device=dev),
        }
        check(lib().sogm_corridor_generate(self._p, start_pva.data_ptr(), t_start.data_ptr(),
                                           route.data_ptr(), route_len.data_ptr(), int(route.shape[1]),
                                           out["polys"].data_ptr(), out["nfaces"].data_ptr(),
                                           out["npoly"].data_ptr(), out["goal"].data_ptr(), _stream()),
              "sogm_corridor_generate")
        return out

    # ---- BezierOpt::setup + optimize ----
    def optimize(self, start_pva, goal_pv, polys, nfaces, npoly):
        A, dev = self.A, start_pva.device
        out = {
            "cpts": torch.zeros((A, SOGM_MAX_PIECES * 15), dtype=torch.float64, device=dev),
            "status": torch.zeros((A,), dtype=torch.int32, device=dev),
            "iters": torch.zeros((A,), dtype=torch.int32, device=dev),
        }
        check(lib().sogm_bezier_qp_solve(self._p, start_pva.data_ptr(), goal_pv.data_ptr(),
                                         polys.data_ptr(), nfaces.data_ptr(), npoly.data_ptr(),
                                         out["cpts"].data_ptr(), out["status"].data_ptr(),
                                         out["iters"].data_ptr(), _stream()), "sogm_bezier_qp_solve")
        return out

    def optimize_timed(self, start_pva, end_pva, time_alloc, polys, nfaces, npoly, max_vel, max_acc):
        """BezierOpt::setup(start, end, time_allocation, constraints, max_vel, max_acc) + optimize in full: any time
        allocation per piece [A, 16], an end state with acceleration [A, 9], the caller's limits."""
        A, dev = self.A, start_pva.device
        out = {
            "cpts": torch.zeros((A, SOGM_MAX_PIECES * 15), dtype=torch.float64, device=dev),
            "status": torch.zeros((A,), dtype=torch.int32, device=dev),
            "iters": torch.zeros((A,), dtype=torch.int32, device=dev),
        }
        check(lib().sogm_bezier_qp_solve_timed(self._p, start_pva.data_ptr(), end_pva.data_ptr(), time_alloc.data_ptr(),
                                               float(max_vel), float(max_acc), polys.data_ptr(), nfaces.data_ptr(),
                                               npoly.data_ptr(), out["cpts"].data_ptr(), out["status"].data_ptr(),
                                               out["iters"].data_ptr(), _stream()), "sogm_bezier_qp_solve_timed")
        return out

    # ---- ParticleATC::isSafeAfterOpt ----
    def isSafeAfterOpt(self, cpts, npoly, records, n_records, ego_ids, t_now):
        safe = torch.empty((self.A,), dtype=torch.int32, device=cpts.device)
        check(lib().sogm_safe_after_opt(self._p, cpts.data_ptr(), npoly.data_ptr(), records.data_ptr(), n_records,
                                        ego_ids.data_ptr(), t_now.data_ptr(), safe.data_ptr(), _stream()),
              "sogm_safe_after_opt")
        return safe

    def setSwarm(self, records, n_records, ego_ids, t_now):
        """replan() then ends with isSafeAfterOpt against `records` (baseline_fake.cpp:453-460); the tensors
        are read by later replan() calls and are kept alive here.  records=None switches the check off."""
        self._swarm = (records, ego_ids, t_now)
        if records is None:
            check(lib().sogm_planner_set_swarm(self._p, None, 0, None, None), "sogm_planner_set_swarm")
        else:
            check(lib().sogm_planner_set_swarm(self._p, records.data_ptr(), n_records, ego_ids.data_ptr(),
                                               t_now.data_ptr()), "sogm_planner_set_swarm")

    def setPublish(self, own_records, next_table=None):
        """replan() then also merges every successful record into `own_records` and writes each agent's current
        record into `next_table` (sogm_planner_set_publish); None switches it off.  Tensors are kept alive here."""
        self._publish = (own_records, next_table)
        check(lib().sogm_planner_set_publish(self._p, own_records.data_ptr() if own_records is not None else None,
                                             next_table.data_ptr() if next_table is not None else None),
              "sogm_planner_set_publish")

    def setPrestamp(self, cloud, cloud_range, cylinders, n_cyl, next_stamp, start_offset, hover, now, t_start, pva,
                    poses=None, world=None):
        """The next replan() also builds the next tick's map and start states (sogm_planner_set_prestamp); the device
        tensors are kept alive here.  `world` (sogm.World): the stamp's inputs are that frame, cropped on the device
        (cloud / cloud_range / cylinders are then ignored).  cloud=None and world=None switches it off."""
        if cloud is None and world is None:
            self._prestamp = None
            check(lib().sogm_planner_set_prestamp(self._p, None), "sogm_planner_set_prestamp")
            return
        ptr = lambda t: t.data_ptr() if t is not None else None
        ps = _abi.SogmPrestamp(ptr(cloud), ptr(cloud_range), ptr(cylinders), int(n_cyl or 0), 0,
                               float(next_stamp), float(start_offset), hover.data_ptr(), now.data_ptr(),
                               t_start.data_ptr(), pva.data_ptr(), ptr(poses),
                               C.pointer(world.c) if world is not None else None)
        self._prestamp = (cloud, cloud_range, cylinders, hover, now, t_start, pva, poses, world)
        check(lib().sogm_planner_set_prestamp(self._p, C.byref(ps)), "sogm_planner_set_prestamp")

    # ---- sogm_flight_run: n ticks of every agent, each on its own clock ----
    def flight(self, worlds, first_tick, t0, period, start_offset, goals, drone_ids, hover, own, tables, log_records,
               log_ok, n_total=None, agent0=0, comm=None):
        """n = len(worlds) replan ticks of every agent in one call (sogm_abi.h "Flight"): `worlds` sogm.World frames of
        ticks first_tick .. first_tick + n - 1, `tables` uint8 [4, n_total, 2064] (ver(j) at j & 3; this planner's agents
        are rows agent0 .. agent0 + A; with n_total > A the other rows are the caller's to fill between calls and n <= 2),
        `log_records` uint8 [n, A, 2064], `log_ok` int32 [n, A].  `comm` (an ncclComm_t handle, e.g. sogm_comm_handle):
        several ranks with the exchange behind the call — no limit of two ticks, every rank passes the same n.
        Asynchronous on the current stream."""
        n = len(worlds)
        arr = (_abi.SogmWorld * n)(*[w.c for w in worlds])
        f = _abi.SogmFlight(n, int(first_tick), float(t0), float(period), float(start_offset), arr, goals.data_ptr(),
                            drone_ids.data_ptr(), hover.data_ptr(), own.data_ptr(), tables.data_ptr(),
                            self.A if n_total is None else int(n_total), int(agent0),
                            log_records.data_ptr(), log_ok.data_ptr(), comm)
        self._flight_keep = (worlds, arr, goals, drone_ids, hover, own, tables, log_records, log_ok)
        check(lib().sogm_flight_run(self._p, C.byref(f), _stream()), "sogm_flight_run")

    def flight_prepare(self, max_cloud_points=0):
        """the flight's streams, their hardware queues and its buffers (crop lists for frames of up to max_cloud_points points),
        created now (sogm_flight_prepare): before several planners of one process fly at once"""
        check(lib().sogm_flight_prepare(self._p, int(max_cloud_points)), "sogm_flight_prepare")

    def flight_stats(self):
        """(per-agent sums [A, 8] in ms: _abi.FLIGHT_STAT_NAMES, control counters [32]) of the last flight; synchronises"""
        ms = np.zeros((self.A, 8), np.float64)
        hdr = np.zeros((32,), np.int32)
        check(lib().sogm_flight_stats(self._p, ms.ctypes.data_as(C.c_void_p), hdr.ctypes.data_as(C.c_void_p)),
              "sogm_flight_stats")
        return ms, hdr

    # ---- BaselinePlanner::replan ----
    def replan(self, start_pva, goal, t_start, drone_ids, out_records=None, out_ok=None):
        A, dev = self.A, start_pva.device
        if out_records is None:
            out_records = torch.zeros((A, _abi.TRAJ_RECORD_BYTES), dtype=torch.uint8, device=dev)
        if out_ok is None:
            out_ok = torch.zeros((A,), dtype=torch.int32, device=dev)
        check(lib().sogm_replan(self._p, start_pva.data_ptr(), goal.data_ptr(), t_start.data_ptr(),
                                drone_ids.data_ptr(), out_records.data_ptr(), out_ok.data_ptr(), _stream()),
              "sogm_replan")
        return out_records, out_ok


def records_from_bytes(buf):
    """uint8 numpy [A, 2064] -> ctypes array of SogmTrajRecord."""
    n = buf.shape[0]
    arr = (_abi.SogmTrajRecord * n).from_buffer_copy(np.ascontiguousarray(buf).tobytes())
    return arr


def traj_eval(records, t):
    """Bezier::getPos/getVel/getAcc for a batch of records (device uint8 [n, 2064]) at absolute times
    t (device float64 [n]).  Returns (pva [n, 9], valid [n])."""
    n = int(t.numel())
    pva = torch.empty((n, 9), dtype=torch.float64, device=t.device)
    ok = torch.empty((n,), dtype=torch.int32, device=t.device)
    check(lib().sogm_traj_eval(records.data_ptr(), n, t.data_ptr(), pva.data_ptr(), ok.data_ptr(), _stream()),
          "sogm_traj_eval")
    return pva, ok


def smoke_check(pop, sogm, orc, spec, sc, dev, m):
    """Used by __graft_entry__.smoke(): one replan of the tiny scene vs the oracle."""
    A = sc["n_agents"]
    ap, pp, qs = pop.config.make_astar_params(), pop.config.make_planner_params(True), pop.config.make_qp_settings()
    P = SogmPlanner(m, ap, pp, qs)
    pva = np.concatenate([sc["starts"], np.zeros((A, 6))], axis=1)
    t_start = sc["stamps"] + 0.02
    rec_d, ok_d = P.replan(_dev(pva, np.float64), _dev(sc["goals"], np.float64), _dev(t_start, np.float64),
                           dev["ego_ids"])
    got = records_from_bytes(rec_d.cpu().numpy())
    ok = ok_d.cpu().numpy()
    cyl = pop.scene.cylinders_to_struct(sc["cylinders"])
    recs = pop.scene.straight_records(sc)
    for a in range(A):
        g = orc.update_gt(spec, sc["cloud"], cyl, dev["n_cyl"], sc["poses"][a])
        orc.project_neighbours(spec, g, recs, A, a, m.body, sc["poses"][a], sc["stamps"][a])
        w_ok, w, _ = orc.replan(spec, ap, pp, qs, g, sc["poses"][a], sc["stamps"][a], pva[a], sc["goals"][a],
                                t_start[a], a)
        assert ok[a] == w_ok and got[a].n_pieces == w.n_pieces, "replan outcome differs from the oracle"
        k = w.n_pieces * 15
        assert np.allclose(np.array(got[a].cpts[:k]), np.array(w.cpts[:k]), atol=1e-4, rtol=0)
    P.close()
