// replan_oracle.cpp — one full FakeBaselinePlanner::replan / BaselinePlanner::replan for one agent
// (plan_manager/src/baseline_fake.cpp:266-472, baseline.cpp:253-453) minus the GLPK deconfliction
// step (isSafeAfterOpt).  TEST INFRASTRUCTURE ONLY (see oracle.h).
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

extern "C" int orc_replan(const SogmSpec *s, const SogmAstarParams *ap, const SogmPlannerParams *pp,
                          const SogmQpSettings *qs, const float *grid, const float pose[3],
                          double stamp, const double start_pva[9], const double goal[3],
                          double t_start, int drone_id, SogmTrajRecord *rec, int stage_fail[1]) {
  std::memset(rec, 0, sizeof(*rec));
  rec->drone_id   = drone_id;
  rec->time_start = t_start;
  stage_fail[0]   = 0;
  // ---- search (baseline_fake.cpp:279-299)
  const double        t_after_map = t_start - stamp;
  std::vector<double> route(64 * 6);
  int                 route_len = 0, stats[4], ntr = 0;
  SogmAstarParams apv   = *ap;  // BaselinePlanner: RiskHybridAstar; FakeBaselinePlanner: FakeRiskHybridAstar
  apv.shot_ignores_time = pp->fake_planner ? 0 : 1;
  ap                    = &apv;
  const int ret = orc_astar_search(s, ap, grid, pose, start_pva, goal, t_after_map,
                                   pp->corridor_tau, route.data(), &route_len, 64, stats, nullptr,
                                   0, &ntr);
  if (ret == 0) {
    stage_fail[0] = 1;
    return 0;
  }
  // ---- corridors
  const int           MF = pp->max_faces;
  std::vector<double> polys((size_t)SOGM_MAX_PIECES * MF * 4);
  int                 nfaces[SOGM_MAX_PIECES];
  double              goal_pv[6];
  const int npoly = orc_corridor_generate(s, pp, grid, pose, stamp, start_pva, t_start,
                                          route.data(), route_len, polys.data(), nfaces, goal_pv);
  if (npoly <= 0) {
    stage_fail[0] = 2;
    return 0;
  }
  // ---- QP (baseline_fake.cpp:420-450)
  double t_alloc[SOGM_MAX_PIECES];
  for (int i = 0; i < npoly; ++i) t_alloc[i] = pp->corridor_tau;
  double final_state[9] = {goal_pv[0], goal_pv[1], goal_pv[2], goal_pv[3], goal_pv[4],
                           goal_pv[5], 0, 0, 0};
  std::vector<double> x((size_t)15 * npoly);
  int                 iters = 0;
  const int st = orc_qp_solve(start_pva, final_state, t_alloc, npoly, polys.data(), nfaces, MF,
                              pp->opt_max_vel, pp->opt_max_acc, qs, x.data(), &iters);
  if (!(st == 1 || st == 2)) {
    stage_fail[0] = 3;
    return 0;
  }
  rec->n_pieces = npoly;
  for (int i = 0; i < npoly; ++i) rec->duration[i] = t_alloc[i];
  for (int i = 0; i < 15 * npoly; ++i) rec->cpts[i] = x[i];
  stage_fail[0] = -iters;  // success: report -iterations for diagnostics
  return 1;
}
