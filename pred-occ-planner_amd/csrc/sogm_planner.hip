// sogm_planner.hip — planner context (search + corridors + QP).  Stage kernels are being brought
// up one at a time; entry points not yet wired return SOGM_ERR_STATE.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "sogm_planner.hpp"

using namespace sogm;

extern "C" {

int sogm_planner_create(sogm_ctx *map, const SogmAstarParams *astar, const SogmPlannerParams *pp,
                        const SogmQpSettings *qp, sogm_planner **out) {
  if (!map || !astar || !pp || !qp || !out) return SOGM_ERR_INVALID_ARG;
  sogm_planner *p = new (std::nothrow) sogm_planner();
  if (!p) return SOGM_ERR_INVALID_ARG;
  std::memset(p, 0, sizeof(*p));
  p->map = map;
  p->ap  = *astar;
  p->pp  = *pp;
  p->qs  = *qp;
  if (astar->allocate_num < 2 || astar->check_num < 1 || !(astar->resolution > 0) ||
      !(astar->time_resolution > 0)) {
    delete p;
    return SOGM_ERR_INVALID_ARG;
  }
  const int A = map->n_agents;
  int       hc = 1;
  while (hc < 2 * astar->allocate_num) hc <<= 1;
  p->aw.hash_cap    = hc;
  p->aw.pool_stride = astar_node_bytes() * (size_t)astar->allocate_num;
  p->route_cap      = 64;
  hipError_t e      = hipMalloc((void **)&p->aw.pool, p->aw.pool_stride * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->aw.heap, sizeof(int) * (size_t)astar->allocate_num * A);
  if (e == hipSuccess) e = hipMalloc(&p->aw.hkeys, 16 * (size_t)hc * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->aw.hvals, sizeof(int) * (size_t)hc * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_ret, sizeof(int32_t) * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_route_len, sizeof(int32_t) * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_stats, sizeof(int32_t) * 4 * A);
  if (e == hipSuccess) e = hipMalloc((void **)&p->d_route, sizeof(double) * 6 * p->route_cap * A);
  {
    if (pp->max_faces < 6 || pp->max_faces > 64 || pp->pc_capacity < 1 || pp->pc_capacity > 16384 ||
        pp->firi_iterations < 1) {
      sogm_planner_destroy(p);
      return SOGM_ERR_INVALID_ARG;
    }
    const size_t slots = (size_t)A * SOGM_MAX_PIECES, cap = (size_t)pp->pc_capacity;
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.pc, sizeof(double) * slots * cap * 3);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.fpc, sizeof(double) * slots * cap * 3);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.tang, sizeof(double) * slots * cap * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.distr, sizeof(double) * slots * cap);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.polys, sizeof(double) * slots * pp->max_faces * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.seg_nfaces, sizeof(int32_t) * slots);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.seg_state, sizeof(int32_t) * slots);
    if (e == hipSuccess) e = hipMalloc((void **)&p->cw.seg_npts, sizeof(int32_t) * slots);
  }
  if (e != hipSuccess) {
    sogm::set_error("sogm_planner_create", e);
    sogm_planner_destroy(p);
    return SOGM_ERR_HIP;
  }
  *out = p;
  return SOGM_OK;
}
void sogm_planner_destroy(sogm_planner *p) {
  if (!p) return;
  void *ptrs[] = {p->aw.pool, p->aw.heap, p->aw.hkeys, p->aw.hvals,
                  p->d_ret,   p->d_route_len, p->d_stats, p->d_route,
                  p->cw.pc,   p->cw.fpc,  p->cw.tang, p->cw.distr, p->cw.polys,
                  p->cw.seg_nfaces, p->cw.seg_state, p->cw.seg_npts};
  for (void *q : ptrs)
    if (q) (void)hipFree(q);
  delete p;
}

int sogm_astar_search(sogm_planner *p, const double *start_pva, const double *goal,
                      const double *t_start, int32_t *out_ret, double *out_route,
                      int32_t *out_route_len, int route_cap, int32_t *out_stats,
                      int32_t *out_trace, int trace_cap, void *stream) {
  if (!p || !start_pva || !goal || !t_start || !out_ret || !out_route || !out_route_len ||
      !out_stats || route_cap < 2)
    return SOGM_ERR_INVALID_ARG;
  if (!p->map->updated) return SOGM_ERR_STATE;
  hipStream_t st = (hipStream_t)stream;
  prof_begin(p->map, SOGM_PROF_ASTAR, st);
  int rc = launch_astar(view_of(p->map), p->ap, p->pp.corridor_tau, p->aw, p->map->n_agents,
                        start_pva, goal, t_start, out_ret, out_route, out_route_len, route_cap,
                        out_stats, out_trace, out_trace ? trace_cap : 0, st);
  prof_end(p->map, SOGM_PROF_ASTAR, st);
  if (rc) {
    sogm::set_error("k_astar", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  return SOGM_OK;
}
int sogm_corridor_generate(sogm_planner *p, const double *start_pva, const double *t_start,
                           const double *route, const int32_t *route_len, int route_cap,
                           double *out_polys, int32_t *out_nfaces, int32_t *out_npoly,
                           double *out_goal, void *stream) {
  if (!p || !start_pva || !t_start || !route || !route_len || !out_polys || !out_nfaces ||
      !out_npoly || !out_goal || route_cap < 2)
    return SOGM_ERR_INVALID_ARG;
  if (!p->map->updated) return SOGM_ERR_STATE;
  hipStream_t st = (hipStream_t)stream;
  prof_begin(p->map, SOGM_PROF_CORRIDOR, st);
  int rc = launch_corridor(view_of(p->map), p->pp, p->cw, p->map->n_agents, start_pva, t_start,
                           route, route_len, route_cap, out_polys, out_nfaces, out_npoly, out_goal,
                           st);
  prof_end(p->map, SOGM_PROF_CORRIDOR, st);
  if (rc) {
    sogm::set_error("k_corridor", hipGetLastError());
    return SOGM_ERR_HIP;
  }
  return SOGM_OK;
}
int sogm_bezier_qp_solve(sogm_planner *, const double *, const double *, const double *,
                         const int32_t *, const int32_t *, double *, int32_t *, int32_t *, void *) {
  return SOGM_ERR_STATE;
}
int sogm_replan(sogm_planner *, const double *, const double *, const double *, const int32_t *,
                SogmTrajRecord *, int32_t *, void *) {
  return SOGM_ERR_STATE;
}
}
