// sogm_planner.hip — planner context (search + corridors + QP).  Stage kernels are being brought
// up one at a time; entry points not yet wired return SOGM_ERR_STATE.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "sogm_device.hpp"

struct sogm_planner {
  sogm_ctx         *map;
  SogmAstarParams   ap;
  SogmPlannerParams pp;
  SogmQpSettings    qs;
};

extern "C" {

int sogm_planner_create(sogm_ctx *map, const SogmAstarParams *astar, const SogmPlannerParams *pp,
                        const SogmQpSettings *qp, sogm_planner **out) {
  if (!map || !astar || !pp || !qp || !out) return SOGM_ERR_INVALID_ARG;
  sogm_planner *p = new (std::nothrow) sogm_planner();
  if (!p) return SOGM_ERR_INVALID_ARG;
  p->map = map;
  p->ap  = *astar;
  p->pp  = *pp;
  p->qs  = *qp;
  *out   = p;
  return SOGM_OK;
}
void sogm_planner_destroy(sogm_planner *p) { delete p; }

int sogm_astar_search(sogm_planner *, const double *, const double *, const double *, int32_t *,
                      double *, int32_t *, int, int32_t *, int32_t *, int, void *) {
  return SOGM_ERR_STATE;
}
int sogm_corridor_generate(sogm_planner *, const double *, const double *, const double *,
                           const int32_t *, int, double *, int32_t *, int32_t *, double *, void *) {
  return SOGM_ERR_STATE;
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
