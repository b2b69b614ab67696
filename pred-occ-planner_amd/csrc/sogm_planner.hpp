// sogm_planner.hpp — planner context + stage launchers shared by the planner translation units.
#pragma once

#include "sogm_device.hpp"

namespace sogm {

// Per-agent A* scratch in HBM (L2-resident while a search runs).
struct AstarWorkspace {
  char  *pool;         // [A][pool_stride] bytes, Node records
  size_t pool_stride;  // bytes per agent
  int   *heap;         // [A][allocate_num]
  void  *hkeys;        // [A][hash_cap] int4
  int   *hvals;        // [A][hash_cap]
  int    hash_cap;     // power of two >= 2 * allocate_num
};

size_t astar_node_bytes();
int    launch_astar(const MapView &m, const SogmAstarParams &ap, double corridor_tau,
                    const AstarWorkspace &wsp, int n_agents, const double *start_pva,
                    const double *goal, const double *t_start, int32_t *out_ret,
                    double *out_route, int32_t *out_route_len, int route_cap, int32_t *out_stats,
                    int32_t *out_trace, int trace_cap, hipStream_t st);

}  // namespace sogm

struct sogm_planner {
  sogm_ctx            *map;
  SogmAstarParams      ap;
  SogmPlannerParams    pp;
  SogmQpSettings       qs;
  sogm::AstarWorkspace aw;
  // internal buffers used by sogm_replan (device)
  int32_t *d_ret, *d_route_len, *d_stats;
  double  *d_route;
  int      route_cap;
};
