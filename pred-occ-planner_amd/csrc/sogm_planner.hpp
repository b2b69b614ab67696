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

// Per-(agent, segment) corridor scratch in HBM.
struct CorridorWorkspace {
  double  *pc;          // [A*P][cap][3] obstacle points
  double  *fpc;         // [A*P][cap][3] points in the ellipsoid frame
  double  *tang;        // [A*P][cap][4] tangent planes
  double  *distr;       // [A*P][cap]
  double  *polys;       // [A*P][max_faces][4] shrunk polytopes
  int32_t *seg_nfaces;  // [A*P]
  int32_t *seg_state;   // [A*P] 1 valid, 0 invalid, -2 no segment, -3 capacity exceeded
  int32_t *seg_npts;    // [A*P]
};

int launch_corridor(const MapView &m, const SogmPlannerParams &pp, const CorridorWorkspace &ws,
                    int n_agents, const double *start_pva, const double *t_start,
                    const double *route, const int32_t *route_len, int route_cap,
                    double *out_polys, int32_t *out_nfaces, int32_t *out_npoly, double *out_goal,
                    hipStream_t st);

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
  sogm::CorridorWorkspace cw;
  // internal buffers used by sogm_replan (device)
  int32_t *d_ret, *d_route_len, *d_stats;
  double  *d_route;
  int      route_cap;
};
