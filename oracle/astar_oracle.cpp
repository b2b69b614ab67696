// astar_oracle.cpp — CPU restatement of FakeRiskHybridAstar.  TEST INFRASTRUCTURE ONLY (oracle.h).
//
// Follows path_searching/src/fake_risk_hybrid_a_star.cpp:
//   search :115-426, estimateHeuristic :428-468, computeShotTraj :470-523, cubic/quartic :525-587,
//   getPathWithVel :663-694, posToIndex/timeToIndex :795-803, stateTransit :812-824,
//   retrievePath :826-836;  node/hash/heap types path_node.h:37-97, grid_node.h:10-51.
// (risk_hybrid_a_star.cpp is the same algorithm; its shot check drops the time argument, :514 —
//  selected with SogmAstarParams.shot_ignores_time.)
//
// Quirks kept on purpose:
//   * hash insert uses (int)pro_node->time while find uses time_idx (:387 vs :271);
//   * unordered_map::insert does not overwrite an existing key (path_node.h:79-82);
//   * f-scores of nodes already in the heap are edited in place without re-heapifying (:349-352);
//   * std::priority_queue == libstdc++ push_heap/pop_heap on a vector — used directly here, the
//     container the reference runs on; the HIP side restates that algorithm;
//   * the shot-trajectory collision check uses shot-relative time (:514).
// Parity unpinned: no reference test covers A*.
//
// cbrt/acos/cos come from include/sogm_detmath.h (deterministic, <= 3 ulp from libm) so that the
// HIP path can be bit-exact; orc_astar_use_libm(1) switches to libm (what the reference calls) —
// tests check that the expansion trace is identical under both on the test scenes.
#include <algorithm>
#include <cmath>
#include <map>
#include <queue>
#include <vector>

#include "../include/sogm_detmath.h"
#include "oracle.h"

int orc_g_use_libm = 0;  // shared with corridor_oracle.cpp

namespace {

#define g_use_libm orc_g_use_libm

double f_cbrt(double x) { return g_use_libm ? std::cbrt(x) : sogm_det::cbrt(x); }
double f_acos(double x) { return g_use_libm ? std::acos(x) : sogm_det::acos(x); }
double f_cos(double x) { return g_use_libm ? std::cos(x) : sogm_det::cos(x); }

enum { IN_CLOSE_SET = 1, IN_OPEN_SET = 2, NOT_EXPAND = 3 };
enum { NO_PATH = 0, INIT_ERR, SEARCH_ERR, REACH_HORIZON, REACH_END, NEAR_END };

struct Node {
  double state[6];
  double input[3];
  double duration;
  double time;
  int    time_idx;
  int    index[3];
  double g, f;
  int    parent;
  int    node_state;
};

struct Search {
  const SogmSpec        *spec;
  const SogmAstarParams *ap;
  const float           *grid;
  const float           *pose;
  std::vector<Node>      pool;
  int                    use_node_num = 0, iter_num = 0;
  double                 map_center[3];
  double                 inv_resolution, inv_time_resolution, time_origin = 0;
  double                 tie_breaker;
  std::vector<int>       node_path;
  bool                   is_shot_succ = false;
  std::vector<int>      *trace        = nullptr;

  struct Cmp {
    const std::vector<Node> *pool;
    bool operator()(int a, int b) const { return (*pool)[a].f > (*pool)[b].f; }  // grid_node.h:46-51
  };
  std::map<std::array<int, 4>, int> expanded;  // NodeHashTable::data_4d_ semantics (no overwrite)

  void posToIndex(const double p[3], int out[3]) const {  // :795-798
    for (int i = 0; i < 3; ++i) out[i] = (int)std::floor((p[i] - map_center[i]) * inv_resolution);
  }
  int timeToIndex(double t) const {  // :800-803
    return (int)std::floor((t - time_origin) * inv_time_resolution);
  }
  // :812-824  state1 = phi * state0 + integral
  static void stateTransit(const double s0[6], double s1[6], const double um[3], double tau) {
    const double h = 0.5 * (tau * tau);  // 0.5 * pow(tau, 2)
    for (int i = 0; i < 3; ++i) {
      s1[i]     = (s0[i] + tau * s0[i + 3]) + h * um[i];
      s1[i + 3] = s0[i + 3] + tau * um[i];
    }
  }
  int query(const double pos[3], double t) const {
    return orc_query_clear_time(spec, grid, pose, pos, t);
  }

  // :525-551
  static int cubic(double a, double b, double c, double d, double out[3]) {
    const double a2 = b / a, a1 = c / a, a0 = d / a;
    const double Q = (3 * a1 - a2 * a2) / 9;
    const double R = (9 * a1 * a2 - 27 * a0 - 2 * a2 * a2 * a2) / 54;
    const double D = Q * Q * Q + R * R;
    if (D > 0) {
      const double S = f_cbrt(R + std::sqrt(D));
      const double T = f_cbrt(R - std::sqrt(D));
      out[0]         = -a2 / 3 + (S + T);
      return 1;
    } else if (D == 0) {
      const double S = f_cbrt(R);
      out[0]         = -a2 / 3 + S + S;
      out[1]         = -a2 / 3 - S;
      return 2;
    } else {
      const double theta = f_acos(R / std::sqrt(-Q * Q * Q));
      out[0]             = 2 * std::sqrt(-Q) * f_cos(theta / 3) - a2 / 3;
      out[1]             = 2 * std::sqrt(-Q) * f_cos((theta + 2 * M_PI) / 3) - a2 / 3;
      out[2]             = 2 * std::sqrt(-Q) * f_cos((theta + 4 * M_PI) / 3) - a2 / 3;
      return 3;
    }
  }
  // :553-587
  static int quartic(double a, double b, double c, double d, double e, double out[4]) {
    const double a3 = b / a, a2 = c / a, a1 = d / a, a0 = e / a;
    double       ys[3];
    cubic(1, -a2, a1 * a3 - 4 * a0, 4 * a2 * a0 - a1 * a1 - a3 * a3 * a0, ys);
    const double y1 = ys[0];
    const double r  = a3 * a3 / 4 - a2 + y1;
    if (r < 0) return 0;
    const double R = std::sqrt(r);
    double       D, E;
    if (R != 0) {
      D = std::sqrt(0.75 * a3 * a3 - R * R - 2 * a2 +
                    0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R);
      E = std::sqrt(0.75 * a3 * a3 - R * R - 2 * a2 -
                    0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R);
    } else {
      D = std::sqrt(0.75 * a3 * a3 - 2 * a2 + 2 * std::sqrt(y1 * y1 - 4 * a0));
      E = std::sqrt(0.75 * a3 * a3 - 2 * a2 - 2 * std::sqrt(y1 * y1 - 4 * a0));
    }
    int n = 0;
    if (!std::isnan(D)) {
      out[n++] = -a3 / 4 + R / 2 + D / 2;
      out[n++] = -a3 / 4 + R / 2 - D / 2;
    }
    if (!std::isnan(E)) {
      out[n++] = -a3 / 4 - R / 2 + E / 2;
      out[n++] = -a3 / 4 - R / 2 - E / 2;
    }
    return n;
  }
  static double dot3(const double *a, const double *b) {
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
  }
  // :428-468
  double estimateHeuristic(const double x1[6], const double x2[6], double &optimal_time) const {
    double dp[3], v0[3], v1[3], vs[3];
    for (int i = 0; i < 3; ++i) {
      dp[i] = x2[i] - x1[i];
      v0[i] = x1[i + 3];
      v1[i] = x2[i + 3];
      vs[i] = v0[i] + v1[i];
    }
    const double c1 = -36 * dot3(dp, dp);
    const double c2 = 24 * dot3(vs, dp);
    const double c3 = -4 * (dot3(v0, v0) + dot3(v0, v1) + dot3(v1, v1));
    const double c4 = 0;
    const double c5 = ap->w_time;
    double       ts[5];
    int          n      = quartic(c5, c4, c3, c2, c1, ts);
    const double v_max  = ap->max_vel * 0.5;
    double       linf   = 0;
    for (int i = 0; i < 3; ++i) linf = std::max(linf, std::fabs(x1[i] - x2[i]));
    const double t_bar = linf / v_max;
    ts[n++]            = t_bar;
    double cost = 100000000, t_d = t_bar;
    for (int i = 0; i < n; ++i) {
      const double t = ts[i];
      if (t < t_bar) continue;
      const double c = -c1 / (3 * t * t * t) - c2 / (2 * t * t) - c3 / t + ap->w_time * t;
      if (c < cost) {
        cost = c;
        t_d  = t;
      }
    }
    optimal_time = t_d;
    return 1.0 * (1 + tie_breaker) * cost;
  }
  // :470-523 (only the feasibility result matters to replan(): it selects the return code)
  bool computeShotTraj(const double s1[6], const double s2[6], double t_d) {
    double a[3], b[3], c[3], d[3];
    for (int i = 0; i < 3; ++i) {
      const double p0 = s1[i], dp = s2[i] - p0, v0 = s1[i + 3], v1 = s2[i + 3], dv = v1 - v0;
      a[i] = 1.0 / 6.0 * (-12.0 / (t_d * t_d * t_d) * (dp - v0 * t_d) + 6 / (t_d * t_d) * dv);
      b[i] = 0.5 * (6.0 / (t_d * t_d) * (dp - v0 * t_d) - 2 / t_d * dv);
      c[i] = v0;
      d[i] = p0;
    }
    const double t_delta = t_d / 10;
    // guard: t_d == 0 (node exactly on the goal) makes the reference's loop spin forever
    int guard = 0;
    for (double time = t_delta; time <= t_d && guard < 64; time += t_delta, ++guard) {
      // coord = d + c t + b t^2 + a t^3 ; powers as products (pow(t,3) restated as (t*t)*t)
      const double t1 = time, t2 = time * time, t3 = (time * time) * time;
      double       coord[3];
      for (int dim = 0; dim < 3; ++dim)
        coord[dim] = ((d[dim] * 1.0 + c[dim] * t1) + b[dim] * t2) + a[dim] * t3;
      // :521 (fake) passes the time; risk_hybrid_a_star.cpp:514 calls getClearOcccupancy(coord) = slice 0
      const int hit = ap->shot_ignores_time ? orc_query_clear_idx(spec, grid, pose, coord, 0) : query(coord, time);
      if (hit != 0) return false;
    }
    is_shot_succ = true;
    return true;
  }
  void retrievePath(int end_node) {  // :826-836
    int cur = end_node;
    node_path.push_back(cur);
    while (pool[cur].parent >= 0) {
      cur = pool[cur].parent;
      node_path.push_back(cur);
    }
    std::reverse(node_path.begin(), node_path.end());
  }

  // dynamic = false (risk_hybrid_a_star.cpp:153-158,271,287,350-358): the reference never writes PathNode::time /
  // time_idx in that branch and reads them all the same (collision sample times :324, exceed_time :177) —
  // uninitialised members of `new PathNode`, or the values a previous search left in the reused pool.  DEFINED here
  // (and in the HIP kernel, search_mode bit 4) as zero: every node's time and time index are 0, which makes the
  // 4-D table / prune / same-voxel tests coincide with the branch's 3-D ones and samples the SOGM at [0, tau].
  bool dynamic = true;
  int search(const double start_pt[3], const double start_v[3], const double start_a[3],
             const double end_pt[3], const double end_v[3], bool init, double time_start) {
    if (!dynamic) time_start = 0.0;
    std::priority_queue<int, std::vector<int>, Cmp> open_set(Cmp{&pool});
    for (int i = 0; i < 3; ++i) map_center[i] = (double)pose[i];  // :126
    Node &n0 = pool[0];
    n0.parent = -1;
    for (int i = 0; i < 3; ++i) {
      n0.state[i]     = start_pt[i];
      n0.state[i + 3] = start_v[i];
    }
    posToIndex(start_pt, n0.index);
    n0.g = 0.0;
    double end_state[6];
    int    end_index[3];
    double time_to_goal;
    for (int i = 0; i < 3; ++i) {
      end_state[i]     = end_pt[i];
      end_state[i + 3] = end_v[i];
    }
    posToIndex(end_pt, end_index);
    n0.f          = ap->lambda_heu * estimateHeuristic(n0.state, end_state, time_to_goal);
    n0.node_state = IN_OPEN_SET;
    open_set.push(0);
    use_node_num += 1;
    time_origin = time_start;
    n0.time     = time_start;
    n0.time_idx = timeToIndex(time_start);
    expanded.insert({{n0.index[0], n0.index[1], n0.index[2], n0.time_idx}, 0});

    bool init_search = init;
    const int tol    = ap->tolerance;
    while (!open_set.empty()) {
      const int cur = open_set.top();
      Node     &cn  = pool[cur];
      double    d3[3] = {cn.state[0] - start_pt[0], cn.state[1] - start_pt[1],
                         cn.state[2] - start_pt[2]};
      const bool reach_horizon = std::sqrt(dot3(d3, d3)) >= ap->horizon;
      const bool near_end      = std::abs(cn.index[0] - end_index[0]) <= tol &&
                            std::abs(cn.index[1] - end_index[1]) <= tol &&
                            std::abs(cn.index[2] - end_index[2]) <= tol;
      const bool exceed_time = cn.time >= ap->max_tau;
      if (reach_horizon || near_end || exceed_time) {
        retrievePath(cur);
        if (near_end) {
          estimateHeuristic(cn.state, end_state, time_to_goal);
          computeShotTraj(cn.state, end_state, time_to_goal);
        }
      }
      if (reach_horizon) return is_shot_succ ? REACH_END : REACH_HORIZON;
      if (near_end) {
        if (is_shot_succ) return REACH_END;
        if (cn.parent >= 0) return NEAR_END;
        return NO_PATH;
      }
      if (exceed_time) return REACH_HORIZON;

      open_set.pop();
      cn.node_state = IN_CLOSE_SET;
      iter_num += 1;
      if (trace) trace->push_back(cur);

      const double res = 1 / 2.0;
      double       cur_state[6];
      for (int i = 0; i < 6; ++i) cur_state[i] = cn.state[i];
      std::vector<int>                   tmp_expand_nodes;
      std::vector<std::array<double, 3>> inputs;
      const double                       tau_fixed = ap->time_resolution;
      if (init_search) {
        inputs.push_back({start_a[0], start_a[1], start_a[2]});
        init_search = false;
      } else {
        const double ma = ap->max_acc;
        for (double ax = -ma; ax <= ma + 1e-3; ax += ma * res)
          for (double ay = -ma; ay <= ma + 1e-3; ay += ma * res)
            for (double az = -0.5 * ma; az <= 0.5 * ma + 1e-3; az += ma * res)
              inputs.push_back({ax, ay, az});
      }
      for (size_t i = 0; i < inputs.size(); ++i) {
        const double um[3] = {inputs[i][0], inputs[i][1], inputs[i][2]};
        const double tau   = tau_fixed;
        double       pro_state[6];
        stateTransit(cur_state, pro_state, um, tau);
        const double pro_t = dynamic ? pool[cur].time + tau : 0.0;
        int          pro_id[3];
        posToIndex(pro_state, pro_id);
        const int pro_t_id = timeToIndex(pro_t);
        auto      it       = expanded.find({pro_id[0], pro_id[1], pro_id[2], pro_t_id});
        int       pro_node = it == expanded.end() ? -1 : it->second;
        if (pro_node >= 0 && pool[pro_node].node_state == IN_CLOSE_SET) continue;
        if (std::fabs(pro_state[3]) > ap->max_vel || std::fabs(pro_state[4]) > ap->max_vel ||
            std::fabs(pro_state[5]) > ap->max_vel)
          continue;
        const Node &c2 = pool[cur];
        const bool  same_vox =
            pro_id[0] == c2.index[0] && pro_id[1] == c2.index[1] && pro_id[2] == c2.index[2];
        const int diff_time = pro_t_id - c2.time_idx;
        if (same_vox && diff_time == 0) continue;
        bool is_occ = false;
        for (int k = 1; k <= ap->check_num; ++k) {
          const double dt = tau * (double)k / (double)ap->check_num;
          double       xt[6];
          stateTransit(cur_state, xt, um, dt);
          const double t = pool[cur].time + dt;
          if (query(xt, t) != 0) {
            is_occ = true;
            break;
          }
        }
        if (is_occ) continue;
        double       ttg;
        const double usq        = (um[0] * um[0] + um[1] * um[1]) + um[2] * um[2];
        const double tmp_g      = (usq + ap->w_time) * tau + pool[cur].g;
        const double tmp_f      = tmp_g + ap->lambda_heu * estimateHeuristic(pro_state, end_state, ttg);
        bool         prune      = false;
        for (size_t j = 0; j < tmp_expand_nodes.size(); ++j) {
          Node &en = pool[tmp_expand_nodes[j]];
          if (pro_id[0] == en.index[0] && pro_id[1] == en.index[1] && pro_id[2] == en.index[2] &&
              pro_t_id == en.time_idx) {
            prune = true;
            if (tmp_f < en.f) {
              en.f = tmp_f;
              en.g = tmp_g;
              for (int q = 0; q < 6; ++q) en.state[q] = pro_state[q];
              for (int q = 0; q < 3; ++q) en.input[q] = um[q];
              en.duration = tau;
              en.time     = dynamic ? pool[cur].time + tau : 0.0;
            }
            break;
          }
        }
        if (!prune) {
          if (pro_node < 0) {
            pro_node = use_node_num;
            Node &pn = pool[pro_node];
            for (int q = 0; q < 3; ++q) pn.index[q] = pro_id[q];
            for (int q = 0; q < 6; ++q) pn.state[q] = pro_state[q];
            pn.f = tmp_f;
            pn.g = tmp_g;
            for (int q = 0; q < 3; ++q) pn.input[q] = um[q];
            pn.duration   = tau;
            pn.parent     = cur;
            pn.node_state = IN_OPEN_SET;
            pn.time       = dynamic ? pool[cur].time + tau : 0.0;
            pn.time_idx   = timeToIndex(pn.time);
            open_set.push(pro_node);
            // :387  insert(pro_id, pro_node->time, pro_node): double -> int truncation
            expanded.insert({{pro_id[0], pro_id[1], pro_id[2], (int)pn.time}, pro_node});
            tmp_expand_nodes.push_back(pro_node);
            use_node_num += 1;
            if (use_node_num == ap->allocate_num) return NO_PATH;
          } else if (pool[pro_node].node_state == IN_OPEN_SET) {
            Node &pn = pool[pro_node];
            if (tmp_g < pn.g) {
              for (int q = 0; q < 6; ++q) pn.state[q] = pro_state[q];
              pn.f = tmp_f;
              pn.g = tmp_g;
              for (int q = 0; q < 3; ++q) pn.input[q] = um[q];
              pn.duration = tau;
              pn.parent   = cur;
              pn.time     = dynamic ? pool[cur].time + tau : 0.0;
            }
          } else {
            return SEARCH_ERR;
          }
        }
      }
    }
    return NO_PATH;
  }

  // :663-694
  int getPathWithVel(double delta_t, double *out, int cap) const {
    std::vector<std::array<double, 6>> list;
    int                                node = node_path.back();
    double                             t_node = 0, t_sample = delta_t;
    std::array<double, 6>              s;
    for (int i = 0; i < 6; ++i) s[i] = pool[node].state[i];
    list.push_back(s);
    while (pool[node].parent >= 0) {
      const double *ut       = pool[node].input;
      const double  duration = pool[node].duration;
      const double *x0       = pool[pool[node].parent].state;
      t_node                 = duration;
      while (true) {
        if (t_sample > t_node) {
          node = pool[node].parent;
          t_sample -= t_node;
          break;
        }
        t_node -= t_sample;
        double xt[6];
        stateTransit(x0, xt, ut, t_node);
        for (int i = 0; i < 6; ++i) s[i] = xt[i];
        list.push_back(s);
        t_sample = delta_t;
      }
    }
    std::reverse(list.begin(), list.end());
    const int n = (int)list.size();
    for (int i = 0; i < n && i < cap; ++i)
      for (int k = 0; k < 6; ++k) out[i * 6 + k] = list[i][k];
    return n;
  }
};

}  // namespace

static std::vector<Node> g_last_pool;  // diagnostics: node pool of the last orc_astar_search

extern "C" {

void orc_astar_use_libm(int on) { orc_g_use_libm = on; }

// diagnostics: rows {state[6], input[3], duration, time, g, f, index[3], time_idx, parent} = 18 doubles per node
int orc_astar_debug_nodes(double *out, int n) {
  int k = 0;
  for (; k < n && k < (int)g_last_pool.size(); ++k) {
    const Node &q = g_last_pool[k];
    double     *o = out + (size_t)k * 18;
    for (int i = 0; i < 6; ++i) o[i] = q.state[i];
    for (int i = 0; i < 3; ++i) o[6 + i] = q.input[i];
    o[9] = q.duration; o[10] = q.time; o[11] = q.g; o[12] = q.f;
    for (int i = 0; i < 3; ++i) o[13 + i] = q.index[i];
    o[16] = q.time_idx; o[17] = q.parent;
  }
  return k;
}

// 0: the replan's call pattern (baseline_fake.cpp:284-291); 1 / 2: one search(…, init = true / false, …);
// + 16: search(…, dynamic = false, …) with node times defined as zero (see Search::dynamic)
static int g_search_mode = 0;
void orc_astar_set_mode(int mode) { g_search_mode = mode; }

int orc_astar_search(const SogmSpec *s, const SogmAstarParams *ap, const float *grid,
                     const float pose[3], const double start_pva[9], const double goal[3],
                     double t_after_map, double corridor_tau, double *out_route,
                     int *out_route_len, int route_cap, int out_stats[4], int *out_trace,
                     int trace_cap, int *out_trace_len) {
  const double     zero[3] = {0, 0, 0};
  std::vector<int> trace;
  int              searches = 0;
  int              rst      = NO_PATH;
  Search           last;
  const int attempt_lo = (g_search_mode & 3) == 2 ? 1 : 0, attempt_hi = (g_search_mode & 3) == 1 ? 1 : 2;
  for (int attempt = attempt_lo; attempt < attempt_hi; ++attempt) {  // baseline_fake.cpp:284-291
    Search S;
    S.spec = s;
    S.ap   = ap;
    S.grid = grid;
    S.pose = pose;
    S.pool.assign(ap->allocate_num, Node());  // reset(): pool reused, parents/states cleared
    for (auto &n : S.pool) {
      n.parent     = -1;
      n.node_state = NOT_EXPAND;
    }
    S.inv_resolution      = 1.0 / ap->resolution;
    S.inv_time_resolution = 1.0 / ap->time_resolution;
    S.tie_breaker         = 1.0 + 1.0 / 10000;
    S.trace               = &trace;
    S.dynamic             = (g_search_mode & 16) == 0;
    rst = S.search(start_pva, start_pva + 3, start_pva + 6, goal, zero, attempt == 0, t_after_map);
    ++searches;
    last = std::move(S);
    if (rst != NO_PATH) break;
  }
  g_last_pool  = last.pool;
  out_stats[0] = last.use_node_num;
  out_stats[1] = last.iter_num;
  out_stats[2] = (int)last.node_path.size();
  out_stats[3] = searches;
  int n        = 0;
  if (rst != NO_PATH && !last.node_path.empty())
    n = last.getPathWithVel(corridor_tau, out_route, route_cap);
  *out_route_len = n;
  if (out_trace_len) *out_trace_len = (int)trace.size();
  if (out_trace)
    for (int i = 0; i < (int)trace.size() && i < trace_cap; ++i) out_trace[i] = trace[i];
  return rst;
}

}  // extern "C"
