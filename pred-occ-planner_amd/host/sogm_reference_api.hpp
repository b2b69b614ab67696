// sogm_reference_api.hpp — per-object shims with the reference's own member signatures, each a ONE-AGENT call
// into the batched C ABI (include/sogm_abi.h).  Header-only C++17, no Eigen / ROS in this file: every vector and
// matrix argument is a template parameter, so Eigen::Vector3d / Eigen::MatrixX4d / Eigen::Matrix3d are passed
// unmodified (what is used of them: v(i) / v.data(), m(i, j), m.rows(), m.resize(r, c)).
//
//   reference object (baseline.h:155-158, baseline_fake.h)          shim
//   map_            RiskBase / FakeParticleRiskVoxel                 sogm_ref::RiskMapView
//   a_star_         RiskHybridAstar / FakeRiskHybridAstar            sogm_ref::RiskHybridAstar
//   firi::firi                                                       sogm_ref::firi::firi
//   traj_optimizer_ traj_opt::BezierOpt                              sogm_ref::BezierOpt
//   collision_avoider_ ParticleATC                                   sogm_ref::ParticleATC
//   Bernstein::Bezier                                                sogm_ref::Bezier
//   FakeBaselinePlanner::{getInitCorridor, ShrinkCorridor, checkCorridorValidity, checkCorridorIntersect,
//                         checkGoalReachability}                     sogm_ref::CorridorTools
//
// An object is bound to one agent of a batched context (AgentBinding).  The per-stage ABI entries run for that
// agent only (sogm_planner_select_agents); inputs and outputs are staged through small device buffers.  This is the
// drop-in / bring-up path — tests/facade_gpu_test.cpp transcribes FakeBaselinePlanner::replan
// (plan_manager/src/baseline_fake.cpp:266-472) against it and compares with the fused sogm_replan; a service that
// wants throughput calls the batched entries (sogm_facade.hpp) directly.
//
// Host arithmetic in this file is limited to what the reference does between the calls (ShrinkCorridor's face
// offsets, stacking rows for an LP); LPs, FIRI, the search, the QP and the separation test run on the GPU.
// Compile the including translation unit without FMA contraction (-ffp-contract=off; the x86-64 baseline has no
// FMA) if ShrinkCorridor is to match the fused path bit for bit.
#pragma once

#include <chrono>
#include <cmath>
#include <functional>
#include <stdexcept>
#include <vector>

#include "sogm_facade.hpp"

namespace sogm_ref {

using sogm_host::check;
using sogm_host::DevBuf;

// path_searching/include/path_searching/dyn_a_star.h:15
enum ASTAR_RET { NO_PATH, INIT_ERR, SEARCH_ERR, REACH_HORIZON, REACH_END, NEAR_END };

// ros::Time as far as the call sites use it (getMapTime().toSec())
struct Time {
  double sec = 0.0;
  double toSec() const { return sec; }
};

// Which agent of which batched context a set of shims stands for.  `planner` may be null for map-only use.
struct AgentBinding {
  sogm_host::RiskMap *map     = nullptr;
  sogm_host::Planner *planner = nullptr;
  int                 agent   = 0;
  SogmPlannerParams   pp{};  // the parameters `planner` was created with (corridor_tau, max_faces, shrink_size ...)
};

// RAII: per-stage entries process this binding's agent only while the guard lives
class SelectAgent {
 public:
  explicit SelectAgent(const AgentBinding &b) : b_(b) {
    check(sogm_planner_select_agents(b_.planner->handle(), b_.agent, 1), "sogm_planner_select_agents");
  }
  ~SelectAgent() { (void)sogm_planner_select_agents(b_.planner->handle(), 0, b_.map->agents()); }

 private:
  const AgentBinding &b_;
};

// ---- Bernstein::Bezier (traj_utils/include/traj_utils/bernstein.hpp:107-216): order-4 pieces -------------------
class Bezier {
 public:
  Bezier() = default;
  Bezier(const std::vector<double> &durations, const std::vector<double> &cpts_rowmajor) {
    const size_t M = durations.size();
    if (M > SOGM_MAX_PIECES || cpts_rowmajor.size() != 15 * M) throw std::invalid_argument("Bezier: size");
    rec_.n_pieces = (int32_t)M;
    for (size_t i = 0; i < M; ++i) rec_.duration[i] = durations[i];
    for (size_t k = 0; k < 15 * M; ++k) rec_.cpts[k] = cpts_rowmajor[k];
  }
  explicit Bezier(const SogmTrajRecord &r) : rec_(r) {}
  int    getNumPieces() const { return rec_.n_pieces; }
  int    getOrder() const { return 4; }
  double getDuration() const {
    double T = 0.0;
    for (int i = 0; i < rec_.n_pieces; ++i) T += rec_.duration[i];
    return T;
  }
  // void getCtrlPoints(Eigen::MatrixXd&) — rows = control points
  template <class Mat>
  void getCtrlPoints(Mat &m) const {
    m.resize(5 * rec_.n_pieces, 3);
    for (int k = 0; k < 5 * rec_.n_pieces; ++k)
      for (int d = 0; d < 3; ++d) m(k, d) = rec_.cpts[k * 3 + d];
  }
  // getPos / getVel / getAcc(t): evaluated by sogm_traj_eval (t relative to the trajectory start)
  template <class V3>
  V3 getPos(double t) const { return eval<V3>(t, 0); }
  template <class V3>
  V3 getVel(double t) const { return eval<V3>(t, 1); }
  template <class V3>
  V3 getAcc(double t) const { return eval<V3>(t, 2); }
  const SogmTrajRecord &record() const { return rec_; }
  SogmTrajRecord       &record() { return rec_; }

 private:
  template <class V3>
  V3 eval(double t, int derivative) const {
    SogmTrajRecord r = rec_;
    r.time_start     = 0.0;
    DevBuf<SogmTrajRecord> d_r(1);
    DevBuf<double>         d_t(1), d_o(9);
    DevBuf<int32_t>        d_v(1);
    d_r.put(&r, 1);
    d_t.put(&t, 1);
    check(sogm_traj_eval(d_r.data(), 1, d_t.data(), d_o.data(), d_v.data(), nullptr), "sogm_traj_eval");
    double o[9];
    d_o.get(o, 9);
    V3 v;
    for (int d = 0; d < 3; ++d) v(d) = o[derivative * 3 + d];
    return v;
  }
  SogmTrajRecord rec_{};
};

// ---- map_ : RiskBase surface of one agent (plan_env/include/plan_env/risk_base.h:62-105) ----------------------
class RiskMapView {
 public:
  explicit RiskMapView(const AgentBinding &b) : b_(b) {}
  Time getMapTime() const {  // risk_base.h:76
    Time t;
    check(sogm_map_state(b_.map->ctx(), b_.agent, &t.sec, nullptr, nullptr), "sogm_map_state");
    return t;
  }
  template <class V3f>
  V3f getMapCenter() const {  // risk_base.h:70 (Eigen::Vector3f)
    float c[3];
    check(sogm_map_state(b_.map->ctx(), b_.agent, nullptr, c, nullptr), "sogm_map_state");
    V3f v;
    for (int d = 0; d < 3; ++d) v(d) = c[d];
    return v;
  }
  // int getClearOcccupancy(const Eigen::Vector3d &pos) / (pos, int t) / (pos, double dt)   risk_base.h:97-99
  template <class V3>
  int getClearOcccupancy(const V3 &pos) const { return b_.map->getClearOcccupancy(b_.agent, arr(pos), 0); }
  template <class V3>
  int getClearOcccupancy(const V3 &pos, int t) const { return b_.map->getClearOcccupancy(b_.agent, arr(pos), t); }
  template <class V3>
  int getClearOcccupancy(const V3 &pos, double dt) const { return b_.map->getClearOcccupancy(b_.agent, arr(pos), dt); }
  // void getObstaclePoints(std::vector<Eigen::Vector3d>&, double t_start, double t_end, lc, hc)   risk_base.h:101-105
  template <class V3>
  void getObstaclePoints(std::vector<V3> &points, double t_start, double t_end, const V3 &lower_corner,
                         const V3 &higher_corner, int cap = 4096) const {
    std::vector<sogm_host::Vec3> tmp;
    b_.map->getObstaclePoints(b_.agent, tmp, t_start, t_end, arr(lower_corner), arr(higher_corner), cap);
    for (const auto &q : tmp) {
      V3 v;
      for (int d = 0; d < 3; ++d) v(d) = q[d];
      points.push_back(v);
    }
  }
  // void addOtherAgents(): overlay of the swarm's trajectories (risk_base.cpp:131-226) — batched over every agent
  void addOtherAgents(const SogmTrajRecord *dev_records, int n, const int32_t *dev_ego_ids) {
    b_.map->addOtherAgents(dev_records, n, dev_ego_ids);
  }

 private:
  template <class V3>
  static sogm_host::Vec3 arr(const V3 &v) { return {v(0), v(1), v(2)}; }
  AgentBinding b_;
};

// ---- a_star_ : RiskHybridAstar (path_searching/include/path_searching/risk_hybrid_a_star.h:96-121) ------------
class RiskHybridAstar {
 public:
  explicit RiskHybridAstar(const AgentBinding &b, int route_cap = 64) : b_(b), cap_(route_cap) {
    const size_t A = (size_t)b_.map->agents();
    d_pva_.resize(A * 9); d_goal_.resize(A * 3); d_t_.resize(A); d_ret_.resize(A); d_len_.resize(A);
    d_stats_.resize(A * 4); d_route_.resize(A * (size_t)cap_ * 6);
  }
  void reset() { len_ = 0; }  // the kernel resets its node pool and hash table at the start of every search
  // ASTAR_RET search(start_pt, start_vel, start_acc, end_pt, end_vel, bool init, bool dynamic, double time_start)
  template <class V3>
  ASTAR_RET search(V3 start_pt, V3 start_vel, V3 start_acc, V3 end_pt, V3 end_vel, bool init, bool dynamic = false,
                   double time_start = -1.0) {
    // dynamic = false: the reference's branch never writes PathNode::time / time_idx / time_origin_
    // (risk_hybrid_a_star.cpp:153-162; path_node.h:44-45 leaves them uninitialised, reset() :84-101 does not touch
    // them) yet reads them for the horizon test (:177), the collision time (:300) and the hash key (:264-265) — it sees
    // whatever an earlier search left in the pool; no call site passes false (baseline.cpp:268-283,
    // baseline_fake.cpp:279-291).  DEFINED here as the oracle defines it (oracle/astar_oracle.cpp, Search::dynamic):
    // every node's time and time index are zero — the spatial search over the SOGM's first tau seconds that a freshly
    // allocated pool gives the reference; time_start is ignored, as the branch ignores it.
    for (int d = 0; d < 3; ++d)
      if (end_vel(d) != 0.0) throw std::invalid_argument("RiskHybridAstar::search: end_vel must be zero");
    const int a = b_.agent;
    double    pva[9];
    for (int d = 0; d < 3; ++d) {
      pva[d]     = start_pt(d);
      pva[3 + d] = start_vel(d);
      pva[6 + d] = start_acc(d);
    }
    double goal[3] = {end_pt(0), end_pt(1), end_pt(2)};
    (void)hipMemcpy(d_pva_.data() + a * 9, pva, sizeof(pva), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_goal_.data() + a * 3, goal, sizeof(goal), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_t_.data() + a, &time_start, sizeof(double), hipMemcpyHostToDevice);
    SelectAgent sel(b_);
    check(sogm_planner_set_search_mode(b_.planner->handle(), 4 | (init ? 1 : 2) | (dynamic ? 0 : 16)),
          "sogm_planner_set_search_mode");
    const int rc = sogm_astar_search(b_.planner->handle(), d_pva_.data(), d_goal_.data(), d_t_.data(), d_ret_.data(),
                                     d_route_.data(), d_len_.data(), cap_, d_stats_.data(), nullptr, 0, nullptr);
    (void)sogm_planner_set_search_mode(b_.planner->handle(), 0);
    check(rc, "sogm_astar_search");
    int32_t ret = 0;
    (void)hipMemcpy(&ret, d_ret_.data() + a, sizeof(ret), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&len_, d_len_.data() + a, sizeof(len_), hipMemcpyDeviceToHost);
    return (ASTAR_RET)ret;
  }
  // std::vector<Eigen::Matrix<double, 6, 1>> getPathWithVel(double delta_t): sampled inside the search kernel with
  // the planner's corridor_tau — the only delta_t the reference passes (baseline_fake.cpp:303)
  template <class V6>
  std::vector<V6> getPathWithVel(double delta_t) const {
    if (delta_t != b_.pp.corridor_tau) throw std::invalid_argument("getPathWithVel: delta_t must be corridor_tau");
    std::vector<double> h((size_t)len_ * 6);
    if (len_ > 0)
      (void)hipMemcpy(h.data(), const_cast<DevBuf<double> &>(d_route_).data() + (size_t)b_.agent * cap_ * 6,
                      h.size() * sizeof(double), hipMemcpyDeviceToHost);
    std::vector<V6> out((size_t)len_);
    for (int i = 0; i < len_; ++i)
      for (int d = 0; d < 6; ++d) out[i](d) = h[(size_t)i * 6 + d];
    return out;
  }
  template <class V3>
  std::vector<V3> getPath(double delta_t) const {  // positions of getPathWithVel
    struct V6 { double v[6]; double &operator()(int i) { return v[i]; } };
    std::vector<V3> out;
    for (auto &q : getPathWithVel<V6>(delta_t)) {
      V3 p;
      for (int d = 0; d < 3; ++d) p(d) = q.v[d];
      out.push_back(p);
    }
    return out;
  }
  // device views for callers that keep the route on the GPU
  const double  *deviceRoute() { return d_route_.data(); }
  const int32_t *deviceRouteLen() { return d_len_.data(); }
  int            routeCap() const { return cap_; }

 private:
  AgentBinding    b_;
  int             cap_;
  int32_t         len_ = 0;
  DevBuf<double>  d_pva_, d_goal_, d_t_, d_route_;
  DevBuf<int32_t> d_ret_, d_len_, d_stats_;
};

// ---- firi::firi (plan_manager/include/sfc_gen/firi.hpp:238-365) --------------------------------------------------
namespace firi {
// bool firi(const Eigen::MatrixX4d &bd, const Eigen::Matrix3Xd &pc, const Vector3d &a, const Vector3d &b,
//           Eigen::MatrixX4d &hPoly, Eigen::Vector3d &r, const int iterations = 4, const double epsilon = 1e-6)
// BD / HP: (i, j), rows(), resize(rows, 4).  PC: (i, j) with i in 0..2 and cols() (a 3 x N matrix or Map).
template <class BD, class PC, class V3, class HP>
inline bool firi(const BD &bd, const PC &pc, const V3 &a, const V3 &b, HP &hPoly, V3 &r, const int iterations = 4,
                 const double epsilon = 1.0e-6) {
  const int M = (int)bd.rows(), N = (int)pc.cols();
  const int max_faces = 128;
  if (M < 1 || M > 32) throw std::invalid_argument("firi: bd must have 1..32 rows");
  if (N > 16384) throw std::invalid_argument("firi: more than 16384 points");
  std::vector<double> hbd((size_t)M * 4), hpc((size_t)(N > 0 ? N : 1) * 3);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < 4; ++j) hbd[(size_t)i * 4 + j] = bd(i, j);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < 3; ++j) hpc[(size_t)i * 3 + j] = pc(j, i);
  const double  ha[3] = {a(0), a(1), a(2)}, hb[3] = {b(0), b(1), b(2)};
  double        hr[3] = {r(0), r(1), r(2)};
  const int32_t range[2] = {0, N};
  DevBuf<double>  d_bd, d_pc, d_a, d_b, d_r, d_hp((size_t)max_faces * 4);
  DevBuf<int32_t> d_range, d_nf(1), d_st(1);
  d_bd.put(hbd.data(), hbd.size()); d_pc.put(hpc.data(), hpc.size()); d_a.put(ha, 3); d_b.put(hb, 3); d_r.put(hr, 3);
  d_range.put(range, 2);
  check(sogm_firi_batched(d_bd.data(), M, d_pc.data(), d_range.data(), d_a.data(), d_b.data(), d_r.data(), iterations,
                          epsilon, 1, N > 0 ? N : 1, max_faces, d_hp.data(), d_nf.data(), d_st.data(), nullptr),
        "sogm_firi_batched");
  int32_t nf = 0, st = 0;
  d_nf.get(&nf, 1); d_st.get(&st, 1);
  if (st == 0) return false;  // a or b outside bd
  if (st < 0) throw std::runtime_error("firi: more than 128 planes selected");
  std::vector<double> hp((size_t)nf * 4);
  if (nf > 0) d_hp.get(hp.data(), hp.size());
  d_r.get(hr, 3);
  hPoly.resize(nf, 4);
  for (int i = 0; i < nf; ++i)
    for (int j = 0; j < 4; ++j) hPoly(i, j) = hp[(size_t)i * 4 + j];
  for (int d = 0; d < 3; ++d) r(d) = hr[d];
  return true;
}
}  // namespace firi

// ---- FakeBaselinePlanner's corridor helpers (plan_manager/src/baseline_fake.cpp:121-211) ------------------------
struct CorridorTools {
  double shrink_size  = 0.2;   // cfg_.shrink_size
  bool   fake_planner = true;  // FakeBaselinePlanner's rules; false: BaselinePlanner's (baseline.cpp:206-228)
  // Eigen::Matrix<double, 6, 4> getInitCorridor(left_higher_corner, right_lower_corner)   :121-136
  template <class M64, class V3>
  static M64 getInitCorridor(const V3 &left_higher_corner, const V3 &right_lower_corner) {
    M64 c;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 4; ++j) c(i, j) = 0.0;
    for (int k = 0; k < 3; ++k) {
      c(k, k)     = 1.0;
      c(k + 3, k) = -1.0;
      c(k, 3)     = -left_higher_corner(k);
      c(k + 3, 3) = right_lower_corner(k);
    }
    return c;
  }
  // sdlp::linprog<3>(c, A, b, x) with A | -b = the rows of `corridor` (one LP on the GPU)
  template <class HP>
  static double linprog3(const double c[3], const HP &corridor, const HP *second, double x[3]) {
    const int m1 = (int)corridor.rows(), m2 = second ? (int)second->rows() : 0, m = m1 + m2;
    std::vector<double> A((size_t)(m > 0 ? m : 1) * 3), bb((size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m; ++i) {
      const HP &src = i < m1 ? corridor : *second;
      const int  k  = i < m1 ? i : i - m1;
      for (int j = 0; j < 3; ++j) A[(size_t)i * 3 + j] = src(k, j);
      bb[i] = -src(k, 3);
    }
    const int32_t   range[2] = {0, m};
    DevBuf<double>  d_c, d_A, d_b, d_x(3), d_v(1);
    DevBuf<int32_t> d_range;
    d_c.put(c, 3); d_A.put(A.data(), A.size()); d_b.put(bb.data(), bb.size()); d_range.put(range, 2);
    check(sogm_linprog_batched(3, d_c.data(), d_A.data(), d_b.data(), d_range.data(), 1, d_x.data(), d_v.data(),
                               nullptr), "sogm_linprog_batched");
    double v = 0.0;
    d_v.get(&v, 1);
    d_x.get(x, 3);
    return v;
  }
  // bool checkCorridorValidity(const Eigen::MatrixX4d &corridor)   :191-204
  template <class HP>
  static bool checkCorridorValidity(const HP &corridor) {
    const double c[3] = {0, 0, 0};
    double       x[3];
    return !std::isinf(linprog3(c, corridor, (const HP *)nullptr, x));
  }
  // bool checkCorridorIntersect(corridor1, corridor2)   :184-189
  template <class HP>
  static bool checkCorridorIntersect(const HP &corridor1, const HP &corridor2) {
    const double c[3] = {0, 0, 0};
    double       x[3];
    return !std::isinf(linprog3(c, corridor1, &corridor2, x));
  }
  // bool checkGoalReachability(corridor, start_pos, goal_pos&)   :138-182
  template <class HP, class V3>
  static bool checkGoalReachability(const HP &corridor, const V3 &start_pos, V3 &goal_pos) {
    const int m = (int)corridor.rows();
    if (m <= 0) return true;
    double mx = -INFINITY;
    for (int i = 0; i < m; ++i) {
      const double v = ((corridor(i, 0) * goal_pos(0) + corridor(i, 1) * goal_pos(1)) + corridor(i, 2) * goal_pos(2)) +
                       corridor(i, 3) * 1.0;
      mx = v > mx ? v : mx;
    }
    if (mx <= 0) return true;
    double c[3], gmax[3], gmin[3];
    for (int j = 0; j < 3; ++j) c[j] = -goal_pos(j) + start_pos(j);
    linprog3(c, corridor, (const HP *)nullptr, gmax);
    for (int j = 0; j < 3; ++j) c[j] = goal_pos(j) - start_pos(j);
    linprog3(c, corridor, (const HP *)nullptr, gmin);
    for (int j = 0; j < 3; ++j) goal_pos(j) = 0.5 * (gmax[j] + gmin[j]);
    return false;
  }
  // void ShrinkCorridor(Eigen::MatrixX4d &corridor, const Eigen::Vector3d &path)   :217-230
  template <class HP, class V3>
  void ShrinkCorridor(HP &corridor, const V3 &path) const {
    for (int i = 0; i < (int)corridor.rows(); ++i) {
      const double A = corridor(i, 0), B = corridor(i, 1), C = corridor(i, 2);
      const double nrm = std::sqrt((A * A + B * B) + C * C);
      if (fake_planner) {  // BaselinePlanner has both tests commented out (baseline.cpp:224-225)
        const double pn = std::sqrt((path(0) * path(0) + path(1) * path(1)) + path(2) * path(2));
        if (((A * path(0) + B * path(1)) + C * path(2)) / nrm / pn > 0.8) continue;  // not shrink front and back
        if (std::abs(C) / nrm > 0.8) continue;                                       // not shrink top and bottom
      }
      corridor(i, 3) += nrm * shrink_size;
    }
  }
  // void ShrinkCorridor(Eigen::MatrixX4d &corridor)   :206-215 (not called by either replan)
  template <class HP>
  void ShrinkCorridor(HP &corridor) const {
    for (int i = 0; i < (int)corridor.rows(); ++i) {
      const double A = corridor(i, 0), B = corridor(i, 1), C = corridor(i, 2);
      if (fake_planner && std::abs(C) > std::sqrt(A * A + B * B)) continue;  // not shrink top and bottom
      corridor(i, 3) += std::sqrt((A * A + B * B) + C * C) * shrink_size;
    }
  }
};

// ---- traj_optimizer_ : traj_opt::BezierOpt (traj_opt/include/bernstein/bezier_optimizer.hpp:57-88) -------------
class BezierOpt {
 public:
  explicit BezierOpt(const AgentBinding &b) : b_(b) {
    const size_t A = (size_t)b_.map->agents(), MF = (size_t)b_.pp.max_faces;
    d_pva_.resize(A * 9); d_goal_.resize(A * 9); d_tal_.resize(A * SOGM_MAX_PIECES);
    d_polys_.resize(A * SOGM_MAX_PIECES * MF * 4);
    d_nf_.resize(A * SOGM_MAX_PIECES); d_np_.resize(A); d_cpts_.resize(A * SOGM_MAX_PIECES * 15); d_st_.resize(A);
    d_it_.resize(A);
  }
  // void setup(const Matrix3d &start, const Matrix3d &end, time_allocation, constraints, max_vel, max_acc)
  // start / end rows = position, velocity, acceleration; any time allocation per piece, any end state, the
  // caller's limits (bezier_optimizer.cpp:27-54,113-260) — sogm_bezier_qp_solve_timed.  (replan()'s call sites,
  // baseline_fake.cpp:429-441 / baseline.cpp:420-432, pass corridor_tau for every piece and a zero final
  // acceleration: the special case the fused sogm_replan assembles.)
  template <class M3, class HP>
  void setup(const M3 &start, const M3 &end, const std::vector<double> &time_allocation,
             const std::vector<HP> &constraints, const double &max_vel = 3.0, const double &max_acc = 3.0) {
    const int M = (int)constraints.size(), MF = b_.pp.max_faces;
    if (M < 1 || M > SOGM_MAX_PIECES || (int)time_allocation.size() != M)
      throw std::invalid_argument("BezierOpt::setup: piece count");
    for (double t : time_allocation)
      if (!(t > 0.0)) throw std::invalid_argument("BezierOpt::setup: time allocation must be positive");
    if (!(max_vel > 0.0) || !(max_acc > 0.0)) throw std::invalid_argument("BezierOpt::setup: limits must be positive");
    vmax_ = max_vel;
    amax_ = max_acc;
    t_    = time_allocation;
    double pva[9], goal[9], tal[SOGM_MAX_PIECES] = {0};
    for (int r = 0; r < 3; ++r)
      for (int d = 0; d < 3; ++d) {
        pva[r * 3 + d]  = start(r, d);
        goal[r * 3 + d] = end(r, d);
      }
    for (int i = 0; i < M; ++i) tal[i] = time_allocation[i];
    std::vector<double>  polys((size_t)SOGM_MAX_PIECES * MF * 4, 0.0);
    std::vector<int32_t> nf(SOGM_MAX_PIECES, 0);
    for (int i = 0; i < M; ++i) {
      const int rows = (int)constraints[i].rows();
      if (rows > MF) throw std::invalid_argument("BezierOpt::setup: polytope has more than max_faces rows");
      nf[i] = rows;
      for (int k = 0; k < rows; ++k)
        for (int j = 0; j < 4; ++j) polys[((size_t)i * MF + k) * 4 + j] = constraints[i](k, j);
    }
    const int     a  = b_.agent;
    const int32_t np = M;
    (void)hipMemcpy(d_pva_.data() + a * 9, pva, sizeof(pva), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_goal_.data() + a * 9, goal, sizeof(goal), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_tal_.data() + (size_t)a * SOGM_MAX_PIECES, tal, sizeof(tal), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_polys_.data() + (size_t)a * SOGM_MAX_PIECES * MF * 4, polys.data(), polys.size() * sizeof(double),
                    hipMemcpyHostToDevice);
    (void)hipMemcpy(d_nf_.data() + a * SOGM_MAX_PIECES, nf.data(), nf.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_np_.data() + a, &np, sizeof(np), hipMemcpyHostToDevice);
    M_ = M;
  }
  bool optimize() {  // OSQP status solved (1) or solved-inaccurate (2) -> true (bezier_optimizer.cpp:318-337)
    if (M_ <= 0) throw std::logic_error("BezierOpt::optimize before setup");
    SelectAgent sel(b_);
    check(sogm_bezier_qp_solve_timed(b_.planner->handle(), d_pva_.data(), d_goal_.data(), d_tal_.data(), vmax_, amax_,
                                     d_polys_.data(), d_nf_.data(), d_np_.data(), d_cpts_.data(), d_st_.data(),
                                     d_it_.data(), nullptr),
          "sogm_bezier_qp_solve_timed");
    (void)hipMemcpy(&status_, d_st_.data() + b_.agent, sizeof(status_), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&iters_, d_it_.data() + b_.agent, sizeof(iters_), hipMemcpyDeviceToHost);
    return status_ == 1 || status_ == 2;
  }
  void getOptBezier(Bezier &bc) const {
    std::vector<double> c((size_t)M_ * 15), t(t_);
    (void)hipMemcpy(c.data(), const_cast<DevBuf<double> &>(d_cpts_).data() + (size_t)b_.agent * SOGM_MAX_PIECES * 15,
                    c.size() * sizeof(double), hipMemcpyDeviceToHost);
    bc = Bezier(t, c);
  }
  Bezier getOptBezier() const {
    Bezier bc;
    getOptBezier(bc);
    return bc;
  }
  template <class Mat>
  void getOptCtrlPtsMat(Mat &m) const { getOptBezier().getCtrlPoints(m); }
  int status() const { return status_; }      // OSQP status code
  int iterations() const { return iters_; }   // ADMM iterations
  // device views (control points of every agent, sogm_bezier_qp_solve layout)
  const double  *deviceCtrlPts() { return d_cpts_.data(); }
  const int32_t *deviceNumPieces() { return d_np_.data(); }

 private:
  AgentBinding    b_;
  int             M_ = 0;
  int32_t         status_ = 0, iters_ = 0;
  double          vmax_ = 3.0, amax_ = 3.0;
  std::vector<double> t_;
  DevBuf<double>  d_pva_, d_goal_, d_tal_, d_polys_, d_cpts_;
  DevBuf<int32_t> d_nf_, d_np_, d_st_, d_it_;
};

// ---- collision_avoider_ : ParticleATC (traj_coordinator/include/traj_coordinator/particle.hpp:105-150) ----------
class ParticleATC {
 public:
  // `clock` stands in for ros::Time::now().toSec() (particles.cpp:147,245)
  ParticleATC(const AgentBinding &b, int drone_id, int max_agents,
              std::function<double()> clock = [] {
                return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
              })
      : b_(b), drone_id_(drone_id), clock_(std::move(clock)), d_swarm_((size_t)max_agents), cap_(max_agents) {}
  int getNumAgents() const { return cap_; }
  // void trajectoryCallback(const traj_utils::BezierTraj::ConstPtr&): latest trajectory per drone_id, ego filtered
  // out (particles.cpp:131-191).  Returns false for a message that does not fit a record (not order 4, too long).
  bool trajectoryCallback(const sogm_host::BezierTrajMsg &traj_msg) {
    if (traj_msg.drone_id == drone_id_) return true;
    SogmTrajRecord r;
    if (!sogm_host::recordFromMsg(traj_msg, r)) return false;
    size_t slot = ids_.size();
    for (size_t i = 0; i < ids_.size(); ++i)
      if (ids_[i] == traj_msg.drone_id) slot = i;
    if (slot == ids_.size()) {
      if ((int)slot >= cap_) return false;
      ids_.push_back(traj_msg.drone_id);
    }
    (void)hipMemcpy(d_swarm_.data() + slot, &r, sizeof(r), hipMemcpyHostToDevice);
    return true;
  }
  // bool isSafeAfterOpt(const Bernstein::Bezier &traj)   particles.cpp:223-283
  bool isSafeAfterOpt(const Bezier &traj) {
    const size_t A = (size_t)b_.map->agents();
    const int    a = b_.agent, M = traj.getNumPieces();
    d_cpts_.resize(A * SOGM_MAX_PIECES * 15); d_np_.resize(A); d_ego_.resize(A); d_now_.resize(A); d_safe_.resize(A);
    const int32_t np = M, ego = drone_id_;
    const double  now = clock_();
    (void)hipMemcpy(d_cpts_.data() + (size_t)a * SOGM_MAX_PIECES * 15, traj.record().cpts, sizeof(double) * 15 * M,
                    hipMemcpyHostToDevice);
    (void)hipMemcpy(d_np_.data() + a, &np, sizeof(np), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_ego_.data() + a, &ego, sizeof(ego), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_now_.data() + a, &now, sizeof(now), hipMemcpyHostToDevice);
    SelectAgent sel(b_);
    check(sogm_safe_after_opt(b_.planner->handle(), d_cpts_.data(), d_np_.data(), d_swarm_.data(), (int)ids_.size(),
                              d_ego_.data(), d_now_.data(), d_safe_.data(), nullptr), "sogm_safe_after_opt");
    int32_t safe = 0;
    (void)hipMemcpy(&safe, d_safe_.data() + a, sizeof(safe), hipMemcpyDeviceToHost);
    return safe != 0;
  }
  // the stored swarm table (device) for RiskMapView::addOtherAgents
  const SogmTrajRecord *deviceSwarm() { return d_swarm_.data(); }
  int                   swarmSize() const { return (int)ids_.size(); }

 private:
  AgentBinding            b_;
  int                     drone_id_;
  std::function<double()> clock_;
  DevBuf<SogmTrajRecord>  d_swarm_;
  int                     cap_;
  std::vector<int32_t>    ids_;
  DevBuf<double>          d_cpts_, d_now_;
  DevBuf<int32_t>         d_np_, d_ego_, d_safe_;
};

}  // namespace sogm_ref
