// sogm_facade.hpp — host C++ facade that keeps the reference's class surface and forwards to the
// C ABI (include/sogm_abi.h).  Header-only, C++17, no Eigen / ROS: vectors are std::array<double,3>
// (an Eigen::Vector3d's storage is layout-compatible: pass v.data()).
//
// One facade object is a VIEW of one agent inside a batched context, so that a plan_manager-style
// caller written against
//     map_->getClearOcccupancy(pos, t) / getObstaclePoints(...)      (risk_base.h:84-105)
//     a_star_->search(...) / getPathWithVel(dt)                      (risk_hybrid_a_star.h:96-121)
//     firi::firi(...)                                                (sfc_gen/firi.hpp:238)
//     traj_optimizer_->setup(...) / optimize() / getOptBezier(...)   (bezier_optimizer.hpp:57-88)
// links against this instead (plan_manager/include/plan_manager/baseline.h:155-158 holds exactly
// these four pointers).  The batched entry points are what a multi-agent service calls directly;
// the per-agent methods below stage one agent's arguments through small device buffers and are
// meant for drop-in use and testing, not for throughput.
#pragma once

#include <hip/hip_runtime_api.h>

#include <array>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sogm_abi.h"

namespace sogm_host {

using Vec3 = std::array<double, 3>;
using State6 = std::array<double, 6>;

inline void check(int rc, const char *what) {
  if (rc != SOGM_OK)
    throw std::runtime_error(std::string(what) + " failed: " + std::to_string(rc) + " " + sogm_last_error());
}

// The library must implement the header this host was compiled against: buffer sizes of existing entry points have
// changed between versions (sogm_abi.h, SOGM_ABI_VERSION).  Every facade object checks once at construction.
inline void check_abi() {
  static const int got = sogm_abi_version();
  if (got != SOGM_ABI_VERSION)
    throw std::runtime_error("libsogm_hip.so implements ABI version " + std::to_string(got) + ", this host was built against " +
                             std::to_string(SOGM_ABI_VERSION));
}

template <typename T>
class DevBuf {  // tiny RAII device buffer
 public:
  explicit DevBuf(size_t n = 0) { resize(n); }
  ~DevBuf() { if (p_) (void)hipFree(p_); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  void resize(size_t n) {
    if (n <= n_) return;
    if (p_) (void)hipFree(p_);
    if (hipMalloc((void **)&p_, n * sizeof(T)) != hipSuccess) throw std::bad_alloc();
    n_ = n;
  }
  void put(const T *h, size_t n) { resize(n); (void)hipMemcpy(p_, h, n * sizeof(T), hipMemcpyHostToDevice); }
  void get(T *h, size_t n) const { (void)hipMemcpy(h, p_, n * sizeof(T), hipMemcpyDeviceToHost); }
  T *data() { return p_; }

 private:
  T     *p_ = nullptr;
  size_t n_ = 0;
};

// ---- wire formats -----------------------------------------------------------------------
// traj_utils/msg/BezierTraj.msg:1-9 as filled by FiniteStateMachine::publishTrajectory
// (plan_manager/src/plan_manager.cpp:364-399) and read by ParticleATC::trajectoryCallback
// (traj_coordinator/src/particles.cpp:131-191): float32 durations, geometry_msgs/Point control points.
struct BezierTrajMsg {
  int32_t             drone_id = 0, traj_id = 0;
  double              start_time = 0.0, pub_time = 0.0;  // ros::Time::toSec()
  uint8_t             order = 4;
  std::vector<float>  duration;
  std::vector<Vec3>   cpts;
};
// message -> fixed-size record (the all-gather payload).  Returns false when the message does not fit
// (more than SOGM_MAX_PIECES pieces, order != 4 or an inconsistent control-point count).
inline bool recordFromMsg(const BezierTrajMsg &m, SogmTrajRecord &r) {
  const size_t n = m.duration.size();
  if (m.order != 4 || n > SOGM_MAX_PIECES || m.cpts.size() != 5 * n) return false;
  r = SogmTrajRecord{};
  r.drone_id   = m.drone_id;
  r.n_pieces   = (int32_t)n;
  r.time_start = m.start_time;
  for (size_t i = 0; i < n; ++i) r.duration[i] = (double)m.duration[i];  // float32 on the wire
  for (size_t k = 0; k < 5 * n; ++k)
    for (int d = 0; d < 3; ++d) r.cpts[k * 3 + d] = m.cpts[k][d];
  return true;
}
inline BezierTrajMsg msgFromRecord(const SogmTrajRecord &r, int32_t traj_id, double pub_time) {
  BezierTrajMsg m;
  m.drone_id   = r.drone_id;
  m.traj_id    = traj_id;
  m.start_time = r.time_start;
  m.pub_time   = pub_time;
  m.order      = 4;
  for (int i = 0; i < r.n_pieces; ++i) m.duration.push_back((float)r.duration[i]);
  for (int k = 0; k < 5 * r.n_pieces; ++k) m.cpts.push_back({r.cpts[k * 3], r.cpts[k * 3 + 1], r.cpts[k * 3 + 2]});
  return m;
}
// map/future_risk std_msgs/Float32MultiArray (plan_env/src/risk_mapping_node.cpp:129-145, consumer
// risk_base.cpp:60-75): data = [V * stride risk values | pose x, y, z | stamp], layout.dim[0].stride = T.
// Splits one message into the arguments of sogm_set_future_risk (host side; grid returned voxel-major [V][T]).
inline bool splitFutureRiskMsg(const std::vector<float> &data, int V, int T, int stride, std::vector<float> &grid_vt,
                               float pose[3], double &stamp) {
  if (stride < T || data.size() < (size_t)V * stride + 4) return false;
  grid_vt.resize((size_t)V * T);
  for (int i = 0; i < V; ++i)
    for (int j = 0; j < T; ++j) grid_vt[(size_t)i * T + j] = data[(size_t)i * stride + j];
  // pose and time follow the V * PREDICTION_TIMES block (risk_base.cpp:72 reads index V*T + 3)
  const size_t o = (size_t)V * T;
  pose[0] = data[o];
  pose[1] = data[o + 1];
  pose[2] = data[o + 2];
  stamp   = (double)data[o + 3];  // float32 seconds on the wire
  return true;
}
inline std::vector<float> futureRiskMsg(const std::vector<float> &grid_vt, const float pose[3], double stamp) {
  std::vector<float> d(grid_vt);
  d.push_back(pose[0]);
  d.push_back(pose[1]);
  d.push_back(pose[2]);
  d.push_back((float)stamp);
  return d;
}

// ---- RiskMap / SOGM : RiskBase + FakeParticleRiskVoxel surface --------------------------------------
class RiskMap {
 public:
  RiskMap(const SogmSpec &spec, int n_agents, int device = 0) : n_(n_agents), spec_(spec) {
    check_abi();
    check(sogm_create(&spec, n_agents, device, &ctx_), "sogm_create");
  }
  ~RiskMap() { sogm_destroy(ctx_); }
  sogm_ctx *ctx() { return ctx_; }
  int       agents() const { return n_; }

  // ParticleATC::initEgoParticles + setCoordinator (risk_base.h:62-65)
  // (pass the particles as ANOTHER drone's ParticleATC holds them: particlesCallback reads Point32 coordinates —
  //  (double)(float)x — particles.cpp:89-108; the library uses what it is given)
  void setCoordinator(const std::vector<Vec3> &body_particles) {
    check(sogm_set_body_particles(ctx_, body_particles[0].data(), (int)body_particles.size()), "set_body");
  }
  // Tick pipelining (no reference counterpart): 0 off, 1 in-place reset after the replan's last map reader, 2 / 3 a pool of
  // two / three grids (returns false and leaves the mode unchanged if HBM has no room for them)
  bool setTickPipelining(int mode) {
    const int rc = sogm_set_overlap_clear(ctx_, mode);
    if (rc == SOGM_ERR_CAPACITY) return false;
    check(rc, "sogm_set_overlap_clear");
    return true;
  }
  // How updateMap's "fill the map with zeros" (fake_particle_risk_voxel.cpp:107-108) is carried out: by zeroing the
  // logged sectors of the previous build's marks (default) or by the dense clear; same cells either way
  // swarm/replan_risk_rate, swarm/num_resample (particles.cpp:33-34) and the injected standard-normal table that
  // stands in for getParticlesWithRisk's time(NULL)-seeded generator (sogm_set_resample); rate 0 = off (the default)
  void setResample(float replan_risk_rate, int num_resample, const float *normal_table_dev, int n_table) {
    check(sogm_set_resample(ctx_, replan_risk_rate, num_resample, normal_table_dev, n_table), "sogm_set_resample");
  }
#ifdef SOGM_ABI_DEBUG_H
  // a tuning knob of this context (include/sogm_abi_debug.h "Tuning knobs"; there when that header was included first)
  void setTuning(const char *key, double value) { check(sogm_set_tuning(ctx_, key, value), "sogm_set_tuning"); }
#endif
  void setSparseReset(bool on, int log_capacity_per_agent = 0) {
    check(sogm_set_sparse_reset(ctx_, on ? 1 : 0, log_capacity_per_agent), "sogm_set_sparse_reset");
  }
  // SOGM::update — FakeParticleRiskVoxel::updateMap for the whole batch (device pointers)
  void update(const float *cloud_xyz, const int32_t *cloud_range, const SogmCylinder *cyl, int n_cyl,
              const float *poses, const double *stamps, hipStream_t st = nullptr) {
    check(sogm_update_gt(ctx_, cloud_xyz, cloud_range, cyl, n_cyl, poses, stamps, st), "sogm_update_gt");
  }
  // FakeParticleRiskVoxel::updateMap including its closing neighbour overlay (fake_particle_risk_voxel.cpp:175-226)
  void updateWithSwarm(const float *cloud_xyz, const int32_t *cloud_range, const SogmCylinder *cyl, int n_cyl,
                       const float *poses, const double *stamps, const SogmTrajRecord *records, int n_records,
                       const int32_t *ego_ids, hipStream_t st = nullptr) {
    check(sogm_update_gt_swarm(ctx_, cloud_xyz, cloud_range, cyl, n_cyl, poses, stamps, records, n_records, ego_ids, st),
          "sogm_update_gt_swarm");
  }
  // updateMap from one sensor frame (MapBase::cloudCallback -> updateMap with the states groundTruthStateCallback
  // delivered, map.cpp:170-171, fake_particle_risk_voxel.cpp:244-264): the PassThrough crop (:88-104) runs on the device
  // around each agent's map centre.  `bounds`: xy bounds of the cloud's blocks of `block_points` points — blockBounds()
  // computes them on the device, once per frame.  records = nullptr: no overlay.
  static void blockBounds(const float *cloud_xyz, int n_points, int block_points, float *bounds, hipStream_t st = nullptr) {
    check(sogm_cloud_block_bounds(cloud_xyz, n_points, block_points, bounds, st), "sogm_cloud_block_bounds");
  }
  static SogmWorld world(const float *cloud_xyz, int n_points, const float *bounds, int block_points,
                         const SogmCylinder *cyl, int n_cyl) {
    return SogmWorld{cloud_xyz, bounds, cyl, n_points, (n_points + block_points - 1) / block_points, block_points, n_cyl};
  }
  void updateWorld(const SogmWorld &frame, const float *poses, const double *stamps, const SogmTrajRecord *records = nullptr,
                   int n_records = 0, const int32_t *ego_ids = nullptr, hipStream_t st = nullptr) {
    check(sogm_update_world(ctx_, &frame, poses, stamps, records, n_records, ego_ids, st), "sogm_update_world");
  }
  // The update of a tick whose map the previous replan pre-stamped (Planner::setPrestamp): grid swap + overlay
  bool prestampPending() const { return sogm_prestamp_pending(ctx_) != 0; }
  void prestampJoin(void *stream = nullptr) { check(sogm_prestamp_join(ctx_, stream), "sogm_prestamp_join"); }
  void updatePrestamped(const SogmTrajRecord *records, int n_records, const int32_t *ego_ids, hipStream_t st = nullptr) {
    check(sogm_update_prestamped(ctx_, records, n_records, ego_ids, st), "sogm_update_prestamped");
  }
  // RiskBase::futureRiskCallback for agent 0 of a 1-agent context: adopt one map/future_risk message.
  // `map_time` is the map stamp to adopt (the ROS header / receive time as a double).  The reference reads the
  // message's trailing float32 field into a dead local (risk_base.cpp:72) and leaves last_update_time_ alone; a
  // float32 epoch time is quantised to ~128 s, so it is only used when no map_time is given (NaN).
  void futureRiskCallback(const std::vector<float> &msg_data, int stride, hipStream_t st = nullptr,
                          double map_time = std::numeric_limits<double>::quiet_NaN()) {
    const SogmSpec sp = spec_;
    const int      V  = sp.L * sp.W * sp.H;
    std::vector<float> grid;
    float  pose[3];
    double stamp = 0.0;
    if (n_ != 1 || !splitFutureRiskMsg(msg_data, V, sp.T, stride, grid, pose, stamp))
      throw std::runtime_error("futureRiskCallback: malformed message");
    if (map_time == map_time) stamp = map_time;
    g_.put(grid.data(), grid.size());
    f_.put(pose, 3);
    t_.put(&stamp, 1);
    check(sogm_set_future_risk(ctx_, g_.data(), f_.data(), t_.data(), st), "sogm_set_future_risk");
  }
  // RiskBase::addOtherAgents
  void addOtherAgents(const SogmTrajRecord *records, int n, const int32_t *ego_ids, hipStream_t st = nullptr) {
    check(sogm_project_neighbours(ctx_, records, n, ego_ids, st), "sogm_project_neighbours");
  }
  // int getClearOcccupancy(const Vector3d& pos, double dt) const  -> 0 free / 1 occupied / -1 out
  int getClearOcccupancy(int agent, const Vec3 &pos, double dt) {
    int32_t a = agent; int8_t r = 0;
    a_.put(&a, 1); p_.put(pos.data(), 3); t_.put(&dt, 1); o_.resize(1);
    check(sogm_query_clear(ctx_, a_.data(), p_.data(), t_.data(), 0, 1, o_.data(), nullptr), "sogm_query_clear");
    o_.get(&r, 1);
    return r;
  }
  int getClearOcccupancy(int agent, const Vec3 &pos, int t_index) {
    int32_t a = agent; int8_t r = 0; double t = t_index;
    a_.put(&a, 1); p_.put(pos.data(), 3); t_.put(&t, 1); o_.resize(1);
    check(sogm_query_clear(ctx_, a_.data(), p_.data(), t_.data(), 1, 1, o_.data(), nullptr), "sogm_query_clear");
    o_.get(&r, 1);
    return r;
  }
  // void getObstaclePoints(std::vector<Vector3d>&, double t0, double t1, const Vector3d& lc, const Vector3d& hc)
  void getObstaclePoints(int agent, std::vector<Vec3> &points, double t0, double t1, const Vec3 &lc,
                         const Vec3 &hc, int cap = 4096) {
    int32_t a = agent, n = 0;
    a_.put(&a, 1); p_.put(lc.data(), 3); q_.put(hc.data(), 3); t_.put(&t0, 1); u_.put(&t1, 1);
    pts_.resize((size_t)cap * 3); cnt_.resize(1);
    check(sogm_obstacle_points(ctx_, a_.data(), p_.data(), q_.data(), t_.data(), u_.data(), 1, pts_.data(),
                               cnt_.data(), cap, nullptr), "sogm_obstacle_points");
    cnt_.get(&n, 1);
    if (n > cap) n = cap;
    std::vector<double> h((size_t)n * 3);
    pts_.get(h.data(), h.size());
    for (int i = 0; i < n; ++i) points.push_back({h[i * 3], h[i * 3 + 1], h[i * 3 + 2]});
  }

 private:
  sogm_ctx       *ctx_ = nullptr;
  int             n_;
  SogmSpec        spec_;
  DevBuf<float>   g_, f_;
  DevBuf<int32_t> a_, cnt_;
  DevBuf<double>  p_, q_, t_, u_, pts_;
  DevBuf<int8_t>  o_;
};

// ---- DspMap : dsp_map::DSPMap as owned by RiskVoxel (risk_voxel.cpp:42-50,138-153,237-254) ----------
// One instance serves every agent of a RiskMap created with map_kind = SOGM_MAP_RISKVOXEL.
class DspMap {
 public:
  // DSPMap() + setPredictionVariance / setObservationStdDev / setNewBornParticle* (risk_voxel.cpp:42-49);
  // the Gaussian and rand() tables the reference seeds from time(0) are supplied by the caller.
  DspMap(RiskMap &map, const SogmDspParams &p, const std::vector<float> &p_gauss,
         const std::vector<float> &v_gauss, const std::vector<int32_t> &rand_tab, int max_points = 5000) {
    check(sogm_dsp_create(map.ctx(), &p, p_gauss.data(), v_gauss.data(), (int)p_gauss.size(), rand_tab.data(),
                          (int)rand_tab.size(), max_points, &h_), "sogm_dsp_create");
  }
  ~DspMap() { sogm_dsp_destroy(h_); }
  // MapBase::filterPointCloud (map.cpp:107-132) — device pointers, camera-frame cloud in, body-frame out
  static void filterPointCloud(RiskMap &map, const float *raw_xyz, const int32_t *raw_range, float filter_res,
                               float *out_xyz, int32_t *out_count, int cap = 5000, hipStream_t st = nullptr) {
    check(sogm_filter_point_cloud(map.ctx(), raw_xyz, raw_range, filter_res, cap, out_xyz, out_count, st),
          "sogm_filter_point_cloud");
  }
  // int DSPMap::update(n, 3, pts, px, py, pz, stamp, qw, qx, qy, qz) for the whole batch (device pointers)
  // labels == nullptr: velocityEstimationThread (clustering + association, :1487-1678) runs on the GPU
  void update(const float *points, const float *labels, const int32_t *cloud_range, const float *sensor_pos,
              const float *sensor_quat, const double *stamps, int32_t *out_ok, hipStream_t st = nullptr) {
    check(sogm_update_dsp(h_, points, labels, cloud_range, sensor_pos, sensor_quat, stamps, out_ok, st),
          "sogm_update_dsp");
  }
  // RiskVoxel::publishMap: getOccupancyMapWithFutureStatus -> risk_maps_ (+ RiskMap::addOtherAgents after)
  void publishMap(int32_t *out_n_occupied = nullptr, hipStream_t st = nullptr) {
    check(sogm_dsp_publish(h_, out_n_occupied, st), "sogm_dsp_publish");
  }

 private:
  sogm_dsp *h_ = nullptr;
};

// ---- GridMap : depth-image occupancy front end (plan_env/src/grid_map.cpp) ------------------------------
class GridMap {
 public:
  GridMap(const SogmGridMapParams &p, int n_agents, int device = 0) {  // GridMap::initMap
    check_abi();
    check(sogm_gridmap_create(&p, n_agents, device, &h_), "sogm_gridmap_create");
  }
  ~GridMap() { sogm_gridmap_destroy(h_); }
  // depthPoseCallback + updateOccupancyCallback for the whole batch (device pointers)
  void update(const uint16_t *depth, const double *cam_pos, const double *cam_rot, int32_t *out_updated = nullptr,
              hipStream_t st = nullptr) {
    check(sogm_gridmap_update(h_, depth, cam_pos, cam_rot, out_updated, st), "sogm_gridmap_update");
  }
  // int getInflateOccupancy(Eigen::Vector3d pos) for a batch of device-resident queries
  void getInflateOccupancy(const int32_t *agent_idx, const double *pos, int n, int8_t *out, hipStream_t st = nullptr) {
    check(sogm_gridmap_query_inflate(h_, agent_idx, pos, n, out, st), "sogm_gridmap_query_inflate");
  }

 private:
  sogm_gridmap *h_ = nullptr;
};

// ---- planner: search (KinodynamicAstar-style), CorridorGen, PolyTrajOptimizer, replan ---------------
class Planner {
 public:
  Planner(RiskMap &map, const SogmAstarParams &ap, const SogmPlannerParams &pp, const SogmQpSettings &qs)
      : map_(map), pp_(pp) {
    check(sogm_planner_create(map.ctx(), &ap, &pp, &qs, &p_), "sogm_planner_create");
  }
  ~Planner() { sogm_planner_destroy(p_); }
  sogm_planner *handle() { return p_; }

  // ASTAR_RET search(start_p, start_v, start_a, end_p, ...) + getPathWithVel(corridor_tau) — batched
  void search(const double *start_pva, const double *goal, const double *t_start, int32_t *ret, double *route,
              int32_t *route_len, int route_cap, int32_t *stats, hipStream_t st = nullptr) {
    check(sogm_astar_search(p_, start_pva, goal, t_start, ret, route, route_len, route_cap, stats, nullptr, 0, st),
          "sogm_astar_search");
  }
  // CorridorGen: getObstaclePoints + firi::firi + ShrinkCorridor + validity/intersection/goal LPs — batched
  void generateCorridors(const double *start_pva, const double *t_start, const double *route,
                         const int32_t *route_len, int route_cap, double *polys, int32_t *nfaces, int32_t *npoly,
                         double *goal_pv, hipStream_t st = nullptr) {
    check(sogm_corridor_generate(p_, start_pva, t_start, route, route_len, route_cap, polys, nfaces, npoly,
                                 goal_pv, st), "sogm_corridor_generate");
  }
  // PolyTrajOptimizer::optimize == BezierOpt::setup + optimize — batched
  void optimize(const double *start_pva, const double *goal_pv, const double *polys, const int32_t *nfaces,
                const int32_t *npoly, double *cpts, int32_t *status, int32_t *iters, hipStream_t st = nullptr) {
    check(sogm_bezier_qp_solve(p_, start_pva, goal_pv, polys, nfaces, npoly, cpts, status, iters, st),
          "sogm_bezier_qp_solve");
  }
  // Publication and pre-stamp of the tick loop (no reference counterparts: plan_manager.cpp:176-196 / :364-399 keep
  // the latest trajectory per drone on the host; here the replan's finishing kernel does, and — with a pre-stamp
  // registered — the replan also builds the next tick's start states and map, see sogm_abi.h)
  void setPublish(SogmTrajRecord *own_records, SogmTrajRecord *next_table = nullptr) {
    check(sogm_planner_set_publish(p_, own_records, next_table), "sogm_planner_set_publish");
  }
  void setPrestamp(const SogmPrestamp *next_tick) {
    check(sogm_planner_set_prestamp(p_, next_tick), "sogm_planner_set_prestamp");
  }
  // n ticks of every agent in one call, every agent on its own clock (the reference's drones each run their own FSM,
  // plan_manager.cpp:92-233): sogm_flight_run, see sogm_abi.h "Flight".  Returns the device's verdict after a
  // synchronisation when `wait` is set: false = a device-side wait timed out (sogm_flight_stats hdr[4]).
  bool flight(const SogmFlight &f, hipStream_t st = nullptr, bool wait = false) {
    check(sogm_flight_run(p_, &f, st), "sogm_flight_run");
    if (!wait) return true;
    int32_t hdr[32];
    check(sogm_flight_stats(p_, nullptr, hdr), "sogm_flight_stats");
    return hdr[4] == 0 && hdr[15] == 0;  // no time-out, and every workgroup of the flight was resident from the start
  }
  // bool BaselinePlanner::replan(t, start_pos, start_vel, start_acc, goal_pos) — batched
  void replan(const double *start_pva, const double *goal, const double *t_start, const int32_t *drone_ids,
              SogmTrajRecord *records, int32_t *ok, hipStream_t st = nullptr) {
    check(sogm_replan(p_, start_pva, goal, t_start, drone_ids, records, ok, st), "sogm_replan");
  }

 private:
  RiskMap          &map_;
  SogmPlannerParams pp_;
  sogm_planner     *p_ = nullptr;
};

}  // namespace sogm_host
