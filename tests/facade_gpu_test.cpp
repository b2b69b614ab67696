// End-to-end use of the C++ facade (pred-occ-planner_amd/host/sogm_facade.hpp) on a GPU: the call sequence of
// BaselinePlanner::replan written against the facade classes — map update, collision queries, obstacle
// points, search, corridors, optimisation, replan — on a two-agent scene with one pillar between them.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "sogm_facade.hpp"

using namespace sogm_host;

#define REQUIRE(cond)                                               \
  do {                                                              \
    if (!(cond)) {                                                  \
      std::printf("REQUIRE failed: %s (line %d)\n", #cond, __LINE__); \
      return 1;                                                     \
    }                                                               \
  } while (0)

int main() {
  if (sogm_device_count() < 1) {
    std::puts("no device");
    return 77;
  }
  SogmSpec spec{};
  spec.L = 66; spec.W = 66; spec.H = 20; spec.T = 6;
  spec.resolution = 0.15f; spec.time_resolution = 0.2f; spec.risk_threshold = 0.2f; spec.clearance = 0.45f;
  spec.ground_height = -0.01f; spec.ceiling_height = 3.0f; spec.risk_threshold_region = 1.2f;
  spec.risk_thres_reg_decay = 0.2f; spec.risk_thres_vox_decay = 0.2f;
  spec.map_kind = SOGM_MAP_FAKE; spec.storage = SOGM_STORE_F32;
  const int A = 2;
  RiskMap map(spec, A);
  // ParticleATC::initEgoParticles(0.4, 0.4, 0.45), STEP 0.15
  std::vector<Vec3> body;
  for (double x = -0.2; x <= 0.2; x += 0.15)
    for (double y = -0.2; y <= 0.2; y += 0.15)
      for (double z = -0.225; z <= 0.225; z += 0.15) body.push_back({x, y, z});
  map.setCoordinator(body);
  // one static pillar at the origin: shell points on a 0.1 m lattice, z in [0, 3)
  std::vector<float> cloud;
  for (int k = 0; k < 48; ++k)
    for (int iz = 0; iz < 30; ++iz) {
      cloud.push_back(0.4f * std::cos(k * 0.1309f));
      cloud.push_back(0.4f * std::sin(k * 0.1309f));
      cloud.push_back(0.1f * iz);
    }
  const int n_pts = (int)cloud.size() / 3;
  SogmCylinder cyl{};
  cyl.type = 3; cyl.x = 0; cyl.y = 0; cyl.z = 1.5; cyl.w = 0.8; cyl.h = 3.0; cyl.qw = 1.0;
  const float  poses[6]  = {-3.f, 0.1f, 1.f, 3.f, -0.1f, 1.f};
  const double stamps[2] = {100.0, 100.0};
  const int32_t range[4] = {0, n_pts, 0, n_pts};
  DevBuf<float> d_cloud, d_poses; DevBuf<int32_t> d_range; DevBuf<SogmCylinder> d_cyl; DevBuf<double> d_stamps;
  d_cloud.put(cloud.data(), cloud.size()); d_poses.put(poses, 6); d_range.put(range, 4); d_cyl.put(&cyl, 1);
  d_stamps.put(stamps, 2);
  map.update(d_cloud.data(), d_range.data(), d_cyl.data(), 1, d_poses.data(), d_stamps.data());
  // getClearOcccupancy: 1 at the pillar, 0 in free space, -1 above the ceiling (fake map)
  REQUIRE(map.getClearOcccupancy(0, {-0.45, 0.0, 1.0}, 0.0) == 1);
  REQUIRE(map.getClearOcccupancy(0, {-2.0, 1.5, 1.0}, 0.3) == 0);
  REQUIRE(map.getClearOcccupancy(0, {-2.0, 1.5, 3.5}, 0.0) == -1);
  std::vector<Vec3> pts;
  map.getObstaclePoints(0, pts, 100.0, 100.3, {-1.0, -1.0, 0.5}, {1.0, 1.0, 1.5});
  REQUIRE(pts.size() > 10);
  // planner: search -> corridors -> optimise, then the fused replan
  SogmAstarParams ap{}; ap.max_tau = 2.0; ap.max_vel = 2.0; ap.max_acc = 6.0; ap.w_time = 5.0; ap.horizon = 5.0;
  ap.lambda_heu = 5.0; ap.resolution = 0.15; ap.time_resolution = 0.3; ap.allocate_num = 10000; ap.check_num = 1;
  ap.tolerance = 1;
  SogmPlannerParams pp{}; pp.corridor_tau = 0.3; pp.init_range = 1.2; pp.shrink_size = 0.2; pp.opt_max_vel = 3.0;
  pp.opt_max_acc = 6.0; pp.fake_planner = 1; pp.firi_iterations = 2; pp.pc_capacity = 16384; pp.max_faces = 64;
  SogmQpSettings qs{}; qs.rho = 0.1; qs.sigma = 1e-6; qs.alpha = 1.6; qs.eps_abs = 1e-3; qs.eps_rel = 1e-3;
  qs.max_iter = 4000; qs.check_termination = 25; qs.scaling_iters = 10; qs.adaptive_rho_interval = 25;
  Planner planner(map, ap, pp, qs);
  const double pva[18]  = {-3, 0.1, 1, 0, 0, 0, 0, 0, 0, 3, -0.1, 1, 0, 0, 0, 0, 0, 0};
  const double goal[6]  = {3, 0.1, 1, -3, -0.1, 1};
  const double tst[2]   = {100.05, 100.05};
  const int32_t ids[2]  = {0, 1};
  const int    cap = 64, MF = 64;
  DevBuf<double> d_pva, d_goal, d_t, d_route((size_t)A * cap * 6), d_polys((size_t)A * SOGM_MAX_PIECES * MF * 4),
      d_gpv(A * 6), d_cpts((size_t)A * SOGM_MAX_PIECES * 15);
  DevBuf<int32_t> d_ret(A), d_len(A), d_stats(A * 4), d_nf(A * SOGM_MAX_PIECES), d_np(A), d_st(A), d_it(A), d_ids, d_ok(A);
  DevBuf<SogmTrajRecord> d_rec(A);
  d_pva.put(pva, 18); d_goal.put(goal, 6); d_t.put(tst, 2); d_ids.put(ids, 2);
  planner.search(d_pva.data(), d_goal.data(), d_t.data(), d_ret.data(), d_route.data(), d_len.data(), cap, d_stats.data());
  int32_t ret[2], len[2];
  d_ret.get(ret, 2); d_len.get(len, 2);
  REQUIRE(ret[0] >= 3 && ret[1] >= 3);  // REACH_HORIZON / REACH_END / NEAR_END (dyn_a_star.h:15)
  REQUIRE(len[0] >= 2 && len[1] >= 2);
  planner.generateCorridors(d_pva.data(), d_t.data(), d_route.data(), d_len.data(), cap, d_polys.data(), d_nf.data(),
                            d_np.data(), d_gpv.data());
  int32_t np[2];
  d_np.get(np, 2);
  REQUIRE(np[0] >= 1 && np[1] >= 1);
  planner.optimize(d_pva.data(), d_gpv.data(), d_polys.data(), d_nf.data(), d_np.data(), d_cpts.data(), d_st.data(),
                   d_it.data());
  int32_t st[2];
  d_st.get(st, 2);
  REQUIRE(st[0] == 1 || st[0] == 2);
  planner.replan(d_pva.data(), d_goal.data(), d_t.data(), d_ids.data(), d_rec.data(), d_ok.data());
  int32_t ok[2];
  SogmTrajRecord rec[2];
  d_ok.get(ok, 2); d_rec.get(rec, 2);
  REQUIRE(ok[0] == 1 && rec[0].n_pieces == np[0] && rec[0].drone_id == 0);
  REQUIRE(std::fabs(rec[0].cpts[0] - (-3.0)) < 2e-3 && std::fabs(rec[0].cpts[1] - 0.1) < 2e-3);  // starts at the start
  // wire format round trip of the planned trajectory
  BezierTrajMsg msg = msgFromRecord(rec[0], 1, 100.06);
  SogmTrajRecord back;
  REQUIRE(recordFromMsg(msg, back) && back.n_pieces == rec[0].n_pieces);
  // the second agent now sees the first one's trajectory in its map
  map.addOtherAgents(d_rec.data(), A, d_ids.data());
  // ---- ABI version 5 (6: SogmFlight::nccl_comm): per-update sensor frames and flights through the facade ----------------------------------------
  static_assert(sizeof(SogmWorld) == 3 * sizeof(void *) + 4 * sizeof(int32_t), "SogmWorld layout (the Python binding mirrors it)");
  static_assert(sizeof(SogmFlight) == 2 * 4 + 3 * 8 + 6 * sizeof(void *) + 2 * 4 + 3 * sizeof(void *), "SogmFlight layout");
  REQUIRE(sogm_abi_version() == SOGM_ABI_VERSION);
  {
    // the same frame through sogm_update_world (device-side crop through the block index): same answers
    RiskMap map2(spec, A);
    map2.setCoordinator(body);
    const int BP = 64, nb = (n_pts + BP - 1) / BP;
    DevBuf<float> d_bounds((size_t)nb * 4);
    RiskMap::blockBounds(d_cloud.data(), n_pts, BP, d_bounds.data());
    const SogmWorld frame = RiskMap::world(d_cloud.data(), n_pts, d_bounds.data(), BP, d_cyl.data(), 1);
    map2.updateWorld(frame, d_poses.data(), d_stamps.data());
    REQUIRE(map2.getClearOcccupancy(0, {-0.45, 0.0, 1.0}, 0.0) == 1);
    REQUIRE(map2.getClearOcccupancy(0, {-2.0, 1.5, 1.0}, 0.3) == 0);
    std::vector<Vec3> pts2;
    map2.getObstaclePoints(0, pts2, 100.0, 100.3, {-1.0, -1.0, 0.5}, {1.0, 1.0, 1.5});
    REQUIRE(pts2.size() == pts.size());
    for (size_t i = 0; i < pts.size(); ++i) REQUIRE(pts2[i] == pts[i]);
    // a flight of four ticks: both agents cross the pillar's neighbourhood, every agent on its own clock
    Planner planner2(map2, ap, pp, qs);
    const int K = 4;
    std::vector<SogmWorld> frames(K, frame);  // (a static world: the same frame for every tick)
    const double hover0[18] = {-3, 0.1, 1, 0, 0, 0, 0, 0, 0, 3, -0.1, 1, 0, 0, 0, 0, 0, 0};
    DevBuf<double> d_hover, d_goal3;
    const double goal3[6] = {3, 0.1, 1, -3, -0.1, 1};
    d_hover.put(hover0, 18); d_goal3.put(goal3, 6);
    DevBuf<SogmTrajRecord> d_own(A), d_tables((size_t)4 * A), d_log((size_t)K * A);
    DevBuf<int32_t> d_logok((size_t)K * A);
    std::vector<SogmTrajRecord> zero((size_t)4 * A);
    std::memset(zero.data(), 0, sizeof(SogmTrajRecord) * zero.size());
    d_tables.put(zero.data(), zero.size()); d_own.put(zero.data(), A);
    SogmFlight f{};
    f.n_ticks = K; f.first_tick = 0; f.t0 = 100.0; f.period = 0.1; f.replan_start_offset = 0.02; f.worlds = frames.data();
    f.goals = d_goal3.data(); f.drone_ids = d_ids.data(); f.hover_inout = d_hover.data(); f.own_inout = d_own.data();
    f.tables = d_tables.data(); f.n_total = A; f.agent0 = 0; f.log_records = d_log.data(); f.log_ok = d_logok.data();
    REQUIRE(planner2.flight(f, nullptr, /*wait*/ true));
    std::vector<int32_t> lok((size_t)K * A);
    std::vector<SogmTrajRecord> lrec((size_t)K * A);
    d_logok.get(lok.data(), lok.size()); d_log.get(lrec.data(), lrec.size());
    int n_ok = 0;
    for (int i = 0; i < K * A; ++i) n_ok += lok[i];
    REQUIRE(n_ok >= K);                                   // (replans succeed)
    REQUIRE(lok[0] == 1 && lrec[0].n_pieces >= 1 && lrec[0].drone_id == 0);
    REQUIRE(std::fabs(lrec[0].cpts[0] - (-3.0)) < 2e-3);  // tick 0 starts where the agent hovers
    for (int k = 0; k < K; ++k) REQUIRE(lrec[(size_t)k * A + 1].drone_id == 1 && std::fabs(lrec[(size_t)k * A].time_start - (100.02 + 0.1 * k)) < 1e-9);
    std::printf("flight through the facade: %d of %d agent-ticks ok\n", n_ok, K * A);
  }
  std::printf("facade gpu ok: A* ret %d/%d, %d/%d pieces, QP status %d, %zu obstacle points\n", ret[0], ret[1], np[0], np[1],
              st[0], pts.size());
  return 0;
}
