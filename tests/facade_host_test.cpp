// Host-only checks of the wire-format helpers of sogm_facade.hpp (no device call is made).
#include <cmath>
#include <cstdio>

#include "sogm_facade.hpp"

using namespace sogm_host;

int main() {
  BezierTrajMsg m;
  m.drone_id   = 3;
  m.traj_id    = 7;
  m.start_time = 1234.5;
  m.duration   = {0.3f, 0.3f};
  for (int k = 0; k < 10; ++k) m.cpts.push_back({0.1 * k, -0.2 * k, 1.0});
  SogmTrajRecord r;
  if (!recordFromMsg(m, r)) return 1;
  if (r.drone_id != 3 || r.n_pieces != 2 || r.time_start != 1234.5) return 2;
  if (r.duration[0] != (double)0.3f || r.duration[2] != 0.0) return 3;  // float32 on the wire
  if (r.cpts[9 * 3 + 1] != -0.2 * 9 || r.cpts[10 * 3] != 0.0) return 4;
  BezierTrajMsg b = msgFromRecord(r, 8, 2.0);
  if (b.duration.size() != 2 || b.cpts.size() != 10 || b.cpts[4][0] != 0.1 * 4 || b.traj_id != 8) return 5;
  m.duration.assign(17, 0.1f);
  m.cpts.assign(85, {0, 0, 0});
  if (recordFromMsg(m, r)) return 6;  // more pieces than the fixed-size record holds
  // future-risk message: V = 4, T = 3, stride 3
  std::vector<float> grid(12);
  for (int i = 0; i < 12; ++i) grid[i] = 0.25f * i;
  const float pose[3] = {1.f, 2.f, 3.f};
  std::vector<float> msg = futureRiskMsg(grid, pose, 77.0);
  if (msg.size() != 16 || msg[12] != 1.f || msg[15] != 77.f) return 7;
  std::vector<float> g2;
  float  p2[3];
  double st;
  if (!splitFutureRiskMsg(msg, 4, 3, 3, g2, p2, st)) return 8;
  if (g2 != grid || p2[2] != 3.f || st != 77.0) return 9;
  if (splitFutureRiskMsg(msg, 5, 3, 3, g2, p2, st)) return 10;
  std::puts("facade host ok");
  return 0;
}
