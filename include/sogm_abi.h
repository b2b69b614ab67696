/*
 * sogm_abi.h — the extern "C" drop-in boundary of the MI355X-native SOGM replan hot path.
 *
 * The reference (siyuanwu99/pred-occ-planner) has no FFI/plugin layer: plan_manager links the
 * map / search / corridor / optimiser class libraries directly (plan_manager/CMakeLists.txt:15-32)
 * and calls them through shared_ptr members (plan_manager/include/plan_manager/baseline.h:155-158).
 * The drop-in boundary is therefore (i) the C++ facade classes in
 * the headers under pred-occ-planner_amd/host/, which keep the reference method names, and (ii) this C ABI,
 * which those facades (and the Python ctypes host) forward to.  Each entry point cites the
 * reference interface it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / Eigen / ROS types;
 *   - every array argument marked "dev" is a DEVICE pointer (HBM resident), "host" a host pointer;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - all calls are asynchronous on `stream` unless documented otherwise;
 *   - return value: 0 = SOGM_OK, negative = sogm_status error; no exceptions cross the boundary;
 *   - one context per thread (thread-compatible, not thread-safe), like the reference objects;
 *   - there is NO CPU fallback: without a HIP device every compute call returns SOGM_ERR_NO_DEVICE.
 */
#ifndef SOGM_ABI_H
#define SOGM_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* status                                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef enum sogm_status {
  SOGM_OK              = 0,
  SOGM_ERR_INVALID_ARG = -1,
  SOGM_ERR_NO_DEVICE   = -2, /* no HIP device / HIP runtime error at create time             */
  SOGM_ERR_HIP         = -3, /* a HIP call failed; see sogm_last_error()                     */
  SOGM_ERR_CAPACITY    = -4, /* an output buffer / pool was too small                        */
  SOGM_ERR_STATE       = -5, /* call order violated (e.g. query before any update)           */
  SOGM_ERR_COMM        = -6  /* RCCL could not be loaded / a collective failed; see sogm_last_error() */
} sogm_status;

/* ------------------------------------------------------------------------------------------ */
/* plain-data records                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* Which reference map class the context reproduces. */
enum {
  SOGM_MAP_FAKE      = 0, /* FakeParticleRiskVoxel  plan_env/src/fake_particle_risk_voxel.cpp    */
  SOGM_MAP_RISKBASE  = 1, /* RiskBase               plan_env/src/risk_base.cpp                   */
  SOGM_MAP_RISKVOXEL = 2  /* RiskVoxel (DSP particle map)  plan_env/src/risk_voxel.cpp: queries
                             as RiskBase; the neighbour overlay SETS cells to 1.0
                             (addObstaclesToRiskMap :311-318) instead of adding risk.            */
};

/*
 * Grid + map parameters.  The reference fixes L,W,H,T with macros
 * (plan_env/include/plan_env/map_parameters.h:5-13); here they are runtime values.
 * Field defaults (reference):  66,66,20,6, 0.15, time_resolution 0.2 (map.cpp:30),
 * risk_threshold 0.2 (map.cpp:35), clearance 0.3 (map.cpp:36; sim_fake.yaml:71 -> 0.45),
 * ceiling 2.0 / ground -1.0 (map.cpp:37-38; sim_fake.yaml:68-69 -> 3.0 / -0.01),
 * region threshold 1.2, decays 0.2 (risk_base.cpp:21-23).
 */
enum { SOGM_STORE_F32 = 0, SOGM_STORE_F16 = 1,
       /* OR-ed into SogmSpec.storage: the cells of a slice are kept in 2 x 2 x 2 tiles (L, W, H even) instead of x-fastest
        * rows — same values at the ABI (sogm_download_reference_layout, queries, sogm_set_future_risk convert), about half
        * the 32-byte sectors written by the stamp and zeroed by the sparse reset (obstacle surfaces run along z).
        * sogm_grid_ptr() then points at tiled slices: cell
        * (x, y, z) of slice t at  t V + ((((z>>1) (W>>1) + (y>>1)) (L>>1) + (x>>1)) << 3 | (z&1) << 2 | (y&1) << 1 | (x&1)). */
       SOGM_LAYOUT_TILED = 16 };
typedef struct SogmSpec {
  int32_t L, W, H, T;
  float   resolution;
  float   time_resolution;
  float   risk_threshold;
  float   clearance;
  float   ground_height;
  float   ceiling_height;
  float   risk_threshold_region;
  float   risk_thres_reg_decay;
  float   risk_thres_vox_decay;
  int32_t map_kind; /* SOGM_MAP_FAKE | SOGM_MAP_RISKBASE | SOGM_MAP_RISKVOXEL */
  int32_t storage;  /* SOGM_STORE_F32 (reference: float risk_maps_) | SOGM_STORE_F16 (half the HBM
                       bytes per voxel; marks and neighbour counts up to 2048 stay exact) */
} SogmSpec;

/* Ground-truth obstacle record, field-for-field `struct Cylinder`
 * (plan_env/include/plan_env/fake_particle_risk_voxel.h:31-44). type 3 = cylinder, 2 = ring. */
typedef struct SogmCylinder {
  int32_t type;
  int32_t _pad;
  double  x, y, z, w, h, vx, vy, qw, qx, qy, qz;
} SogmCylinder;

/*
 * One shared trajectory = the payload of traj_utils/msg/BezierTraj.msg:1-9 as stored by
 * ParticleATC::trajectoryCallback (traj_coordinator/src/particles.cpp:131-191) in
 * SwarmParticleTraj (traj_coordinator/include/traj_coordinator/particle.hpp:30-37).
 * Fixed size so that one RCCL all-gather moves the whole swarm's records.
 * n_pieces == 0 means "no trajectory received from this drone".
 * cpts: piece-major, 5 control points per piece, xyz-minor (row k = piece*5 + j).
 */
#define SOGM_MAX_PIECES 16
typedef struct SogmTrajRecord {
  int32_t drone_id;
  int32_t n_pieces;
  double  time_start; /* absolute seconds (traj_msg->start_time) */
  double  duration[SOGM_MAX_PIECES];
  double  cpts[SOGM_MAX_PIECES * 5 * 3];
} SogmTrajRecord;

/* Search parameters = FakeRiskHybridAstar::setParam (path_searching/src/fake_risk_hybrid_a_star.cpp:62-82),
 * values from plan_manager/config/sim_fake.yaml:15-29. */
typedef struct SogmAstarParams {
  double  max_tau;
  double  max_vel;
  double  max_acc;
  double  w_time;
  double  horizon;
  double  lambda_heu;
  double  resolution;      /* search/resolution_astar */
  double  time_resolution; /* search/time_resolution  */
  int32_t allocate_num;
  int32_t check_num;
  int32_t tolerance; /* search/tolerance, default 1 */
  int32_t shot_ignores_time; /* 0 = FakeRiskHybridAstar: the shot trajectory is checked with
                                getClearOcccupancy(coord, time) (fake_risk_hybrid_a_star.cpp:521);
                                1 = RiskHybridAstar: getClearOcccupancy(coord), i.e. slice 0
                                (risk_hybrid_a_star.cpp:514, risk_base.cpp:251-253) — the only difference
                                between the two classes.  sogm_planner_create() sets it from
                                SogmPlannerParams.fake_planner (BaselinePlanner owns a RiskHybridAstar,
                                FakeBaselinePlanner a FakeRiskHybridAstar: baseline.h:155, baseline_fake.h). */
} SogmAstarParams;

/* Planner parameters = BaselineParameters (plan_manager/include/plan_manager/baseline.h:45-94). */
typedef struct SogmPlannerParams {
  double  corridor_tau;
  double  init_range;
  double  shrink_size;
  double  opt_max_vel; /* finite and positive: the QP's velocity / acceleration rows are two-sided boxes; values */
  double  opt_max_acc; /* >= 1e20 ("unbounded") are rejected by sogm_planner_create (SOGM_ERR_INVALID_ARG)      */
  int32_t fake_planner; /* 1 = FakeBaselinePlanner rules (baseline_fake.cpp), 0 = BaselinePlanner */
  int32_t firi_iterations; /* 2 at baseline.cpp:352 */
  int32_t pc_capacity;     /* max obstacle points per corridor box */
  int32_t max_faces;       /* max faces kept per polytope */
} SogmPlannerParams;

/* QP solver settings = OSQP v0.6 defaults as used through IOSQP (traj_opt/include/iosqp.hpp:40-115,
 * traj_opt/src/bezier_optimizer.cpp:269).  adaptive_rho_interval: 0 = fixed rho (one KKT factor
 * for the whole solve); k > 0 = OSQP's adaptive rho evaluated every k iterations (OSQP's
 * wall-clock-derived interval lands on its check_termination multiple, 25, for problems this size).
 * residual_fp32 (BASELINE configs[4], "mixed-precision ADMM residuals"; 0 = off, the default): the termination
 * checks evaluate the residual norms, their normalisations and the workgroup reductions in fp32 (the iteration
 * itself, the rho estimate's inputs and the infeasibility certificate's A'dy stay fp64).  A check near its threshold
 * may then pass one check earlier or later than OSQP's fp64 test: the contract of this mode is "same status,
 * coefficients within 1e-4 of the fp64 solve", not "same iteration count". */
typedef struct SogmQpSettings {
  double  rho;
  double  sigma;
  double  alpha;
  double  eps_abs;
  double  eps_rel;
  int32_t max_iter;
  int32_t check_termination;
  int32_t scaling_iters;
  int32_t adaptive_rho_interval;
  int32_t residual_fp32;
  int32_t reserved_;
} SogmQpSettings;

typedef struct sogm_ctx sogm_ctx;

/* This header is what a plan_manager host binds: contexts, the map updates, the overlay, the queries, search, corridors,
 * QP, replan, the flight, the trajectory exchange, version and errors.  Everything a host does NOT need to fly — tuning
 * knobs, per-kernel profiling, device clocks, traffic counters, parity downloads of internal state, test hooks — is
 * declared in include/sogm_abi_debug.h (same library, same version number). */
/* ------------------------------------------------------------------------------------------ */
/* library                                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* ABI version of this header; bumps on any change of a signature OR of the size of a buffer an entry point writes
 * (version 5: sogm_sparse_reset_state writes out[8] and sogm_profile_read SOGM_PROF_N = 8 doubles, where version 4's
 * first revision wrote 7; SogmWorld / the world-frame update entries / the flight entries were added; version 6: SogmFlight
 * gained nccl_comm, sogm_flight_stats reports hdr[15], the profiling / tuning / debug entries moved to sogm_abi_debug.h).  A host
 * compares sogm_abi_version() with the SOGM_ABI_VERSION it was compiled against before any other call: the Python
 * binding and host/sogm_facade.hpp refuse a library of another version. */
#define SOGM_ABI_VERSION 6
int         sogm_abi_version(void);
/* Text of the last HIP error seen by this thread ("" if none). */
const char *sogm_last_error(void);
/* Number of visible HIP devices (0 without a GPU; never fails). */
int         sogm_device_count(void);

/* ------------------------------------------------------------------------------------------ */
/* map context:  MapBase::init / RiskBase::init / FakeParticleRiskVoxel::init                   */
/*   (plan_env/src/map.cpp:42-105, risk_base.cpp:15-58, fake_particle_risk_voxel.cpp:20-73)    */
/* ------------------------------------------------------------------------------------------ */
/* Allocates the batched SOGM  sogm[n_agents][T][H][W][L]  (fp32 or fp16 per spec->storage, time-major
 * slabs) on `device`. */
int  sogm_create(const SogmSpec *spec, int n_agents, int device, sogm_ctx **out);
void sogm_destroy(sogm_ctx *ctx);
/* Bytes of HBM held by the grid. */
int64_t sogm_grid_bytes(const sogm_ctx *ctx);
/* Device pointer of the grid (layout above; __half elements when storage is SOGM_STORE_F16).  The caller may write
 * through it: the grid's next reset is then the dense clear (see sogm_set_sparse_reset). */
float  *sogm_grid_ptr(sogm_ctx *ctx);

/* Sparse reset of the map.  The reference rebuilds the SOGM from zero at every update
 * (fake_particle_risk_voxel.cpp:107-108, a fill over all V x T cells, then the marks).  The library produces the
 * same cells without touching the ones that are zero already: every mark written by sogm_update_gt[_swarm] /
 * sogm_project_neighbours is logged as the index of its 32-byte sector (one log per agent and per grid of the pool,
 * log_capacity entries per agent: 0 keeps the current value; default V*T/80, at least 2^20, or SOGM_LOG_CAP), and the grid's next reset
 * zeroes exactly the logged sectors.  A grid written by a dense writer (sogm_set_future_risk, sogm_dsp_publish, the
 * caller through sogm_grid_ptr), a freshly allocated one, and an agent whose log overflowed are cleared densely.
 * enable = 0 restores the dense clear everywhere (also: SOGM_SPARSE_RESET=0 in the environment at sogm_create).
 * Synchronises the device; call between ticks. */
int sogm_set_sparse_reset(sogm_ctx *ctx, int enable, int log_capacity);
/* Resample branch of ParticleATC::getParticlesWithRisk (particles.cpp:365-409; swarm/replan_risk_rate and
 * swarm/num_resample, particles.cpp:33-34 — 0.00 in every shipped configuration, which is also the default here).  With
 * a rate > 0 a neighbour's body particle at slice time t is replaced, once replan_risk_rate * (t - time_start) >= 1e-3,
 * by num_resample Gaussian samples weighted exp(-|n|^2 / 2 sigma^2), the weights normalised to num_resample per
 * particle, and the map receives the weights instead of 1.0.  The reference draws the noise from
 * std::default_random_engine(time(NULL)), re-seeded at every call: every call of one wall-clock second replays one
 * sequence.  The host injects that sequence as a table of standard normals (device, float, kept alive by the caller,
 * >= 3 * body particles * num_resample entries): sample i of particle e uses entries 3 (e n + i) + {0, 1, 2} times
 * sigma.  Applies to sogm_project_neighbours / sogm_update_gt_swarm / sogm_update_prestamped on FAKE and RISKBASE maps
 * with fp32 cells (RiskVoxel's overlay does not resample, risk_voxel.cpp:258-339).  Call after
 * sogm_set_body_particles.  rate 0 or num_resample 0 switches it off. */
int sogm_set_resample(sogm_ctx *ctx, float replan_risk_rate, int num_resample, const float *normal_table_dev,
                      int n_table);

/* Body particles of one drone: ParticleATC::initEgoParticles (particles.cpp:62-87).
 * host: xyz[n*3] offsets (fp64).  Every drone of the swarm uses the same set.
 * The overlay places ANOTHER drone's particles, and those reach a ParticleATC through particlesCallback
 * (particles.cpp:89-108) as geometry_msgs/Polygon points — Point32, i.e. float32 coordinates cast back to double.  A host
 * that wants the reference's cells to the last bit passes the offsets as received ((double)(float)x; the Python mirror's
 * scene.received_body_particles does); the library uses what it is given. */
int sogm_set_body_particles(sogm_ctx *ctx, const double *xyz_host, int n);


/*
 * Tick pipelining (mode): 0 = off (every update clears its grid in stream order).
 * 1 = in-place pre-clear in two parts on an internal side stream: a narrow launch clears the head of the grid
 *     as soon as sogm_replan()'s obstacle-point kernels have read the SOGM (it shares the machine with the FIRI
 *     kernels), and a full-width launch clears the rest once the corridor stage is done with global memory (only
 *     the LDS-resident QP runs beside it).  While that pre-clear is pending the map counts as "not
 *     updated": queries return SOGM_ERR_STATE until the next update.
 * 2 = double-buffered: a second grid is allocated (SOGM_ERR_CAPACITY if HBM has no room; the mode is then
 *     unchanged) and sogm_replan() clears it under the WHOLE replan; the next update swaps it in.  The current
 *     map stays valid for queries.  sogm_grid_ptr() changes at every update in this mode.
 * The next sogm_update_gt() / sogm_set_future_risk() / sogm_dsp_publish() waits for the pending clear instead
 * of issuing its own.  Changing the mode while a pre-clear is in flight synchronises the device and drops it.
 */
int sogm_set_overlap_clear(sogm_ctx *ctx, int mode);


/* ------------------------------------------------------------------------------------------ */
/* SOGM update                                                                                 */
/* ------------------------------------------------------------------------------------------ */
/*
 * FakeParticleRiskVoxel::updateMap (fake_particle_risk_voxel.cpp:80-222) without the neighbour
 * overlay, for every agent of the batch:
 *   crop cloud to pose +- range (:88-104), zero the grid (:107-108), mark slice 0 (:111-116),
 *   for each occupied voxel look up the GT velocity (:127-154) and stamp slices 1..T-1 (:155-160).
 * dev  cloud_xyz     [n_points*3] fp32, world frame
 * dev  cloud_range   [n_agents*2] int32 {begin,end} point indices seen by each agent
 * dev  cylinders     [n_cyl]      SogmCylinder          (shared GT state, may be NULL if n_cyl==0)
 * dev  poses         [n_agents*3] fp32  map centre (MapBase::pose_)
 * dev  stamps        [n_agents]   fp64  last_update_time_ (absolute seconds)
 */
int sogm_update_gt(sogm_ctx *ctx, const float *cloud_xyz, const int32_t *cloud_range,
                   const SogmCylinder *cylinders, int n_cyl, const float *poses,
                   const double *stamps, void *stream);

/*
 * One sensor frame of the fake-perception map — what MapBase::cloudCallback (plan_env/src/map.cpp:170-171) and
 * FakeParticleRiskVoxel::groundTruthStateCallback (fake_particle_risk_voxel.cpp:244-264) delivered for ONE update —
 * with the spatial index the device-side crop needs: the cloud in blocks of `block_points` consecutive points and the
 * xy bounds of every block (sogm_cloud_block_bounds computes them; a sensor driver that emits obstacles one after the
 * other gets tight blocks for free).  updateMap's PassThrough crop (fake_particle_risk_voxel.cpp:88-104) then runs on the
 * device, per agent, around the agent's CURRENT map centre: one wave lists the blocks that intersect the window, the
 * stamp walks the list.  All pointers are device pointers; the struct itself is read on the host at call time.
 */
typedef struct SogmWorld {
  const float        *cloud_xyz;    /* dev [n_points*3] fp32, world frame                                         */
  const float        *block_bounds; /* dev [n_blocks*4] {xmin, xmax, ymin, ymax} of points [b*block_points, ...)  */
  const SogmCylinder *cylinders;    /* dev [n_cyl] GT obstacle states of the same instant (NULL if n_cyl == 0)     */
  int32_t             n_points;
  int32_t             n_blocks;     /* ceil(n_points / block_points)                                              */
  int32_t             block_points; /* 64 .. 4096                                                                 */
  int32_t             n_cyl;
} SogmWorld;
/* xy bounds of every block of `block_points` consecutive points of a cloud; dev out_bounds[ceil(n/block_points)*4]. */
int sogm_cloud_block_bounds(const float *cloud_xyz, int n_points, int block_points, float *out_bounds, void *stream);
/*
 * FakeParticleRiskVoxel::updateMap for every agent of the batch from one SogmWorld frame: as sogm_update_gt_swarm, with
 * the crop done on the device around `poses` (no per-agent ranges from the host).  records may be NULL (n_records 0):
 * no overlay.  The maps are identical to sogm_update_gt[_swarm] fed with any superset of each agent's window.
 */
int sogm_update_world(sogm_ctx *ctx, const SogmWorld *world, const float *poses, const double *stamps,
                      const SogmTrajRecord *records, int n_records, const int32_t *ego_ids, void *stream);

/*
 * Neighbour overlay: RiskBase::addOtherAgents (risk_base.cpp:136-168) ==
 * fake_particle_risk_voxel.cpp:178-218, through ParticleATC::getParticlesWithRisk
 * (particles.cpp:346-422, replan_risk_rate == 0 branch) and addParticlesToRiskMap
 * (risk_base.cpp:199-208).  Uses the poses/stamps of the last update.
 * dev  records  [n_records] SogmTrajRecord  (the swarm's latest trajectories)
 * dev  ego_ids  [n_agents]  int32           drone_id of each batch agent (skipped as "ego")
 */
int sogm_project_neighbours(sogm_ctx *ctx, const SogmTrajRecord *records, int n_records,
                            const int32_t *ego_ids, void *stream);

/*
 * FakeParticleRiskVoxel::updateMap as ONE call: sogm_update_gt followed by the neighbour overlay the reference runs
 * at the end of the same function (plan_env/src/fake_particle_risk_voxel.cpp:175-226; RiskBase / RiskVoxel:
 * risk_base.cpp:71, risk_voxel.cpp:179).  Arguments as sogm_update_gt + sogm_project_neighbours; the resulting maps
 * are identical to the two separate calls.  Needs sogm_set_body_particles().
 */
int sogm_update_gt_swarm(sogm_ctx *ctx, const float *cloud_xyz, const int32_t *cloud_range,
                         const SogmCylinder *cylinders, int n_cyl, const float *poses, const double *stamps,
                         const SogmTrajRecord *records, int n_records, const int32_t *ego_ids, void *stream);

/*
 * RiskBase::futureRiskCallback (risk_base.cpp:60-80): adopt an externally produced SOGM.
 * dev  grid_vt  [n_agents][V][T] fp32 in the REFERENCE layout (voxel-major); transposed into slabs.
 */
int sogm_set_future_risk(sogm_ctx *ctx, const float *grid_vt, const float *poses,
                         const double *stamps, void *stream);


/*
 * RiskBase::getMapTime().toSec() and getMapCenter() of one agent (plan_env/include/plan_env/risk_base.h:70,76;
 * map.h:109-111): the stamp and pose the last update / sogm_set_future_risk adopted.  Host outputs (either may be
 * NULL); waits for `stream`.  SOGM_ERR_STATE before the first update.
 */
int sogm_map_state(sogm_ctx *ctx, int agent, double *out_map_time_host, float *out_center_host /* [3] */,
                   void *stream);

/*
 * Bezier::getPos / getVel / getAcc (traj_utils/include/traj_utils/bernstein.hpp:174-187,
 * traj_utils/src/bernstein.cpp:25-59) for a batch of shared trajectories:
 * out_pva[i] = {pos, vel, acc} of records[i] at absolute time t[i] (clamped to the trajectory's
 * span, as FiniteStateMachine does when it samples the replan start state,
 * plan_manager/src/plan_manager.cpp:169-175).  out_valid[i] = 0 when the record holds no trajectory.
 * dev records[n], dev t[n] fp64, dev out_pva[n*9] fp64, dev out_valid[n] int32.
 */
int sogm_traj_eval(const SogmTrajRecord *records, int n, const double *t, double *out_pva,
                   int32_t *out_valid, void *stream);


/*
 * MapBase::filterPointCloud (plan_env/src/map.cpp:107-132; duplicate risk_mapping_node.cpp:74-99) for
 * every agent: pcl::VoxelGrid centroid filter with leaf `filter_res` (one fp32 centroid per occupied
 * leaf, emitted in ascending leaf index — PCL >= 1.8 applyFilter, third-party), camera->body axis
 * swap (x = z, y = -x, z = -y), isInRange against the map's local range, stop at `cap` points
 * (5000 in the reference).  The output feeds sogm_update_dsp as `points`.
 * dev raw_xyz   [n_total*3] fp32 camera-frame points (non-finite points are skipped)
 * dev raw_range [n_agents*2] int32 {begin,end}
 * dev out_xyz   [n_agents*cap*3] fp32,  dev out_count [n_agents] int32 (-1: the cloud's bounding box
 *               has more leaves than the accumulator array, see sogm_filter_reserve)
 */
int sogm_filter_point_cloud(sogm_ctx *ctx, const float *raw_xyz, const int32_t *raw_range,
                            float filter_res, int cap, float *out_xyz, int32_t *out_count,
                            void *stream);
/* Leaf accumulators per agent (default 2^20 = a 15 m cube at 0.15 m); call before the first filter call. */
int sogm_filter_reserve(sogm_ctx *ctx, int max_cells_per_agent);

/* ------------------------------------------------------------------------------------------ */
/* particle-filter SOGM:  dsp_map::DSPMap  (plan_env/include/plan_env/dsp_dynamic.h)            */
/*   as owned and configured by RiskVoxel (plan_env/src/risk_voxel.cpp:42-50,237-254)          */
/* ------------------------------------------------------------------------------------------ */
/* Compile-time constants of the reference (plan_env/include/plan_env/map_parameters.h:5-54) and
 * the DSPMap settings RiskVoxel::init applies, as runtime values.  The grid (L,W,H,T,resolution)
 * comes from the SogmSpec of the map context; T = PREDICTION_TIMES. */
#define SOGM_DSP_MAX_T 16
typedef struct SogmDspParams {
  int32_t max_particle_num_voxel; /* MAX_PARTICLE_NUM_VOXEL 7; slots per voxel = 2x (:51)        */
  int32_t half_fov_h;             /* 43 deg (:22)                                                */
  int32_t half_fov_v;             /* 29 deg (:24)                                                */
  int32_t angle_resolution;       /* ANGLE_RESOLUTION 1 (:9)                                     */
  int32_t newborn_num;            /* setNewBornParticleNumberofEachPoint(20) risk_voxel.cpp:47   */
  int32_t obs_max_per_pyramid;    /* observation_max_points_num_one_pyramid 100 (:51)            */
  float   prediction_times[SOGM_DSP_MAX_T]; /* prediction_future_time {0.3 .. 1.8} (:19)         */
  float   sigma_observation;      /* map/sigma_observation 0.05 (risk_voxel.cpp:20,45)           */
  float   p_detection;            /* 0.95 (dsp_dynamic.h:135)                                    */
  float   kappa;                  /* 0.01 (:134)                                                 */
  float   newborn_weight;         /* setNewBornParticleWeight(0.0001) risk_voxel.cpp:49          */
  float   obstacle_thickness;     /* obstacle_thickness_for_occlusion 0.3 (map_parameters.h:54)  */
} SogmDspParams;

typedef struct sogm_dsp sogm_dsp;

/*
 * DSPMap::DSPMap + setInitParameters (dsp_dynamic.h:118-163,566-632) for every agent of `map`.
 * The reference fills three 1e7-entry Gaussian tables from a time(0)-seeded engine (:1229-1239)
 * and draws uniforms from rand(); here the tables are inputs so that runs are reproducible:
 * host p_gauss[n_gauss] (position noise, N(0, 0.05)), v_gauss[n_gauss] (velocity noise),
 * rand_tab[n_rand] (values of rand(), 0..RAND_MAX).  The tables are read cyclically.
 * max_points: capacity of one agent's cloud (5000 in MapBase::filterPointCloud, map.cpp:126).
 * Particle store per agent: V x 16 slots x 25 B (SoA) — SOGM_ERR_HIP if it does not fit in HBM.  max_points <= 8192.
 */
int  sogm_dsp_create(sogm_ctx *map, const SogmDspParams *params, const float *p_gauss,
                     const float *v_gauss, int n_gauss, const int32_t *rand_tab, int n_rand,
                     int max_points, sogm_dsp **out);
void sogm_dsp_destroy(sogm_dsp *d);

/*
 * DSPMap::update (dsp_dynamic.h:165-364) for every agent: observation binning into FOV pyramids,
 * mapPrediction (:663), mapUpdate (:750), mapAddNewBornParticlesByObservation (:852),
 * mapOccupancyCalculationAndResample (:993).  Stream-ordered, no host round trip.
 * dev points  [n_total*3] fp32  sensor-frame points, already voxel-filtered (MapBase::filterPointCloud)
 * dev labels  [n_total*4] fp32  {vx, vy, vz, intensity} per point = an externally supplied output of
 *             velocityEstimationThread (new-born particles are then created in the order the points are given),
 *             or NULL: velocityEstimationThread (:1487-1678) runs on the GPU — ground split, Euclidean
 *             clustering (tolerance 2 x 0.15 m, 5..10000 points; PCL's seed / size order), centres, gated
 *             optimal assignment to the previous frame's clusters (Munkres' role), velocity = centre
 *             displacement / dt, and the new-born list in the reference's order [possibly-dynamic clusters]
 *             [ground points][static clusters].  Needs a voxel-filtered cloud (<= 128 neighbours within the
 *             tolerance, <= 256 clusters, <= 64 possibly-dynamic ones; beyond: an error counter, never a hang)
 * dev cloud_range [n_agents*2] int32 {begin,end} points of each agent
 * dev sensor_pos [n_agents*3] fp32, dev sensor_quat [n_agents*4] fp32 (w,x,y,z), dev stamps [n_agents] fp64
 * dev out_ok  [n_agents] int32: DSPMap::update's return value (0 = rejected odometry, :186-203)
 */
int sogm_update_dsp(sogm_dsp *d, const float *points, const float *labels,
                    const int32_t *cloud_range, const float *sensor_pos, const float *sensor_quat,
                    const double *stamps, int32_t *out_ok, void *stream);

/*
 * The map half of RiskVoxel::publishMap (risk_voxel.cpp:138-153): getOccupancyMapWithFutureStatus
 * (dsp_dynamic.h:445-469) into the SOGM grid of the map context (which also adopts the sensor
 * position / stamp of the last update as map pose / map time), clearing the future accumulators,
 * then the zeroing loop over the inflate kernel.  dev out_n_occupied [n_agents] int32 or NULL.
 * Follow with sogm_project_neighbours() for the overlay (risk_voxel.cpp:161-163).
 */
int sogm_dsp_publish(sogm_dsp *d, int32_t *out_n_occupied, void *stream);


/* ------------------------------------------------------------------------------------------ */
/* depth-image front end:  GridMap  (plan_env/src/grid_map.cpp, plan_env/src/raycast.cpp)      */
/*   SURVEY section 8 row f1 — not on the reference's SOGM path today; occupancy grid of its own */
/* ------------------------------------------------------------------------------------------ */
/* grid_map/ parameters (GridMap::initMap, grid_map.cpp:15-63).  The reference ships no YAML for them
 * (defaults are -1): every value is the caller's. */
typedef struct SogmGridMapParams {
  double  resolution;
  double  map_size[3];
  double  local_update_range[3];
  double  obstacles_inflation;
  double  fx, fy, cx, cy;
  double  depth_filter_maxdist, depth_filter_mindist;
  double  k_depth_scaling_factor;
  double  p_hit, p_miss, p_min, p_max, p_occ; /* 0.70 0.35 0.12 0.97 0.80 (:43-47) */
  double  max_ray_length;
  double  virtual_ceil_height, ground_height;
  int32_t use_depth_filter;    /* default true (:35) */
  int32_t depth_filter_margin;
  int32_t skip_pixel;
  int32_t local_map_margin;    /* 1 (:60) */
  int32_t rows, cols;          /* depth image size (480 x 640) */
} SogmGridMapParams;

typedef struct sogm_gridmap sogm_gridmap;

/* GridMap::initMap for n_agents independent maps (log-odds buffer fp64 like the reference). */
int  sogm_gridmap_create(const SogmGridMapParams *params, int n_agents, int device, sogm_gridmap **out);
void sogm_gridmap_destroy(sogm_gridmap *g);
/*
 * depthPoseCallback + updateOccupancyCallback (grid_map.cpp:585-665) for every agent:
 * projectDepthImage (:210-311), raycastProcess (:313-445: 3-D DDA per pixel from the ray end towards the
 * camera with the per-frame ray-end / traversed-voxel de-duplication, hit/miss log-odds fusion),
 * clearAndInflateLocalMap (:469-583).  The de-duplication makes the reference order dependent (a ray
 * stops at the first voxel an EARLIER ray traversed); it is reproduced exactly by iterating
 * "first ray to arrive" to its fixed point.
 * dev depth   [n_agents*rows*cols] uint16 (depth * k_depth_scaling_factor)
 * dev cam_pos [n_agents*3] fp64, dev cam_rot [n_agents*9] fp64 row-major camera-to-world rotation
 * dev out_updated [n_agents] int32 or NULL: 0 when the camera is outside the map (:656-662)
 */
int sogm_gridmap_update(sogm_gridmap *g, const uint16_t *depth, const double *cam_pos,
                        const double *cam_rot, int32_t *out_updated, void *stream);
/* GridMap::getInflateOccupancy (grid_map.h:342-349): dev agent_idx[n], pos[n*3] fp64 -> out[n] int8 {-1,0,1} */
int sogm_gridmap_query_inflate(sogm_gridmap *g, const int32_t *agent_idx, const double *pos, int n,
                               int8_t *out, void *stream);

/*
 * Tick glue of a batched planner service, each ONE launch (the FSM bookkeeping around replan()):
 * sogm_tick_inputs — the replan start state of every agent: its executing trajectory own_records[a] sampled at
 * stamp + replan_start_offset (FiniteStateMachine, plan_manager/src/plan_manager.cpp:169-175); an agent without a
 * trajectory (n_pieces == 0) starts from hover_inout[a] (odom, :127-133).  hover_inout [n][9] is refreshed to
 * {start position, 0, 0}.  Outputs (dev): out_now [n] = stamp (map stamp and "now" of the deconfliction),
 * out_t_start [n] = stamp + offset, out_pva [n][9], out_poses [n][3] fp32 (the map centres).
 * sogm_merge_latest — latest-wins per drone (traj_coordinator/src/particles.cpp:179-190): own_inout[a] =
 * new_records[a] where ok[a] != 0, else unchanged (a failed replan keeps executing the previous trajectory,
 * plan_manager.cpp:176-196); all_or_null, if given, receives a copy of the merged table (the swarm table of a
 * single-process run; multi-GPU runs use sogm_traj_allgather instead).
 */
int sogm_tick_inputs(const SogmTrajRecord *own_records, int n, double stamp, double replan_start_offset,
                     double *hover_inout, double *out_now, double *out_t_start, double *out_pva, float *out_poses,
                     void *stream);
int sogm_merge_latest(const SogmTrajRecord *new_records, const int32_t *ok, SogmTrajRecord *own_inout,
                      SogmTrajRecord *all_or_null, int n, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* queries                                                                                     */
/* ------------------------------------------------------------------------------------------ */
/*
 * getClearOcccupancy(pos, double dt)  — fake_particle_risk_voxel.cpp:309-346 / risk_base.cpp:228-260.
 * dev agent_idx[n_q] int32, dev pos_xyz[n_q*3] fp64 (world), dev t[n_q] fp64 (seconds after the map
 * stamp; if t_is_index != 0 the value is an integer slice index, the (pos,int) overload).
 * dev out[n_q] int8: 0 free, 1 occupied, -1 out of bound.
 */
int sogm_query_clear(sogm_ctx *ctx, const int32_t *agent_idx, const double *pos_xyz,
                     const double *t, int t_is_index, int n_q, int8_t *out, void *stream);

/*
 * BaselinePlanner::isTrajSafe(T) (plan_manager/src/baseline.cpp:45-68) for every agent: the executed
 * trajectory records[a] is sampled every 0.1 s from t_now[a] to min(T, duration) and checked with
 * getClearOcccupancy(pos, t + traj_start - map_time).  The trajectory's own time_start is used where the
 * reference reads the planner's traj_start_time_ member (identical after a successful replan).
 * dev records [n_agents], dev t_now [n_agents] fp64, dev out_safe [n_agents] int32 (1 safe).
 */
int sogm_traj_safe(sogm_ctx *ctx, const SogmTrajRecord *records, const double *t_now,
                   double check_duration, int32_t *out_safe, void *stream);

/*
 * getObstaclePoints(points, t_start, t_end, lc, hc) — map.cpp:480-518 (fake map) /
 * risk_base.cpp:295-337 (RiskBase, decayed threshold).  One box per entry, points appended in the
 * reference's z,y,x,slice order.
 * dev agent_idx[n_b], box_lo[n_b*3], box_hi[n_b*3], t0[n_b], t1[n_b] (absolute seconds);
 * dev out_pts[n_b*cap*3] fp64, dev out_counts[n_b] int32 (true count; > cap means truncated).
 */
int sogm_obstacle_points(sogm_ctx *ctx, const int32_t *agent_idx, const double *box_lo,
                         const double *box_hi, const double *t0, const double *t1, int n_b,
                         double *out_pts, int32_t *out_counts, int cap, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* planner context: search + corridors + QP, batched over the same agents as the map            */
/* ------------------------------------------------------------------------------------------ */
typedef struct sogm_planner sogm_planner;

/* FakeBaselinePlanner::init / BaselinePlanner::init (baseline_fake.cpp:18-51, baseline.cpp:15-43).
 * Lifetime: a planner refers to its map context until it is destroyed — destroy planners BEFORE their sogm_ctx
 * (sogm_planner_destroy unregisters the gate / progress words the map's kernels poll and waits for them on the map's
 * device; a C++ host declares the map member before the planner member, as host/sogm_reference_api.hpp does). */
int  sogm_planner_create(sogm_ctx *map, const SogmAstarParams *astar, const SogmPlannerParams *pp,
                         const SogmQpSettings *qp, sogm_planner **out);
void sogm_planner_destroy(sogm_planner *p);

/*
 * FakeRiskHybridAstar::reset + search (fake_risk_hybrid_a_star.cpp:84-426), then
 * getPathWithVel(corridor_tau) (:663-694), for every agent.  The `init=true` search is retried with
 * `init=false` when it returns NO_PATH, as baseline_fake.cpp:284-291 does.
 * dev start_pva [n_agents*9] fp64 rows pos,vel,acc;  dev goal [n_agents*3] fp64;
 * dev t_start   [n_agents]   fp64 absolute trajectory start time (traj_start_time_)
 * dev out_ret   [n_agents]   int32 ASTAR_RET (dyn_a_star.h:15)
 * dev out_route [n_agents*route_cap*6] fp64, dev out_route_len [n_agents] int32
 * dev out_stats [n_agents*4] int32 {use_node_num, iter_num, n_path_nodes, searches_run}
 * dev out_trace [n_agents*trace_cap] int32 or NULL: pool id of every popped node, in order.
 */
int sogm_astar_search(sogm_planner *p, const double *start_pva, const double *goal,
                      const double *t_start, int32_t *out_ret, double *out_route,
                      int32_t *out_route_len, int route_cap, int32_t *out_stats,
                      int32_t *out_trace, int trace_cap, void *stream);

/*
 * Corridor generation for given routes: baseline_fake.cpp:300-412 — per segment local box,
 * getObstaclePoints, firi::firi (sfc_gen/firi.hpp:238-365), ShrinkCorridor, validity LP;
 * adjacent-intersection LPs; goal reachability.  Polytopes are rows h0 x + h1 y + h2 z + h3 <= 0.
 * dev out_polys  [n_agents*SOGM_MAX_PIECES*max_faces*4] fp64
 * dev out_nfaces [n_agents*SOGM_MAX_PIECES] int32,  dev out_npoly [n_agents] int32
 * dev out_goal   [n_agents*6] fp64 local goal pos,vel
 */
int sogm_corridor_generate(sogm_planner *p, const double *start_pva, const double *t_start,
                           const double *route, const int32_t *route_len, int route_cap,
                           double *out_polys, int32_t *out_nfaces, int32_t *out_npoly,
                           double *out_goal, void *stream);

/*
 * BezierOpt::setup + optimize (traj_opt/src/bezier_optimizer.cpp:27-285) for every agent.
 * dev out_cpts [n_agents*SOGM_MAX_PIECES*15] fp64, dev out_status [n_agents] int32 (OSQP status_val:
 * 1 solved, -2 max iter, -3 primal infeasible ...), dev out_iters [n_agents] int32.
 */
int sogm_bezier_qp_solve(sogm_planner *p, const double *start_pva, const double *goal_pv,
                         const double *polys, const int32_t *nfaces, const int32_t *npoly,
                         double *out_cpts, int32_t *out_status, int32_t *out_iters, void *stream);
/*
 * BezierOpt::setup in full (traj_opt/src/bezier_optimizer.cpp:27-54,113-260) + optimize: an arbitrary time
 * allocation t_[i] per piece (continuity rows scaled by 1 / t, 1 / t^2 of either side of a knot :164-165,191-192;
 * velocity / acceleration boxes by t, t^2 :220-246), an end state with velocity AND acceleration (:150-160,205-215)
 * and the caller's limits — replan() itself only ever asks for t_[i] = corridor_tau and a final acceleration of
 * zero (baseline.cpp:411,423), which is what sogm_bezier_qp_solve / sogm_replan assemble.
 * dev end_pva    [n_agents*9]  fp64 rows pos, vel, acc of the end state
 * dev time_alloc [n_agents*SOGM_MAX_PIECES] fp64, entries [0, npoly) used
 * Everything else as sogm_bezier_qp_solve.  (The reference writes pow(t, 2); the kernel multiplies.)
 */
int sogm_bezier_qp_solve_timed(sogm_planner *p, const double *start_pva, const double *end_pva,
                               const double *time_alloc, double max_vel, double max_acc, const double *polys,
                               const int32_t *nfaces, const int32_t *npoly, double *out_cpts,
                               int32_t *out_status, int32_t *out_iters, void *stream);

/*
 * sdlp::linprog<d> (traj_utils/include/traj_utils/sdlp.hpp:709-787) for a batch of independent problems
 *   min c^T x  s.t.  A x <= b,  d = 3 or 4
 * — the LP behind checkCorridorValidity / checkCorridorIntersect / checkGoalReachability
 * (plan_manager/src/baseline_fake.cpp:143-199) and the MVIE's deepest interior point
 * (plan_manager/include/sfc_gen/firi.hpp:150-165), one wave per problem.  Same arithmetic as sdlp's projective
 * Seidel LP; the plane insertion order is a fixed permutation of the row count (sdlp draws it from a
 * process-global std::mt19937_64, :686-705).
 * dev c [n*d], dev A [total_rows*d] row-major, dev b [total_rows], dev row_range [n*2] int32 {begin,end};
 * dev out_x [n*d], dev out_min [n]: the minimum, +inf infeasible, -inf unbounded (out_x then holds a
 * direction, :777-781); NaN when a problem has more than 152 rows (the LDS capacity of one wave's LP).
 */
int sogm_linprog_batched(int d, const double *c, const double *A, const double *b,
                         const int32_t *row_range, int n, double *out_x, double *out_min, void *stream);

/*
 * firi::firi(bd, pc, a, b, hPoly, r, iterations, epsilon) (plan_manager/include/sfc_gen/firi.hpp:238-365) as a
 * standalone call for n independent problems — the same wave-per-problem code the replan runs per path segment,
 * without its box construction, ShrinkCorridor and validity LP.  No context: the current device, scratch is
 * allocated stream-ordered.
 * dev bd [n][n_bd][4] row-major (n_bd <= 32; rows h.x + h3 <= 0), dev pc_xyz packed fp64 points with
 * dev pc_range [n][2] = {first, last+1} into it (at most max_points <= 16384 points per problem),
 * dev a, b [n][3] (the seed segment), dev r [n][3] in/out (ellipsoid radii: the reference's callers pass ones),
 * iterations >= 1 (2 at baseline.cpp:352), epsilon 1e-6 in the reference.
 * dev out_hpoly [n][max_faces][4] (max_faces <= 128), out_nfaces [n], out_status [n]: 1 = firi returned true,
 * 0 = it returned false (a or b outside bd; nothing else written), -3 = over capacity (more than max_points
 * points, or more than max_faces / 128 selected planes: the polytope is truncated).
 */
int sogm_firi_batched(const double *bd, int n_bd, const double *pc_xyz, const int32_t *pc_range, const double *a,
                      const double *b, double *r, int iterations, double epsilon, int n, int max_points,
                      int max_faces, double *out_hpoly, int32_t *out_nfaces, int32_t *out_status, void *stream);

/*
 * Per-object use of the per-stage entries (host/sogm_reference_api.hpp): after select_agents(first, count),
 * sogm_astar_search / sogm_corridor_generate / sogm_bezier_qp_solve / sogm_safe_after_opt process agents
 * [first, first + count) only; every array keeps its full [n_agents] layout and is indexed by the absolute agent.
 * (0, n_agents) restores the default.  sogm_replan always processes every agent.
 * set_search_mode: 0 (default) = the replan's call pattern — search(…, init_search = true, …) and, if that returns
 * NO_PATH, reset() + search(…, false, …) (baseline_fake.cpp:284-291); 1 / 2 = exactly one
 * RiskHybridAstar::search with init_search = true / false (risk_hybrid_a_star.h:103-110).  Adding 4 makes
 * sogm_astar_search read t_start[] as that function's time_start argument (seconds after the map stamp) instead of
 * an absolute time.  Adding 16 selects search(…, dynamic = false, …): the reference's branch reads node times it never
 * writes (risk_hybrid_a_star.cpp:153-158,177,271,324); they are DEFINED as zero here (oracle/astar_oracle.cpp does the
 * same): a spatial search over the SOGM's first tau seconds, t_start ignored.
 */
int sogm_planner_select_agents(sogm_planner *p, int first, int count);

int sogm_planner_set_search_mode(sogm_planner *p, int mode);

/*
 * ParticleATC::isSafeAfterOpt (traj_coordinator/src/particles.cpp:223-283) for every agent: the new
 * trajectory's control points must be linearly separable (separator::Separator::solveModel,
 * utils/separator/src/separator_glpk.cpp:75-190 — a GLPK feasibility LP, solved here with the
 * planner's own LP) from the remaining control points of every other agent's active trajectory.
 * dev cpts [n_agents*SOGM_MAX_PIECES*15] fp64 (sogm_bezier_qp_solve layout), dev npoly [n_agents],
 * dev records [n_records], dev ego_ids [n_agents], dev t_now [n_agents] fp64 ("ros::Time::now()" of the
 * check), dev out_safe [n_agents] int32 (1 safe, 0 collides / LP capacity exceeded; agents with
 * npoly <= 0 report 1).
 */
int sogm_safe_after_opt(sogm_planner *p, const double *cpts, const int32_t *npoly,
                        const SogmTrajRecord *records, int n_records, const int32_t *ego_ids,
                        const double *t_now, int32_t *out_safe, void *stream);
/*
 * Makes sogm_replan() finish like FakeBaselinePlanner::replan does (baseline_fake.cpp:453-460): with
 * a swarm set, a trajectory that fails isSafeAfterOpt against `records` counts as a failed replan.
 * The device pointers are read by later sogm_replan() calls; records == NULL switches the check off.
 */
int sogm_planner_set_swarm(sogm_planner *p, const SogmTrajRecord *records, int n_records,
                           const int32_t *ego_ids, const double *t_now);

/*
 * Cumulative counters of sogm_replan() since creation / the last reset: where each replan ended (the early
 * returns of baseline_fake.cpp:292,405-419,447,455) and how often a capacity limit that the reference does not
 * have (it grows std::vectors) turned a corridor / a deconfliction check into a failure.  Synchronous.
 * host out[SOGM_CNT_N] int64.
 */
enum {
  SOGM_CNT_REPLAN_OK           = 0, /* replan() returned true                                        */
  SOGM_CNT_FAIL_SEARCH         = 1, /* A* NO_PATH after the retry (:284-295)                         */
  SOGM_CNT_FAIL_CORRIDOR       = 2, /* no usable corridor (:405-419)                                 */
  SOGM_CNT_FAIL_QP             = 3, /* OSQP status not SOLVED / SOLVED_INACCURATE (:447)             */
  SOGM_CNT_FAIL_UNSAFE         = 4, /* isSafeAfterOpt false (:455-460)                               */
  SOGM_CNT_CORRIDOR_CAPACITY   = 5, /* corridor chains cut at a segment that exceeded pc_capacity points,
                                       128 selected planes or max_faces (treated as invalid)         */
  SOGM_CNT_PIECES_CAPACITY     = 6, /* routes longer than SOGM_MAX_PIECES segments (truncated)       */
  SOGM_CNT_DECONFLICT_CAPACITY = 7, /* pairs with more than 144 LP rows (treated as unsafe)          */
  SOGM_CNT_N                   = 8
};
/*
 * Publication inside the replan.  The reference's FSM publishes a successful replan's trajectory and keeps executing
 * the previous one otherwise (plan_manager.cpp:176-199,364-399); every other drone stores what it receives
 * (particles.cpp:131-191).  sogm_merge_latest does that in a launch of its own after sogm_replan; with a table
 * registered here the replan's finishing kernel does it per agent as its chain completes (same stores, overlapped with
 * the other agents' chains — beside the streaming clear a separate store-heavy launch takes 2 ms):
 *   own_records [n_agents]  dev: overwritten with the new record where the replan succeeded (latest wins);
 *   next_table  [n_agents]  dev or NULL: receives every agent's CURRENT record (new, or the one it keeps executing) —
 *                           the swarm table of the NEXT tick for a single-process host (n_total == n_agents), which
 *                           must differ from the table registered with sogm_planner_set_swarm for this replan
 *                           (sogm_replan returns SOGM_ERR_INVALID_ARG when the two are the same table, or when
 *                           own_records overlaps its out_records: the finishing kernel would write what other agents'
 *                           deconfliction reads in the same launch).
 * NULL, NULL switches it off (the default).  Pointers are read by later sogm_replan calls.
 */
int sogm_planner_set_publish(sogm_planner *p, SogmTrajRecord *own_records, SogmTrajRecord *next_table);

/* Pre-stamp.  The tick's critical path is "map update -> chain of the slowest agent", yet an agent's next map centre
 * depends on its own new record only, which is final when its replan finishes.  With a SogmPrestamp registered (and
 * publication on, two or three grids per agent, the sparse reset, the dataflow replan), sogm_replan also builds the NEXT
 * tick's map —
 * sogm_tick_inputs' start states, then FakeParticleRiskVoxel::updateMap's stamp (fake_particle_risk_voxel.cpp:80-170)
 * — agent by agent as their records are published, into the pool's next grid; the next tick then calls
 * sogm_update_prestamped (the grid swap + the neighbour overlay) instead of sogm_tick_inputs + sogm_update_gt_swarm.
 * Same cells, same start states.  ps = NULL switches it off; the arrays stay owned by the caller and must not be the
 * ones the replan in flight reads (start_pva, t_start: double-buffer them).  A replan that could not pre-stamp (no
 * spare grid ready yet: first ticks) leaves sogm_update_prestamped returning SOGM_ERR_STATE: fall back to the two
 * calls.
 * Stream order: a pre-stamping sogm_replan leaves `stream` behind its own outputs (out_records, out_ok), NOT behind the
 * pre-stamp, which goes on for a few hundred microseconds on a stream of the context.  The SogmPrestamp outputs and the
 * pre-stamped grid are complete in stream order behind the next sogm_update_prestamped / sogm_update_* / sogm_replan
 * call on a stream (each joins it), or after a device synchronisation — sogm_update_prestamped launches its overlay
 * under that tail, an agent's additions waiting for that agent's stamp.  Tuning key "splat_overlap" = 0 restores the
 * join inside sogm_replan.
 * Failed ticks: if the pre-stamping replan failed on the device (sogm_planner_flow_failures), its grid is incomplete;
 * sogm_update_prestamped then returns SOGM_ERR_STATE once the failure has reached the host — rebuild the map with
 * sogm_tick_inputs + sogm_update_gt_swarm, which discards the grid and clears the stamp's bitmask. */
typedef struct SogmPrestamp {
  const float        *cloud_xyz;    /* next update's inputs, as for sogm_update_gt (device) */
  const int32_t      *cloud_range;
  const SogmCylinder *cylinders;
  int32_t             n_cyl;
  int32_t             reserved_;
  double              next_stamp;           /* sogm_tick_inputs' stamp of the next tick */
  double              replan_start_offset;
  double             *hover_inout;          /* sogm_tick_inputs' in/out and outputs for the next tick (device) */
  double             *out_now;
  double             *out_t_start;
  double             *out_pva;
  float              *out_poses;            /* the next map centres, [A][3] (may be NULL: the context keeps its own) */
  const SogmWorld    *world;                /* host pointer or NULL.  Non-NULL: the stamp's inputs are this frame, cropped on the
                                               device around each agent's NEXT map centre, and cloud_xyz / cloud_range /
                                               cylinders above are ignored (may be NULL).  CAUSALITY: a pre-stamp runs inside
                                               tick k, so the newest frame a live host can hand it is tick k's — the map tick
                                               k + 1 plans on is then one tick staler than the reference's, which updates from
                                               the cloud that arrived for that update (map.cpp:170-171).  bench.py reports this
                                               as config.map_input_staleness_ticks = 1 and keeps the 0-staleness tick
                                               (sogm_update_world at the start of the tick) as the headline. */
} SogmPrestamp;
int sogm_planner_set_prestamp(sogm_planner *p, const SogmPrestamp *ps);
/* 1 if the last sogm_replan pre-stamped the next grid (no synchronisation: host-side state). */
int sogm_prestamp_pending(const sogm_ctx *ctx);
/* `stream` waits for the end of the last replan's pre-stamp, if nothing has joined it yet (see "Stream order" above):
 * for a host that touches the SogmPrestamp arrays itself — e.g. sogm_tick_inputs on the same hover_inout — before its
 * next sogm_update_* / sogm_replan call.  No-op otherwise. */
int sogm_prestamp_join(sogm_ctx *ctx, void *stream);
/* The update of a pre-stamped tick: adopts the grid, its map centres and stamps, then adds the neighbour overlay
 * (records may be NULL with n_records = 0).  A grid whose pre-stamping replan FAILED is refused (SOGM_ERR_STATE: build the
 * map with sogm_update_gt* / sogm_update_world instead) — decided on two levels: while the pre-stamp is still running the
 * overlay waits per agent on the device and writes nothing once that replan's error word is set (the tick is then
 * reported as failed by sogm_planner_flow_failures); the HOST check (the failure count against its value when the pre-stamp
 * was queued) sees a failure only once the failing replan's report has run, i.e. it is exact after a synchronisation of
 * the stream and best-effort for a host that pipelines ticks without one. */
int sogm_update_prestamped(sogm_ctx *ctx, const SogmTrajRecord *records, int n_records, const int32_t *ego_ids,
                           void *stream);
/*
 * Flight: n_ticks replan ticks of every agent of the batch in ONE call, every agent on its own clock — the reference's
 * drones each run their own FSM and read whatever trajectories arrived last (plan_manager/src/plan_manager.cpp:92-233,
 * traj_coordinator/src/particles.cpp:179-190); a lock-step sogm_replan per tick makes 127 agents wait for the slowest
 * chain of every tick.  RESULTS are fixed by a staleness rule, so a flight is reproducible whatever the schedule:
 *   agent a's tick k  =  FakeParticleRiskVoxel::updateMap from worlds[k] around a's start state of tick k (sampled from
 *   its own record of tick k - 1 at t0 + k * period + replan_start_offset), overlay and isSafeAfterOpt against table
 *   ver(k - 2) — every agent's executed record as of ITS tick k - 2 —, then FakeBaselinePlanner::replan; a may start tick
 *   k as soon as its own tick k - 1 is finished and EVERY agent has finished tick k - 2.
 * (sogm_replan's lock-step tick reads ver(k - 1): the flight's neighbour records are one tick staler, like a record that
 * missed one broadcast period.)  tables is a ring of four versions, ver(j) at tables[(j & 3) * n_total]: on entry ver(first_tick - 1)
 * and ver(first_tick - 2) must hold the swarm's records of those ticks (a fresh flight: the initial table — empty or hover
 * records — in both); on return ver(first_tick + n_ticks - 1) and ver(first_tick + n_ticks - 2) are complete, i.e. a
 * following call with first_tick advanced by n_ticks continues the flight.  Needs the sparse reset (the agent's single grid is
 * reset through its mark log at the start of each of its ticks), body particles, 32-byte aligned agent grids.
 * Several ranks (n_total > n_agents: the batch is rows agent0 .. agent0 + n_agents - 1, the other rows belong to other
 * ranks).  With `nccl_comm` set (an ncclComm_t, e.g. sogm_comm_handle) the exchange runs BEHIND the call, no host step
 * between ticks: for every tick j of the call the library queues, on the context's exchange stream, a one-lane kernel that
 * waits until every LOCAL agent has finished tick j, the in-place ncclAllGather of this rank's rows of ver(j)
 * (tables[(j & 3) * n_total + agent0], n_agents records), and a one-lane kernel that marks ver(j) complete and queues the
 * overlays of tick j + 2 parked at the gate; the gate of tick k's overlay is then "the all-gather of ver(k - 2) has completed
 * here" — the same staleness rule, so the records equal those of one process flying all agents.  Every rank passes the same
 * n_ticks (the collectives are matched one to one); a rank whose flight failed still runs all of its collectives.  The
 * call's last two versions are complete when the exchange stream has drained: `stream` waits for it as for the flight.
 * Without `nccl_comm` a call flies at most TWO ticks — ticks k and k + 1 read ver(k - 2) and ver(k - 1), which the host
 * completes between two calls by all-gathering every rank's rows of the versions the previous call finished
 * (sogm_traj_allgather, in place).  Both forms: tests/test_exchange_gpu.py (two ranks on one GPU, stand-in RCCL).  Four persistent kernels on
 * four streams with disjoint compute-unit masks (tuning keys flight_*_units, 16 CUs per unit); asynchronous: `stream`
 * waits for the flight's end.  SOGM_ERR_STATE if the planner was created without the dataflow path.
 */
typedef struct SogmFlight {
  int32_t          n_ticks;              /* 1 .. 64 */
  int32_t          first_tick;           /* absolute index of this call's first tick (>= 0) */
  double           t0, period;           /* stamp of tick k = t0 + k * period */
  double           replan_start_offset;  /* fsm/replan_start_time */
  const SogmWorld *worlds;               /* host array [n_ticks]: the sensor frame of tick first_tick + i */
  const double    *goals;                /* dev [A][3] */
  const int32_t   *drone_ids;            /* dev [A]  (ego ids of the batch = rows agent0 .. agent0 + A - 1 of the tables) */
  double          *hover_inout;          /* dev [A][9] where an agent without a trajectory hovers (sogm_tick_inputs) */
  SogmTrajRecord  *own_inout;            /* dev [A] the records the agents execute (latest wins) */
  SogmTrajRecord  *tables;               /* dev [4][n_total] */
  int32_t          n_total, agent0;
  SogmTrajRecord  *log_records;          /* dev [n_ticks][A] every tick's sogm_replan-style output record ... */
  int32_t         *log_ok;               /* dev [n_ticks][A] ... and ok flag */
  void            *nccl_comm;            /* several ranks: the communicator of the exchange behind the call (see above); NULL:
                                            one process owns every row, or the host all-gathers between calls of two ticks */
} SogmFlight;
int sogm_flight_run(sogm_planner *p, const SogmFlight *flight, void *stream);
/* Optional: creates the flight's control block and its five streams (four masked ones + the exchange stream), runs an empty
 * kernel on each, so that their hardware queues exist before any flight is in the air, and allocates what the first
 * sogm_flight_run would allocate (crop lists for frames of up to max_cloud_points points, the stamp's scratch) — the first
 * call otherwise synchronises the DEVICE while it grows them.  For a host that flies several planners in ONE process (a
 * second planner's first call would wait for the first planner's flight to end).  Reads the flight_* tuning keys like the
 * first sogm_flight_run would.  Synchronises. */
int sogm_flight_prepare(sogm_planner *p, int max_cloud_points);
/* After a flight (synchronises): host out_ms[A][8] = per-agent sums over the last flight in ms {wait at the tick k - 2 gate,
 * map (reset + stamp + overlay), search (queue + A*), corridors (queue + FIRI), QP (queue + solve), finish, whole chain,
 * ticks completed}; host out_hdr[32] = the flight's control counters ([4] = error code, 0 = none; [5] = agent-ticks
 * finished; [15] = workgroups of the flight's four kernels that were NOT running within 1 ms of their kernel's first workgroup — must be
 * 0: every kernel's compute-unit mask holds the same number of units in every shader engine it touches and a launch has
 * exactly the workgroups that mask holds at once, so nothing waits in a dispatcher that a queue save / restore could start
 * ahead of a restored wave (DESIGN.md 3.1 "Liveness of the flight"); [16..24] = wave time of the map / corridor + finish kernels by activity, in units of 10 us: map workers idle,
 * reset, bits, marks, overlay, heads, light waves idle, corridor segments, finish; [25..31] = descriptor counts). */
int sogm_flight_stats(sogm_planner *p, double *out_ms_host, int32_t *out_hdr_host);
int sogm_planner_counters(sogm_planner *p, int64_t *out_host, int reset);
/* sogm_replan() chains its kernels per agent through device-side ready lists ("dataflow replan");
 * a wait that exceeds 3 s marks the tick as failed instead of hanging the GPU: agents whose chain did not complete
 * report ok = 0 and an empty record (n_pieces = 0) for that tick.  Synchronises the device and returns 0 if the
 * last sogm_replan() completed normally, a positive code if one of its waits timed out, negative = sogm_status. */
int sogm_planner_flow_error(sogm_planner *p);
/* The same without synchronising anything: out[0] = code of the most recent failed tick (0 = none ever),
 * out[1] = number of sogm_replan() calls that failed so far, as of the ticks that have COMPLETED on the device
 * (the words live in pinned host memory and are written by the last kernel of each replan).  A tick driver polls
 * this once per tick and stops merging records when the count moves (the reference has no analogue: its replan()
 * cannot time out). */
int sogm_planner_flow_failures(sogm_planner *p, int32_t out[2]);

/*
 * One full FakeBaselinePlanner::replan (baseline_fake.cpp:266-472; isSafeAfterOpt only when a swarm has
 * been set with sogm_planner_set_swarm) for every
 * agent: search -> corridors -> QP, stream-ordered, no host round trip.  On success writes the
 * agent's SogmTrajRecord (time_start = t_start) into out_records; on failure writes n_pieces = 0.
 * dev out_ok [n_agents] int32 (1 = replan() returned true).
 */
int sogm_replan(sogm_planner *p, const double *start_pva, const double *goal,
                const double *t_start, const int32_t *drone_ids, SogmTrajRecord *out_records,
                int32_t *out_ok, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* trajectory exchange: the /broadcast_traj topic as ONE RCCL all-gather per replan tick          */
/*   (plan_manager/src/plan_manager.cpp:364-399 publish, traj_coordinator/src/particles.cpp:131-191 receive)  */
/* ------------------------------------------------------------------------------------------ */
/*
 * Agents are sharded over the GPUs of a node, one process per GPU; rank r owns a contiguous block of agents.
 * After sogm_replan() (and the caller's latest-wins merge) every rank contributes the SogmTrajRecord of each of
 * its n_local agents and receives the whole swarm's: all_records[r * n_local + i] = rank r's local_records[i].
 * `nccl_comm` is an ncclComm_t (RCCL; <rccl/rccl.h>) whose ranks all call this with the same n_local.  RCCL is
 * resolved with dlopen("librccl.so.1") the first time it is needed, so a host that already links RCCL (or runs
 * PyTorch-ROCm) shares its instance and may pass a communicator of its own; sogm_comm_* creates one for hosts
 * that do not have RCCL headers.
 * The collective runs on an internal exchange stream: it starts when the work queued on `stream` so far is done
 * and sogm_project_neighbours / sogm_replan (deconfliction) / sogm_safe_after_opt wait for it on their own stream,
 * so the next tick's clear, stamp and trajectory sampling overlap with it.  A host that reads all_records by other
 * means calls sogm_exchange_wait(ctx, its_stream) first.
 * dev local_records [n_local], dev all_records [n_local * n_ranks].
 */
int sogm_traj_allgather(sogm_ctx *ctx, void *nccl_comm, const SogmTrajRecord *local_records, int n_local,
                        SogmTrajRecord *all_records, void *stream);
int sogm_exchange_wait(sogm_ctx *ctx, void *stream);

/* Communicator helpers (thin wrappers of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy).  Rank 0 makes
 * an id and ships its 128 bytes to the other ranks by any out-of-band means (torch.distributed's store, MPI, a
 * file); every rank then calls sogm_comm_create.  sogm_comm_handle returns the ncclComm_t to pass above. */
#define SOGM_COMM_ID_BYTES 128
typedef struct sogm_comm sogm_comm;
int   sogm_comm_unique_id(char *out_id_host /* [SOGM_COMM_ID_BYTES] */);
int   sogm_comm_create(const char *id_host, int rank, int n_ranks, int device, sogm_comm **out);
void  sogm_comm_destroy(sogm_comm *comm);
void *sogm_comm_handle(sogm_comm *comm);
/* what RCCL itself says about the communicator: host out[2] = {ncclCommCount, ncclCommUserRank} (a multi-GPU bench line
 * reports them: a run that silently fell back to fewer ranks cannot print the requested N) */
int   sogm_comm_info(sogm_comm *comm, int32_t *out_host);

#ifdef __cplusplus
}
#endif
#endif /* SOGM_ABI_H */
