/* sogm_abi_debug.h — the part of libsogm_hip.so's C ABI that a plan_manager host does not need to fly: tuning knobs,
 * per-kernel profiling, device clocks, traffic counters, parity downloads of internal state and test hooks.  Used by
 * bench.py, tools/ and tests/ (the Python binding declares both headers' symbols; tests/test_abi_symbols.py checks both
 * against the library).  Same library, same SOGM_ABI_VERSION as include/sogm_abi.h, which it includes.
 * host/sogm_facade.hpp includes only sogm_abi.h; its setTuning() exists when this header was included first. */
#ifndef SOGM_ABI_DEBUG_H
#define SOGM_ABI_DEBUG_H
#include "sogm_abi.h"
#ifdef __cplusplus
extern "C" {
#endif

/* host out[8]: {enabled, log capacity per agent, 1 if the current grid is covered by its log, largest per-agent
 * entry count of the current grid's log (above the capacity: that agent's next reset is dense), entries of all
 * agents together (saturating), sparse resets launched since the previous call, mean entries read per such
 * launch, mean KiB zeroed per such launch (the stores the reset kernel issued, counted on the device)}.
 * Synchronises. */
int sogm_sparse_reset_state(sogm_ctx *ctx, int32_t *out_host);

/* Tuning knobs.  Every internal width and switch of the library has a default chosen on MI355X; a host that wants
 * another value sets it per context — nothing is read from the environment (the library's only environment switches are
 * SOGM_SPARSE_RESET, SOGM_FLOW and SOGM_RCCL_LIB).  Keys (defaults in parentheses):
 *   read by sogm_planner_create (set them before):  "groups" (2) agent groups of the grouped-stream replan,
 *     "spec_astar" (1) speculative second search, "clear_gate_frac" (1.0);
 *   read at every call:  "qp_wgs" (0 = half the CUs) persistent QP workgroups, "stamp_wgs" (256), "stamp_bits_wgs" (0 = stamp_wgs), "stamp_cached" (0), "stamp_lds_log" (1), "stamp_lds_kb" (0), "splat_wgs" (256),
 *     "splat_overlap" (1), "reset_wgs" (32), "reset_lanes" (0 = auto: 2 under the replan, 4 alone), "reset_unroll"
 *     (0 = auto: 1 / 8), "reset_late" (1), "prestamp_bits" (32), "prestamp_marks" (64), "prestamp_wgs" (0 = 8 per CU), "prestamp_stream" (1), "prestamp_gate_frac" (0.9: the pre-stamp of the next map starts when that share of the agents' corridors is final — an event recorded behind a gate kernel on the resets' stream; 1.0: when all are),
 *     "prestamp_late_agents" (8), "prestamp_late_bits" (128), "prestamp_late_marks" (256), and the dense clear's
 *     "clear_wgs" (0 = auto), "clear_throttle" (0), "clear_nt" (1), "clear_wide_wgs" (256; 0 = fixed width),
 *     "clear_wide_bound" (0), "clear_head_gb" (1e9), "clear_early" (0), "clear_retire_at_end" (0); "qp_ablate" (0;
 *     profiling builds only);
 *   sogm_flight_run (read at its first call: the masked streams are created once):  "flight_qp_units" (4),
 *     "flight_search_units" (2), "flight_map_units" (4) — compute units in units of 16 for the QP / search / map kernels,
 *     the corridor + finish kernel takes the rest —, "flight_masks" (1; 0 = unmasked streams), "flight_spec" (1: both
 *     search attempts side by side), and per (agent, tick) one-wave tickets "flight_reset" (8), "flight_bits" (16),
 *     "flight_marks" (32), "flight_splat" (4); "flight_admit" (48) agents whose map may be under construction at once, "flight_pace_us" (20; 40 until round 6) microseconds between two
 *     admissions to the map stage (agents then reach every stage at a steady rate; 0 = unpaced), "flight_heads" (32)
 *     admitting waves of the map kernel; "flight_urgent" (8): an agent among the last n finishers of a tick — the agents the
 *     swarm waits for at the next gate — builds the map of its next tick through a lane of its own (four of the heads, no
 *     admission order / pace / window; a work queue of its own that the first "flight_urgent_waves" (4096 = all) map workers
 *     look at before they take plain work and while they wait for it; its maps are cut into "flight_urgent_fine" (4) times
 *     more tickets; 0 = no such lane); "flight_gate_pace_us" (20; 40 until round 6): the staleness rule's gate — tick k reads the neighbours'
 *     records of tick k - 2 — sits in front of a map's overlay, the only phase that reads them: an agent builds the rest of its
 *     next map while it waits, and the finish that opens a gate queues the waiting overlays this many microseconds apart;
 *     "flight_neighbour_lag" (2): tick k's overlay and isSafeAfterOpt read the neighbours' records of tick k - lag.  2 is the
 *     flight's rule (an agent runs up to two ticks ahead of the slowest); 1 is the REFERENCE's staleness — a record is at most
 *     one broadcast period old (particles.cpp:179-190), what the lock-step tick reads — with less overlap: an agent's overlay
 *     waits until every agent has finished the previous tick.  This key DOES change records (it selects the rule);
 *     "flight_engines" (4), "flight_engine_first" (0): the shader engines (of every XCD) the flight's kernels share — a flight on
 *     half of them leaves the other half to a second flight on the same device (two ranks as two threads: tests) —,
 *     "flight_exchange_units" (0): units given to NO kernel, room for a collective's own kernels beside a multi-rank flight
 *     (the Python driver sets 1 when torch.distributed runs over more than one rank: untested on hardware);
 *     "flight_light_per_cu" (4), "flight_map_per_cu" (8): one-wave workgroups of the corridor + finish / the map kernel per
 *     compute unit of their partition (what a unit holds at once; fewer leaves slack).  Whatever the unit counts, every
 *     kernel's mask gets the same number of units in every shader engine it touches and a launch exactly the workgroups
 *     that mask holds at once (sogm_flight_stats hdr[15] counts workgroups that were not resident from the start: 0).
 *     None of these keys but flight_neighbour_lag changes a cell or a record.
 *   sogm_update_world, experimental (measured, not adopted: profiles/EXPERIMENTS.md round 5):  "update_flow" (0; 1 = the
 *     maps are built agent by agent on a stream of the context's own — one persistent launch over one-wave tickets, per
 *     agent occupancy bits -> marks -> overlay, agents in the order of their previous chain's length — and sogm_replan's
 *     searches start per agent as their map completes; every other reader of the grid joins the flow's end by itself;
 *     identical cells and records), with "update_bits" (16), "update_marks" (64), "update_splat" (40) tickets per agent,
 *     "update_wgs" (0 = 16 per CU), "update_chunk" (1) tickets per claim, "update_cached" (0), "update_order" (1).
 * Not thread-safe against calls on the same context (like every other call).  Unknown key: SOGM_ERR_INVALID_ARG.
 * sogm_tuning_key(i) enumerates the keys (NULL past the last). */
int         sogm_set_tuning(sogm_ctx *ctx, const char *key, double value);
int         sogm_get_tuning(const sogm_ctx *ctx, const char *key, double *out_value_host);
const char *sogm_tuning_key(int index);

/* What the map kernels moved since the last reset of these counters (device-side counts, for the roofline figures of
 * bench.py); host out[6]: {resets through the mark logs, log entries those resets read, bytes they zeroed, stamps
 * launched (sogm_update_gt* and pre-stamps), marks (cells set to 1) the stamps wrote, log entries the stamps
 * appended}.  Counted while the sparse reset is on.  reset != 0 zeroes the counters.  Synchronises. */
int sogm_map_traffic(sogm_ctx *ctx, int64_t *out_host, int reset);

/* How the CURRENT grid (the one queries and planning read) came to be; host out[4]: {pool slot, resets of that slot
 * through its mark log since the pool was built, dense clears of that slot (fake_particle_risk_voxel.cpp:107-108's
 * fill), 1 if the grid was built by the previous sogm_replan's pre-stamp and adopted by sogm_update_prestamped}.
 * Host-side launch counts; does not synchronise.  (Parity tests use it to assert which path built the map.) */
int sogm_grid_history(sogm_ctx *ctx, int32_t *out_host);

/* Per-kernel timing with HIP events recorded on the caller's stream around each launch (used by
 * bench.py for the roofline figure).  Slots: */
enum {
  SOGM_PROF_CLEAR = 0, /* the grid's reset: k_reset_sectors (sparse) or k_clear_slabs / k_clear_chunks (dense) */
  SOGM_PROF_STAMP = 1, /* k_cull_cylinders + k_stamp_bits + k_stamp_marks    */
  SOGM_PROF_SPLAT = 2, /* k_splat_neighbours                                 */
  SOGM_PROF_ASTAR = 3,
  SOGM_PROF_CORRIDOR = 4,
  SOGM_PROF_QP = 5,
  SOGM_PROF_CLEAR_HEAD = 6, /* single-grid pipelining: the narrow first part of a two-part clear (slot 0 = the rest) */
  SOGM_PROF_EXCHANGE = 7, /* sogm_traj_allgather: the ncclAllGather on the exchange stream */
  SOGM_PROF_N = 8
};
int sogm_set_profiling(sogm_ctx *ctx, int enable);
/* The same for a choice of slots (bit k of slot_mask = slot k; 0 = off): every timed launch costs two event records on
 * its stream, which a tick's critical path notices — bench.py times only the rated kernel inside its timed region. */
int sogm_set_profiling_slots(sogm_ctx *ctx, int slot_mask);
/* Synchronises the device, then writes the duration (ms) of the LAST launch of each slot
 * (negative if that slot has not run since profiling was enabled).  host out_ms[SOGM_PROF_N]. */
int sogm_profile_read(sogm_ctx *ctx, double *out_ms_host);
/* Every launch of a slot since sogm_set_profiling(ctx, 1) keeps its own event pair (the last 1024 are kept), so a
 * whole timed region can be measured launch by launch with no synchronisation inside it.  Synchronises the
 * device, then writes the durations (ms, oldest first) of the last min(cap, 1024, launches) launches of `slot`
 * to host out_ms[cap] and their number to *out_n. */
int sogm_profile_read_all(sogm_ctx *ctx, int slot, double *out_ms_host, int cap, int *out_n);

/* Where a tick's wall time goes (bench.py's sustained.slowest_tick).  sogm_device_clock runs a one-lane kernel on `stream`
 * that reads the device's 100 MHz wall clock, synchronises the stream and returns the value: a host that brackets the call with
 * its own clock learns the offset between the two clocks to within the synchronisation's return latency (also arms the
 * stamps below).  sogm_tick_clock: host out[2] = that clock at the start of the last map update's first kernel and in the
 * last sogm_replan's closing kernel (pinned memory, no synchronisation: read them after one). */
int sogm_device_clock(sogm_ctx *ctx, int64_t *out_ticks_host, void *stream);
int sogm_tick_clock(sogm_ctx *ctx, int64_t *out2_host);

/* Copy agent `a`'s grid to host in the reference layout risk_maps_[V][T] (map.h:52). Synchronous. */
int sogm_download_reference_layout(sogm_ctx *ctx, int agent, float *out_vt_host);

/* Parity I/O (synchronous): agent's particle store in the reference layout
 * voxels_with_particle[V][2*max][9] (slot 8, the update time, is written as 0) and
 * voxels_objects_number[V][4+T]; counters[16]: {voxel_full, pyramid_full, moved_out, candidates,
 * p_seq, v_seq, rand_seq, slots_per_voxel, slots_per_pyramid, n_pyramids, unconverged_rounds,
 * pool_overflow, ...}.  Any pointer may be NULL. */
int sogm_dsp_download_state(sogm_dsp *d, int agent, float *store_host, float *objnum_host,
                            int32_t *counters_host);
/* input_cloud_with_velocity of the last update (the new-born list, rows {x,y,z,vx,vy,vz,intensity}, at most `cap`
 * rows), its length, and counters4 = {clusters, possibly dynamic, matched, error code of the velocity estimation}. */
int sogm_dsp_download_born(sogm_dsp *d, int agent, float *born_host, int cap, int32_t *n_born, int32_t *counters4);
/* Observation tables of the last update: nobs[n_pyramids], pc[n_pyramids*obs_max*5], maxlen[n_pyramids] */
int sogm_dsp_download_observations(sogm_dsp *d, int agent, int32_t *nobs_host, float *pc_host,
                                   float *maxlen_host);

/* Parity I/O (synchronous): occupancy_buffer_ [nx*ny*nz] fp64, occupancy_buffer_inflate_ int8,
 * bounds[6] = local_bound_min, local_bound_max, counters[4] = {rays, active rays, rounds used, errors}. */
int sogm_gridmap_download(sogm_gridmap *g, int agent, double *occupancy_host, int8_t *inflate_host,
                          int32_t *bounds_host, int32_t *counters_host);
/* Test hook: sets raycast_num_ (the de-duplication flags are chars and stop matching after frame 127). */
int sogm_gridmap_force_frame(sogm_gridmap *g, int raycast_num);

#ifdef __cplusplus
}
#endif
#endif /* SOGM_ABI_DEBUG_H */
