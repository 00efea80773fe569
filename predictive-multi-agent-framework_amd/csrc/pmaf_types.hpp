// pmaf_types.hpp -- plain structs shared by the host side (pmaf_host.cpp, compiled by g++) and the kernel
// translation units (pmaf_k_*.hip, compiled by hipcc), and the launch interface between them. No device code here.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

// doubles in front of the path of a winner record (include/pmaf.h: pmaf_winner_record_doubles)
#define PMAF_WINNER_HDR 8
// doubles per population in the host-visible mailbox k_manager writes (ManagerArgs::out)
#define PMAF_MBOX 16
// health bits of a tick (mailbox entry 15; include/pmaf.h: PMAF_HEALTH_*)
#define PMAF_HB_SETPOINT_NAN 1
#define PMAF_HB_FORCE_NAN 2
#define PMAF_HB_ACC_CLAMPED 4
#define PMAF_HB_COST_NAN 8
// peer mailboxes (include/pmaf.h "peer mailboxes"): doubles per header slot (header[8], sequence number, the
// publisher's own coupling [9] = source rank, [10] = source population (-1: uncoupled; -2: unknown, a host-written
// initial header), padding to one 128-byte line) and the largest world
#define PMAF_PEER_SLOT 16
#define PMAF_PEER_MAX_WORLD 64

namespace pmaf {

// arithmetic policy of the tuned rollout kernels (see pmaf_device.hpp "arithmetic policy")
enum : int { MATH_IEEE = 0, MATH_FAST = 1, MATH_XACT = 2, MATH_FMA = 3 };

// LDS list area of the wave-per-agent step (pmaf_rollout_w64.hpp: circ_and_scale_w64) in doubles: (64 * slots + 8 padding
// + 64 scratch) entries of 4 doubles, sized for two slots per lane when the kernel has one or two and for four otherwise;
// the wave-minimum cell is the double right behind it. Host (lds_rollout / lds_manager) and kernels agree through this.
constexpr int pmaf_list_area_doubles(int slots_per_lane) { return (64 * (slots_per_lane <= 2 ? 2 : 4) + 8 + 64) * 4; }

struct PopConst {
  double dt, vel_max, approach, shell, mass, rad;
  // exact squared thresholds (computed on the host, pmaf_host.cpp:sq_gt/sq_ge):
  // for every z >= 0   sqrt(z) > 1e-5  <=>  z >= zf_gt
  //                    sqrt(z) > 13.0  <=>  z >= zacc_gt
  //                    sqrt(z) < 0.2   <=>  z <  zinit_lt
  // (sqrt is monotonic and correctly rounded, so each predicate has one
  // boundary double); they let the w64 kernel skip square roots whose value
  // is only compared, never used.
  //                    sqrt(z) < 0.5 vmax        <=>  z < zvhalf_lt
  //                    sqrt(z) < vmax - 0.1 vmax <=>  z < zv09_lt
  double zf_gt, zacc_gt, zinit_lt, zvhalf_lt, zv09_lt;
};

}  // namespace pmaf

// ---------------------------------------------------------------------------
// device-side views
// ---------------------------------------------------------------------------
struct DevView {
  int P, N, n_obs, cap;
  pmaf::PopConst C;
  // per population
  const double *goal;        // [P][3]
  double *agent_init_pos;    // [P][3]  CfAgent::init_pos_ (gate)
  double *start_pos;         // [P][3]  position all agents start the next rollout from
  double *start_vel;         // [P][3]
  double *obs_start;         // [P][7][n_obs] SoA: agents' private obstacle copies at rollout start
  int32_t *known_start;      // [P][n_obs]
  double *obs_live;          // [P][7][n_obs] SoA: live obstacles (moveRealEEAgent / resetEEAgents argument)
  // per agent
  const double *k_attr, *k_circ, *k_repel, *k_damp;  // [P][N]
  const int32_t *types;      // [N]
  double *rot;               // [P][N][3][n_obs]  field_rotation_vecs_
  const double *rnd;         // [P][N][3][n_obs]  random_vecs_
  double *paths;             // [P][N][cap][3]
  int32_t *n_points;         // [P][N]
  double *agent_vel;         // [P][N][3]
  double *min_obs;           // [P][N]
  double *cost_ws;           // [P][N]  sum of workspace penalties over the path
  double *path_len;          // [P][N]
  double *goal_dist;         // [P][N]
  int32_t *reached;          // [P][N]
  int32_t *known_out;        // [P][N][n_obs] known_obstacles_ after the rollout
  double *costs;             // [P][N]
  // real agent
  double *real_pos, *real_vel, *real_force, *real_init_pos;  // [P][3]
  int32_t *real_known;       // [P][n_obs]
  double *real_rot;          // [P][3][n_obs]
  // best agent copy
  int32_t *has_best, *best_id, *best_type;  // [P]
  double *best_rnd;          // [P][3][n_obs]
  int32_t *best_idx;         // [P] last evaluate result
  unsigned long long *step_counter;  // [1] agent-steps executed by all rollouts
  unsigned long long *pred_ticks;    // [P][N] rollout duration in wall_clock64() ticks (CfAgent::prediction_time_)
  const double *zsent_lt;            // [P] exact squared-distance boundary of the repel range test
  int ablate;                        // timing experiments only (PMAF_ABLATE), 0 in production
  int n_simds;                       // SIMDs of the device (4 per CU): a launch of more waves doubles them up
  int prio_slice_log2, prio_younger_of_8;   // wave-per-agent SLICE loop (pmaf_k_w64.hip): slice length in 2^k ticks of the 100 MHz clock, the younger wave's share
  // closest-other table of the rollout-start obstacles (round 3): closest_idx[p][i] = the index the Obstacle /
  // GoalObstacle heuristics' scan for the field obstacle nearest to obstacle i returns (B/src/cf_agent.cpp:434-446 /
  // :480-492), valid while closest_ok[p] != 0: k_manager computes it when it writes obs_start from a NEW live list whose
  // field obstacles are all at rest (then their positions -- and the table -- hold for the whole rollout and for every
  // later rollout until the caller hands over new obstacles); the wave-per-agent kernels' first-contact latch reads
  // it instead of searching.
  int32_t *closest_idx;              // [P][n_obs]
  int32_t *closest_ok;               // [P]
};

struct CostParams {
  double k_goal_dist, k_path_len, k_safe_dist, k_workspace;
  double ws[6];
};


// Peer mailboxes: every rank owns an INBOX in its device memory, [2 parities][world][P][PMAF_PEER_SLOT] doubles,
// mapped into every peer process (hipIpcOpenMemHandle). k_manager of tick t on rank r stores its populations' record
// headers + sequence number t straight into slot [t & 1][r][pop] of EVERY rank's inbox (one hop over xGMI, no
// collective launch), and -- for a population coupled to (src_rank, src_pop) -- reads the header with sequence number
// t - 1 from its OWN inbox (local HBM) as the position of its trailing repulsive obstacle. This struct lives in
// device memory; k_manager gets a pointer to it.
struct PeerView {
  int world, rank, P, pad;
  double *inbox;                          // this rank's inbox (== peer[rank])
  double *peer[PMAF_PEER_MAX_WORLD];      // every rank's inbox as mapped into THIS process
  const int32_t *couple;                  // [P][2] (src_rank, src_pop) of a coupled population, src_rank < 0: none
  const double *couple_radius;            // [P] radius given to the coupled trailing obstacle (live list)
  unsigned long long timeout_ticks;       // bound of the in-kernel wait for a peer's header (wall_clock64 ticks)
};

struct ManagerArgs {
  int do_select, do_move, do_reset;
  int reset_from_real;     // 1: reset to the real agent's state, 0: reset_in
  int rollout_follows;     // 1: the rollout kernel is launched right behind this one (pmaf_tick)
  int tuned_real_step;     // 1: real_step_w64 (default arithmetic policy, M <= 256), 0: generic LDS-table path
  const double *live_src;  // [P][7][n_obs] live obstacles in mapped pinned HOST memory (pmaf_tick: no copy command);
                           // NULL: D.obs_live already holds them
  double dt_real;
  const int32_t *agent_id; // [P] gains index for the real step; NULL = best_idx of this launch
  const double *reset_in;  // [P][6] pos, vel
  double *out;             // [P][PMAF_MBOX] host-visible: best_idx, next_pos[3], next_vel[3], dist_from_goal, force[3],
                           // seq; [12] wait for the coupled peer header, [13] peer publish (both wall_clock64
                           // ticks), [14] peer status (0 ok, 1 the awaited header did not arrive in time, 2 the source
                           // had already overwritten it -- it ran ahead, 3 the source is not coupled back to this
                           // population: couplings must be pairwise mutual), [15] health bits (PMAF_HB_*)
  double seq;              // written to out[11] after the other entries are visible to the host (0: not written)
  const PeerView *peer;    // peer mailboxes (device memory), NULL: none; used by launches with select+move+reset
  double peer_tick;        // t: sequence number this launch publishes (it consumes t - 1)
  double *winner_hdr;      // [P][winner_stride] winner-record headers (send buffer of the sharded runs' all-gather),
  int winner_stride;       // written after a selection: {cost, idx, n_points, type, next_pos[3], goal_dist}; NULL: none
  int compute_closest;     // with do_reset: the live obstacles changed since the closest-other table was last computed
                           // (set by launch_manager from the handle's dirty flag)
  // winner path (pmaf_enable_winner_path): after the set-point has been published, the selected agent's SCORED path
  // goes into mapped pinned host memory -- wp_out [P][cap][3], wp_hdr [P][4] = {n_points, agent index, 0, sequence
  // number (= seq, written last behind a system-scope fence)}; NULL: off
  double *wp_out;
  double *wp_hdr;
  // closed loop (pmaf_set_real_position = CfManager::setRealEEAgentPosition, B/src/cf_manager.cpp:216-218): the measured
  // position [P][3] in mapped pinned HOST memory replaces the real agent's latest position before anything else of this
  // launch reads it (no sync, no copy command in front of the tick); NULL: D.real_pos holds it
  const double *real_pos_src;
  // ... and for handles of at most PMAF_RP_INLINE populations (the node's single manager) BY VALUE in the kernel arguments:
  // the position arrives with the dispatch packet instead of through a PCIe read on the real step's critical path
  // (closed-loop set-point latency 15.8 -> 14.4 us at C2, the open-loop figure). real_pos_inline = 1: use real_pos_val.
  int real_pos_inline;
  double real_pos_val[3 * 4];
  // the individual calls of the node's sequence (pmaf_move_real / pmaf_reset_agents) hand their small inputs over the same
  // way for <= PMAF_RP_INLINE populations instead of through a synchronous copy each (round 5: the five calls 106 -> ~45 us)
  int agent_id_inline;          // 1: agent_id_val[pop] is the gains index of the real step
  int32_t agent_id_val[4];
  int reset_in_inline;          // 1: reset_in_val[pop][6] = pos, vel
  double reset_in_val[6 * 4];
};
#define PMAF_RP_INLINE 4

// synchronous stepping (CfAgent::cfPlanner, B/src/cf_agent.cpp:278-300): k_plan_steps
struct PlanArgs {
  const double *obs;       // [P][7][n_obs] SoA: the CALLER's obstacle list (positions, velocities, radii), not advanced
  double dt;               // the call's delta_t
  int steps;               // steps per cfPlanner call
  const int32_t *only;     // [P] agent index to step (CfManager::moveAgent), NULL: every agent (moveAgents)
  int max_calls;           // moveAgent: repeat cfPlanner(steps) while distGoal > 0.05, at most this often; 1 otherwise
  int until_goal;          // 1: moveAgent's loop condition applies
  int32_t *calls_out;      // [P] cfPlanner calls made for the `only` agent (moveAgent), may be NULL
};

// ---------------------------------------------------------------------------
// launch interface: implemented in pmaf_k_w64.hip / pmaf_k_grp.hip / pmaf_k_misc.hip
// ---------------------------------------------------------------------------
// k_rollout_w64<TILES, MATH, DPPSUM, PLAIN> on grid (N, P); tiles in {1,2,4}; plain: every k_attr != 0 and unit mass
// (the step without those two cases, pmaf_k_w64.hip); dppsum: the ordered force sum by the DPP
// row-broadcast chain (always for tiles >= 2) or by LDS batches. Returns false if this build holds no such variant.
// e0 / e1 (may be NULL): HIP events attached to the kernel's own dispatch packet (hipExtLaunchKernel start / stop
// events) -- no marker packets in the stream, so timing a launch does not put anything between two kernels
bool pmaf_k_launch_w64(const DevView &D, const CostParams &cp, int tiles, int math, bool dppsum, bool plain, size_t lds,
                       hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, bool slice = false);
// k_rollout_grp<LPA, TILES, MATH> on grid (n_blocks, P); lpa in {8,16,32}, tiles in {1,2,4}, math XACT, IEEE or FMA
// W waves per agent (pmaf_k_mw.hip): waves in 2..4, per = field obstacles per wave (<= 61); every policy but MATH_IEEE
// lds_kb: dynamic LDS per block in KB (0: the launcher's placement rule)
bool pmaf_k_launch_mw(const DevView &D, const CostParams &cp, int waves, int per, int math, bool plain, int lds_kb,
                      hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
bool pmaf_k_launch_grp(const DevView &D, const CostParams &cp, int lpa, int tiles, int math, int n_blocks, size_t lds,
                       hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// generic k_rollout<LPA>, any power-of-two lpa 1..64
bool pmaf_k_launch_generic(const DevView &D, const CostParams &cp, int lpa, int n_blocks, size_t lds, hipStream_t s,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// done (may be NULL): event signalled by the manager kernel's completion (the winner exchange waits for it)
void pmaf_k_launch_manager(const DevView &D, const CostParams &cp, const ManagerArgs &A, size_t lds, hipStream_t s,
                           hipEvent_t done = nullptr);
void pmaf_k_launch_score(const DevView &D, const CostParams &cp, hipStream_t s);
void pmaf_k_launch_restart_paths(const DevView &D, const double *pos, hipStream_t s);
void pmaf_k_launch_link_force(int n, const double *link_pos, const double *k_r, const double *sent, double rad,
                              double shell, double *out, hipStream_t s);
void pmaf_k_launch_debug_math(int op, int n, const double *a, const double *b, double *out, hipStream_t s);
void pmaf_k_launch_winner(const DevView &D, double *dst, hipStream_t s);
// path part of the winner records whose headers k_manager wrote into dst: the selected agents' paths out of `paths`
void pmaf_k_launch_winner_path(const DevView &D, const double *paths, double *dst, hipStream_t s);
// CfManager::moveAgents / moveAgent (synchronous stepping), any power-of-two lpa 1..64
bool pmaf_k_launch_plan_steps(const DevView &D, const PlanArgs &A, int lpa, int n_blocks, size_t lds, hipStream_t s);
// CfManager::setEEAgentPositions / setEEAgentPosAndVels: pos [P][3], vel [P][3] or NULL
void pmaf_k_launch_set_agents(const DevView &D, const double *pos, const double *vel, hipStream_t s);
// CfAgent::evalObstacleDistance for every agent: obs [P][7][n_obs] SoA, out [P][N]
void pmaf_k_launch_eval_obstacle_distance(const DevView &D, const double *obs, double *out, hipStream_t s);
// opt the kernels that take dynamic LDS into more than the 64 KB default
hipError_t pmaf_k_set_lds_limits(size_t lds_manager, size_t lds_rollout);
