// go2_tables.h — the robot tables the kernels stage into LDS, and the per-launch parameter block.
// Filled on the host from include/go2_model_data.h (numbers generated from the Go2 URDF).
#pragma once
#include <stdint.h>
#include "../../include/go2sim.h"
#include "../../include/go2_model_data.h"

#define GO2_NLEG_OTHER GO2_LEG_OTHER_PTS
#define GO2_LANE_BASE_PTS 4   // the base / head points are dealt to the 4 legs (i & 3 == leg) and there to the 4 sub-lanes
static_assert(GO2_BASE_PTS <= 4 * GO2_LANE_BASE_PTS, "one base point per (leg, sub-lane)");
// the per-leg candidates are tabulated link by link (gen_go2_model.py emits hip, thigh, calf in this order, the counts GO2_N_*_PTS with them; checked at create)
static_assert(GO2_N_HIP_PTS == 2 && GO2_N_THIGH_PTS == 12 && GO2_N_CALF_PTS == 8, "the dealing below: 3 thigh + 2 calf points per sub-lane, the 2 hip points on sub-lanes 2, 3");
static_assert(GO2_N_HIP_PTS + GO2_N_THIGH_PTS + GO2_N_CALF_PTS == GO2_NLEG_OTHER, "candidate layout");

// the collision candidates ONE sub-lane of a leg tests (go2_lane.h phaseC).  Seven table slots with a fixed link type each, so that the
// link's pose is known at compile time: slots 0..2 thigh points (8 box corners + round 4's 4 long-edge midpoints), slots 3, 4 calf points (6 capsule
// end spheres + round 4's 2 mid-segment spheres), slot 5 a hip point (sub-lanes 2, 3), slot 6 one of the leg's share of the base / head points
// (sub-lane k < n_base).  idx = position in the sequential scan order (0..21 leg points, 22.. base points; tie-break), -1 = empty slot.
#define GO2_SUB_CANDS 7
#define GO2_SC_HIP 5
#define GO2_SC_BASE 6
struct SubCand { float pt[GO2_SUB_CANDS][4]; int32_t idx[GO2_SUB_CANDS]; int32_t body[GO2_SUB_CANDS]; };

// per-leg constant table (one per leg index 0..3); plain floats/ints so it can be memcpy'd into LDS
struct LegTab {
  float o1[3], o2[3], o3[3];        // joint origins in the parent link frame (hip origin is in the base frame)
  float body[4][10];                // hip, thigh, calf, foot: {m, h(3), J(6: xx,yy,zz,xy,xz,yz)} about the moving link origin, link axes
  float lim_lo[3], lim_hi[3], vel_lim[3], eff_lim[3];
  float foot_pt[4];                 // collision sphere: centre in calf frame, radius
  float foot_off[4];                // foot body frame origin in the calf frame (+pad)
  float other_pt[GO2_NLEG_OTHER][4];
  int32_t other_link[GO2_NLEG_OTHER];   // 1 hip, 2 thigh, 3 calf
  int32_t other_body[GO2_NLEG_OTHER];   // body index in the 19-body list
  float base_pt[GO2_LANE_BASE_PTS][4];  // this lane's share of the base-attached candidates (base frame)
  int32_t base_body[GO2_LANE_BASE_PTS];
  int32_t n_base;
  int32_t body_index[4];            // hip, thigh, calf, foot body indices
  int32_t mass_ratio_index[4];      // index into link_mass_ratio[18] (= body index - 1)
  SubCand cand[4];                  // the same candidates, dealt to the 4 sub-lanes
  float cull_ext[4][4];             // per group (hip, thigh, calf, base share): half-extents (+ radius) of the group's spheres along the link axes
};
struct BaseTab {
  float m0, c0[3], Ic0[6];          // base body: mass, COM, inertia about COM
  float head[2][10];                // Head_upper, Head_lower about the base origin
  float body_off[3][4];             // frame origins of base, Head_upper, Head_lower in the base frame
};
struct Go2Tables { LegTab leg[4]; BaseTab base; uint8_t slot_code[GO2_NUM_UNIFORMS]; /* include/go2sim_rng.h */ int32_t layout_ok; };

// The contact slots of a leg besides the foot.  One CANDIDATE per body group (go2_lane.h phaseC): the deepest calf sphere, the deepest thigh
// point, the deepest hip sphere, the deepest of the leg's share of the base / head points; the groups that are in contact are then COMPACTED
// into "virtual slots" 0, 1, 2, 3 in that order (a leg with a calf and a base contact uses virtual slots 0 and 1), so that the work of a wave
// follows the largest number of simultaneous non-foot contacts any of its legs has, not the number of groups that are in contact somewhere.
// Virtual slot 0 keeps its rows in registers (like the foot); the rows of virtual slots 1..3 and the joint-limit rows are parked in LDS between
// their construction and the end of the substep's solve and pass through one register-resident slot while they are swept:
//   win per leg  : the group's winning candidate as the sub-lane that found it left it: gap, normal, centre (base frame), radius, body
//   jy  per lane : this sub-lane's 3-number slices J, Y of the slot's three rows (normal, two tangents; limits: the three joints)
//   dp  per lane : this sub-lane's part of the row's diagonal (summed over the quad, with the split base, in solve_prepare)
//   sc  per leg  : [0] free velocity + bias, [1] inverse diagonal (written by solve_prepare)
//   nrm per leg  : world contact normal (the tangents follow from it)
#define GO2_NTYPE 4          // body groups: calf, thigh, hip, base share (canonical sweep order)
#define GO2_T_CALF 0
#define GO2_T_THIGH 1
#define GO2_T_HIP 2
#define GO2_T_BASE 3
#define GO2_NPARK 4          // parked row sets: virtual slots 1, 2, 3 and the joint limits
#define GO2_PARK_LIMITS 3
#define GO2_WG_LANES 256
struct Go2RowsLds {
  float jy[GO2_NPARK][3][6][GO2_WG_LANES];
  float dp[GO2_NPARK][3][GO2_WG_LANES];
  float sc[GO2_NPARK][3][2][GO2_WG_LANES / 4];
  float nrm[GO2_NPARK - 1][3][GO2_WG_LANES / 4];
  float win[GO2_NTYPE][9][GO2_WG_LANES / 4];
};

#define GO2_TOP_CELL 4        // cells per side of a block of the coarse "highest surface nearby" map
#define GO2_TOP_REACH 8       // ... dilated by this many cells (0.8 m at the tasks' 0.1 m grid: a leg's spheres stay within 0.6 m of its hip joint,
                              //     the base / head points within 0.4 m of the base origin)
// the contact surface over one grid cell: heights (vscale units) at the corners (i,j) (i+1,j) (i,j+1) (i+1,j+1) as seen from inside the cell
// (include/go2sim.h Go2SimCfg.hf_cells); one aligned 8-byte load per contact query
struct alignas(8) Go2Cell { int16_t h[4]; };
// With hf_walls (mesh_type 'trimesh') a cell's record also carries what the query needs of its eight neighbours, so that a query is ONE 32-byte load
// instead of six scattered 8-byte ones: the heights of the four edge neighbours along the common edge (xm: cell (i-1,j) at its corners 1, 3; xp: (i+1,j) at
// 0, 2; ym: (i,j-1) at 2, 3; yp: (i,j+1) at 0, 1) and of the four diagonal neighbours at the common corner (dg[k]: at this cell's corner k).  A neighbour
// outside the grid repeats the cell's own heights (no face, no edge).  Built at go2sim_create from hf_cells.
struct alignas(16) Go2CellW { int16_t h[4]; int16_t xm[2], xp[2], ym[2], yp[2]; int16_t dg[4]; };

// Device/host pointers of every per-env field.  The HIP library stores per-env fields FIELD-MAJOR (SoA):
// logical [N, a, b] lives at ((b_idx * A + a_idx) * N + env), i.e. C order of the reversed logical dims,
// so that consecutive lanes (envs) touch consecutive addresses.  obs_buf and privileged_obs_buf are the
// exception: they are the policy's GEMM inputs and stay row-major [N,45] / [N,263].
#define GO2_PTRS_BODY(G) \
  G float *root, *dof, *contact, *rigid, *obs, *priv, *rew; G uint8_t *reset, *time_out; G int64_t* ep_len; \
  G float *torques, *actions, *last_actions, *last_last_actions, *last_dof_vel, *last_root_vel, *commands, *cmd_timer, *cmd_xy_acc; \
  G uint8_t *stop_heading, *last_is_limit_vel; G float *base_lin_vel, *base_ang_vel, *proj_gravity, *rpy, *heights, *max_move, *to_timer, *feet_air_time; \
  G uint8_t *last_contacts, *last_contacts2; G float *strength, *zero_off, *kp_mul, *kd_mul, *origins; G int64_t *terrain_levels, *terrain_types; \
  G float *ep_sums, *friction, *restitution, *added_mass, *added_com, *mass_ratio, *episode_info, *foot_impulse; \
  /* internal */ \
  G int32_t* terrain_kind; G uint8_t* reset_mask /*[N], go2sim_reset_idx*/; G float* ep_accum /*[NUM_REWARDS+2]: episode sums of the envs reset in this pass, their number, #envs resampled by the callback*/; const G float* inj_storage; const G int16_t* hf; const G Go2Cell* hf_cells; const G int16_t* hf_top /*[hf_trows][hf_tcols]: coarse map of the highest surface near a point (go2sim_create)*/; const G float* terrain_origins; \
  const G Go2Tables* tables; G long long* dbg_clock;
struct Go2Ptrs { GO2_PTRS_BODY() };
// The kernels read these pointers out of the device block (scalar loads), where the compiler cannot know their address space and would
// emit FLAT loads/stores (which also tick the LDS counter: every ds_read wait then waits for outstanding stores).  The kernel-side view
// of the same bytes types them as global (address space 1) -> global_load / global_store.
#if defined(__HIP_DEVICE_COMPILE__)
#define GO2_GLOBAL_AS __attribute__((address_space(1)))
struct Go2PtrsK { GO2_PTRS_BODY(GO2_GLOBAL_AS) };
static_assert(sizeof(Go2PtrsK) == sizeof(Go2Ptrs), "same layout");
#define GO2_GENERIC(T, ptr) ((T)(ptr))          /* explicit global -> generic cast for a callee that takes a plain pointer */
#define GO2_AS1 GO2_GLOBAL_AS
#define GO2_AS3 __attribute__((address_space(3)))
#else
#define GO2_AS1
#define GO2_AS3
typedef Go2Ptrs Go2PtrsK;
#define GO2_GENERIC(T, ptr) (ptr)
#endif

// everything a launch needs besides pointers: config constants + the host-side scalars of this step
struct Go2Launch {
  double hf_inv_hscale;   // 1.0 / (double)hf_hscale: x * this, rounded to fp32, is the correctly rounded fp32 quotient x / hf_hscale (go2_math.h)
  int32_t N, env_offset, decimation, solver_iterations;
  uint32_t seed_lo, seed_hi;
  float sim_dt, dt, gravity[3], contact_offset, erp, max_depen_vel, bounce_thr, cfm, armature, limit_margin, max_lin_vel, max_ang_vel;
  int32_t terrain_mode, hf_walls, hf_rows, hf_cols; float hf_hscale, hf_vscale, hf_border, terrain_friction, terrain_restitution;
  // candidate cull on the height field (go2_lane.h phaseC): hf_top[i][j] = highest corner of any cell within GO2_TOP_REACH cells of the
  // GO2_TOP_CELL x GO2_TOP_CELL block (i, j); hf_nzmin = smallest z component of any facet normal of the map
  int32_t hf_trows, hf_tcols; float hf_nzmin;
  int32_t terrain_num_levels, terrain_num_types, terrain_curriculum, move_down_by_acc, measure_heights, full_body_states; float terrain_length;
  float kp[12], kd[12], q0[12], action_scale, clip_actions, clip_obs, base_init[13];
  int32_t control_type;   // 0 P, 1 V, 2 T (legged_robot.py:607-617)
  int32_t cmd_track_curr; float cmd_max_curr;   // commands.curriculum / max_curriculum (:728-737)
  int32_t rand_strength, rand_offset, rand_pd, push_robots, push_interval, rand_delay;
  float strength_rng[2], offset_rng[2], kp_rng[2], kd_rng[2], push_xy, push_ang;
  float resampling_time; int32_t heading_command, dynamic_resample; float limit_vel_prob; int32_t limit_invert, stop_heading_at_limit;
  float limit_ang_zero_prob; int32_t comb_count; float comb[36][3];
  float cmd_ranges0[4][2], terrain_max_cmd[9][4][2];
  float rew_scale_dt[GO2_NUM_REWARDS];  // raw scale * dt
  float rew_to_scale_dt[GO2_NUM_REWARDS];   // turn_over_scales * dt (all 0 unless init_state.turn_over)
  int32_t rew_on[GO2_NUM_REWARDS];      // term computed: non-zero in either table (legged_robot.py:927-930)
  uint32_t rew_mask_all;                // bit t = rew_on[t]
  int32_t turn_over; float to_prop[3], to_height[2][2], to_zero_time[2], to_roll_thr;
  int32_t rew_curr_count, rew_curr_term[4]; float rew_curr[4][4];     // curriculum_rewards (start_iter,end_iter,start,end)
  int32_t cmd_curr_count; float cmd_curr[4][9];                       // command_range_curriculum
  int32_t zero_curr_enabled; float zero_curr[4]; int32_t num_steps_per_env;
  int32_t only_positive; float tracking_sigma; int32_t dyn_sigma; float dyn_sigma_vel[4], dyn_sigma_max[9];
  float soft_vel_limit, soft_torque_limit, base_height_target, max_contact_force, min_legs_distance, soft_limits[12][2];
  float os_lin, os_ang, os_dof_pos, os_dof_vel, os_height; int32_t add_noise; float noise_vec[GO2_NUM_OBS];
  float max_episode_length, episode_length_s;
};

// counters that live in device memory and are advanced by the device itself, so that a step is a pure
// enqueue (HIP-graph replayable): nothing the host computes per step is baked into kernel arguments
struct Go2Dyn { uint64_t step_count; int64_t common_step_counter; int32_t use_injected;
                int32_t cmd_stage_seen;   // command_range_curriculum stage whose lin_vel_x the tracked list was last set from (-2: none yet)
                float cmd_x_range[2];     // command_ranges['lin_vel_x'] as update_command_curriculum (:728-737) keeps it
                int32_t cb_any, pad2; };  // heading_command only: some env of the batch is resampled by the coming step's callback (go2_cb_scan_kernel)

// per-step scalars (what the reference keeps as Python floats), recomputed by every workgroup from
// Go2Dyn.common_step_counter into LDS: they are pure functions of counter // num_steps_per_env
struct Go2Step {
  uint32_t step_lo, step_hi; int32_t initial_reset /* 0 step, 1 reset_idx(all) before the first step, 2 go2sim_reset_idx */; const float* injected;
  float rew_scale[GO2_NUM_REWARDS];  // scale * dt * curriculum
  float rew_to_scale[GO2_NUM_REWARDS];   // turn_over scale * dt * curriculum
  float cmd_ranges[4][2], max_lin_vel, zero_cmd_proba;
  // The reference refreshes its command ranges lazily — inside _resample_commands, when called with >= 1 env (legged_robot.py:433-446).  The
  // resampled commands themselves therefore always see the stage of the current iteration (cmd_ranges above); the heading controller's clip
  // (:411-419), which runs for every env right after the callback's _resample_commands, sees the new stage only if SOME env of the batch was
  // resampled by the callback in this step.  stage_pending: a stage has started that no _resample_commands call has picked up yet
  // (heading_command only) -> Go2Dyn.cb_any, scanned from the batch's command timers by a small kernel in FRONT of the step kernel (inside
  // it, other workgroups would already be rewriting the timers); yaw_range_seen: ang_vel_yaw of the stage last picked up.
  int32_t stage_pending; float yaw_range_seen[2];
  uint32_t rew_mask;                 // bit t: reward term t is computed (Go2Launch.rew_on[t]); 0 during the initial reset
  uint32_t rew_mask_all;             // the same, also during the initial reset
};
struct Go2DevBlock { Go2Ptrs p; Go2Launch L; Go2Dyn dyn; };
