/* go2sim.h — C ABI of the MI355X-native vectorised Go2 simulator.
 *
 * This is the drop-in boundary for the hot path of wty-yy/go2_rl_gym: it replaces the Isaac Gym
 * tensor API as the reference uses it (acquire_*_tensor / refresh_*_tensor / simulate /
 * set_dof_actuation_force_tensor / set_*_state_tensor_indexed; call sites
 * legged_gym/envs/base/legged_robot.py:82-92,107-109,632-634,705-707,722-724,769-787) AND the
 * per-env torch code around it (LeggedRobot.step :60-100, post_physics_step :102-142,
 * reset_idx :180-245, _resample_commands :423-592, _compute_torques :594-618, _push_robots :709-724,
 * compute_reward :247-274 + _reward_* :1228-1441, Go2Robot.compute_observations go2_env.py:23-53,
 * RolloutStorage.compute_returns rsl_rl/rsl_rl/storage/rollout_storage.py:123-137).
 *
 * Two libraries export exactly this ABI:
 *   go2_rl_gym_amd/csrc  -> libgo2sim_hip.so   the product: HIP kernels for gfx950; buffers are device
 *                                              pointers; `stream` is a hipStream_t.
 *   oracle/              -> libgo2oracle_f32/f64.so   TEST INFRASTRUCTURE ONLY: a plain-C CPU
 *                                              restatement; buffers are host pointers; `stream` ignored.
 *
 * Conventions: all functions return 0 on success, a negative GO2SIM_E* code on failure, and never
 * throw across the ABI.  One handle = one caller thread at a time; distinct handles are independent.
 * All calls on the HIP library are asynchronous on the passed stream.  The library owns every buffer
 * it hands out; pointers stay valid until go2sim_destroy.  Quaternions are (x,y,z,w)
 * (legged_robot_config.py:91).  All floating point is fp32 (the f64 oracle build widens everything).
 */
#ifndef GO2SIM_H
#define GO2SIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO2SIM_ABI_VERSION 8   /* 8: Go2GatherJob.dst_pitch (a gathered row may land inside a wider destination row: the [latent | obs] input matrices of CTS), any nclear;  7: + go2sim_shuffle_gather / go2sim_shuffle_index;  6: go2sim_ppo_loss: workspace 24*ceil(B/64) floats (was B/256), A <= 16;  5: Go2SimCfg gained control_type / cmd_tracking_curriculum / cmd_max_curriculum, go2sim_reset_idx, episode_info is GO2_EPISODE_INFO_LEN long;  4: Go2SimCfg gained hf_cells / hf_walls (trimesh wall geometry); test hooks go2sim_debug_*;  3: 3: Go2SimCfg gained max_linear_velocity / max_angular_velocity; go2sim_elu_backward_bias workspace is C*ceil(B/64) */

#define GO2SIM_EINVAL   (-1)  /* bad argument / config */
#define GO2SIM_ENOMEM   (-2)
#define GO2SIM_EDEVICE  (-3)  /* HIP runtime error (message via go2sim_last_error) */
#define GO2SIM_ENOTSUP  (-4)

#define GO2_NUM_OBS 45           /* go2_config.py:34 */
#define GO2_NUM_PRIV_OBS 263     /* go2_config.py:36 */
#define GO2_NUM_ACTIONS 12
#define GO2_NUM_BODIES_ABI 19
#define GO2_NUM_HEIGHT_POINTS 187 /* 17 x 11, legged_robot_config.py:26-27 */

/* Reward terms, in the order the library sums them.  (The reference iterates a Python set,
 * legged_robot.py:927-936, so its order is not defined; ours is this enum.)  A term is active
 * iff its scale != 0 (legged_robot.py:914-920). */
enum {
  GO2_REW_TRACKING_LIN_VEL = 0, /* :1322 */
  GO2_REW_TRACKING_ANG_VEL,     /* :1336 */
  GO2_REW_LIN_VEL_Z,            /* :1228 */
  GO2_REW_ANG_VEL_XY,           /* :1232 */
  GO2_REW_ORIENTATION,          /* :1236 */
  GO2_REW_BASE_HEIGHT,          /* :1245 */
  GO2_REW_TORQUES,              /* :1261 */
  GO2_REW_DOF_VEL,              /* :1265 */
  GO2_REW_DOF_ACC,              /* :1269 */
  GO2_REW_ACTION_RATE,          /* :1273 */
  GO2_REW_COLLISION,            /* :1277 */
  GO2_REW_DOF_POS_LIMITS,       /* :1285 */
  GO2_REW_DOF_VEL_LIMITS,       /* :1291 */
  GO2_REW_TORQUE_LIMITS,        /* :1296 */
  GO2_REW_FEET_AIR_TIME,        /* :1347 */
  GO2_REW_STUMBLE,              /* :1360 (scale key feet_stumble has no function in the reference; key "stumble") */
  GO2_REW_STAND_STILL,          /* :1365 */
  GO2_REW_FEET_CONTACT_FORCES,  /* :1369 */
  GO2_REW_ACTION_SMOOTHNESS,    /* :1373 */
  GO2_REW_DOF_POWER,            /* :1381 */
  GO2_REW_CORRECT_BASE_HEIGHT,  /* :1399 */
  GO2_REW_FEET_REGULATION,      /* :1404 */
  GO2_REW_SIMILAR_TO_DEFAULT,   /* :1416 */
  GO2_REW_UPRIGHT,              /* :1420 */
  GO2_REW_LEGS_DISTANCE,        /* :1423 */
  GO2_REW_HIP_TO_DEFAULT,       /* go2_env.py:55 */
  GO2_REW_X_COMMAND_HIP_REGULAR,/* go2_env.py:62 */
  GO2_REW_TERMINATION,          /* :1281, added after the positive clip (:271-274) */
  GO2_NUM_REWARDS
};
#define GO2_NUM_TERRAIN_KINDS 9   /* wave, slope, rough slope, stairs up, stairs down, obstacles, stepping stones, gap, flat (go2_config.py:129-139) */
#define GO2_EPISODE_INFO_LEN (GO2_NUM_REWARDS + 13)   /* Go2SimBuffers.episode_info */

/* Per-env per-step uniform slots.  In normal operation each slot is Philox4x32-10(key = seed,
 * counter = {global env id, slot/4, step_lo, step_hi})[slot%4] mapped to [0,1); in test mode
 * (go2sim_inject_uniforms) the caller supplies the [num_envs][GO2_NUM_UNIFORMS] table, which is how
 * parity with the reference's torch.rand draws is checked. */
enum {
  GO2_U_DELAY = 0,            /* legged_robot.py:72   randint(0, decimation+1) */
  GO2_U_RSA = 1,              /* resample in _post_physics_step_callback: x,y,yaw,prob,comb,ang,dir  (:454-477,509,525,571,575) */
  GO2_U_RESET_STRENGTH = 8,   /* :197 (12) */
  GO2_U_RESET_OFFSET = 20,    /* :202 (12) */
  GO2_U_RESET_KP = 32,        /* :205 (12) */
  GO2_U_RESET_KD = 44,        /* :206 (12) */
  GO2_U_RESET_TERRAIN = 56,   /* :1166 */
  GO2_U_RESET_DOF = 57,       /* :628 (12) */
  GO2_U_RESET_YAW = 69,       /* :645 */
  GO2_U_RESET_XY = 70,        /* :698 (2) */
  GO2_U_RESET_VEL = 72,       /* :703 (6) */
  GO2_U_RSB = 78,             /* resample inside reset_idx (:227): same 7 slots as RSA */
  GO2_U_PUSH = 85,            /* :718-719 (2 + 3) */
  GO2_U_NOISE = 90,           /* go2_env.py:53 (45) */
  GO2_U_TURN = 135,           /* init_state.turn_over (:654,661,672,674): category, backflip height, sideflip height, side sign */
  GO2_NUM_UNIFORMS = 140      /* padded to a multiple of 4 */
};

typedef struct Go2SimCfg {
  uint32_t struct_size;       /* sizeof(Go2SimCfg), checked */
  uint32_t abi_version;       /* GO2SIM_ABI_VERSION */
  int32_t  num_envs;          /* envs in this shard (one shard per process/GPU) */
  int32_t  env_offset;        /* global index of this shard's env 0 (multi-GPU sharding) */
  int32_t  num_envs_global;   /* total envs over all shards */
  int32_t  _pad0;
  uint64_t seed;

  /* ---- sim: legged_robot_config.py:242-259 ---- */
  float    sim_dt;            /* 0.005 */
  int32_t  decimation;        /* 4 (go2_config.py:85) */
  float    gravity[3];        /* 0,0,-9.81 */
  int32_t  solver_iterations; /* PGS sweeps per substep */
  float    contact_offset;    /* 0.01: candidate becomes a constraint when gap < contact_offset */
  float    erp;               /* fraction of penetration removed per substep */
  float    max_depenetration_velocity; /* 1.0 */
  float    bounce_threshold_velocity;  /* 0.5 */
  float    contact_cfm;       /* relative softness added to the Delassus diagonal */
  float    joint_armature;    /* 0 (legged_robot_config.py:133) */
  float    joint_limit_margin;/* rad: joint-limit row active within this distance of a hard limit */
  float    max_linear_velocity;  /* 1000 m/s   (asset.max_linear_velocity,  legged_robot_config.py:132): the base's linear velocity is clamped */
  float    max_angular_velocity; /* 1000 rad/s (asset.max_angular_velocity, :131): ... and its angular velocity; the state stays finite */

  /* ---- terrain: legged_robot_config.py:15-40 ---- */
  int32_t  terrain_mode;      /* 0 = plane, 1 = heightfield */
  float    terrain_friction;  /* 1.0 */
  float    terrain_restitution; /* 0 */
  int32_t  hf_rows, hf_cols;  /* height_samples[hf_rows][hf_cols], x -> rows (legged_robot.py:1213-1220) */
  float    hf_hscale, hf_vscale, hf_border;
  const int16_t* hf_samples;  /* HOST pointer, copied at create; NULL for plane */
  const int16_t* hf_cells;    /* HOST [hf_rows-1][hf_cols-1][4], copied at create, or NULL: the contact surface cell by cell — heights (vscale units)
                               * of the surface at the cell's corners (i,j) (i+1,j) (i,j+1) (i+1,j+1) as seen from INSIDE the cell.  NULL = the
                               * continuous surface of hf_samples (mesh_type 'heightfield').  mesh_type 'trimesh' passes the surface of the mesh
                               * convert_heightfield_to_trimesh builds with terrain.slope_treshold (legged_gym/utils/terrain.py:46-49,
                               * legged_robot.py:1127-1141): 1-cell ramps become flat cells ending in a vertical face, i.e. neighbouring cells
                               * that disagree on their common edge (go2_rl_gym_amd/utils/terrain.py:displaced_cell_heights). */
  int32_t  hf_walls;          /* 1: the contact query also tests the vertical faces between disagreeing neighbour cells and the vertical edge at the nearest cell
                               * corner where the diagonal-neighbour cell stands higher (outside corners of stair rings and blocks) (set with hf_cells) */
  int32_t  terrain_num_levels, terrain_num_types; /* 10 x 20 */
  const float* terrain_origins; /* HOST [levels][types][3]; NULL for plane */
  const int32_t* terrain_type_id; /* HOST [types] -> terrain kind 0..8 (terrain.cols2id); NULL for plane */
  int32_t  terrain_curriculum;  /* cfg.terrain.curriculum */
  int32_t  max_init_terrain_level;
  int32_t  move_down_by_accumulated_xy_command;
  float    terrain_length;    /* 8.0: used by the resampler even on a plane (:447) */
  float    env_spacing;       /* 3.0 (plane grid, :1081-1091) */
  int32_t  measure_heights;   /* 1 */
  int32_t  full_body_states;  /* 0: rigid_body_states is filled for the 4 feet only — the only rows the reference reads
                               * (legged_robot.py:1252,1407-1408,1426); 1: all 19 bodies (3.7x the bytes of that tensor) */

  /* ---- control: go2_config.py:77-85 ---- */
  float    kp[12], kd[12];
  float    default_dof_pos[12];
  float    action_scale;      /* 0.25 */
  int32_t  control_type;      /* _compute_torques (legged_robot.py:607-617): 0 'P' position targets (every go2 task), 1 'V' velocity targets
                               * tau = Kp (a s - qd) - Kd (qd - last_dof_vel) / sim_dt, 2 'T' tau = a s; all clipped to the torque limits */
  float    clip_actions;      /* 100 */
  float    clip_observations; /* 100 */

  /* ---- init state ---- */
  float    base_init_state[13]; /* pos(3) quat(4) lin(3) ang(3), :1000-1001 */

  /* ---- domain randomisation: go2_config.py:41-75 ---- */
  int32_t  randomize_friction;      float friction_range[2];
  int32_t  randomize_restitution;   float restitution_range[2];
  int32_t  randomize_base_mass;     float added_mass_range[2];
  int32_t  randomize_link_mass;     float link_mass_range[2];
  int32_t  randomize_base_com;      float base_com_range[2];
  int32_t  randomize_pd_gains;      float stiffness_mult_range[2]; float damping_mult_range[2];
  int32_t  randomize_motor_zero_offset; float motor_zero_offset_range[2];
  int32_t  randomize_motor_strength;    float motor_strength_range[2];
  int32_t  push_robots;             int32_t push_interval; float max_push_vel_xy; float max_push_ang_vel;
  int32_t  randomize_action_delay;

  /* ---- commands: go2_config.py:97-146 ---- */
  float    cmd_resampling_time;     /* 5 s */
  int32_t  heading_command;         /* 0 */
  int32_t  dynamic_resample_commands; /* 1 */
  float    limit_vel_prob;          /* 0.2 */
  int32_t  limit_vel_invert_when_continuous;
  int32_t  stop_heading_at_limit;
  float    limit_ang_vel_at_zero_command_prob; /* 0.2 */
  int32_t  limit_vel_comb_count;    /* rows of product(limit_vel...) = 12 (:827-831) */
  float    limit_vel_comb[36][3];   /* up to 36 rows of {-1,0,1} */
  int32_t  zero_cmd_curriculum_enabled; float zero_cmd_curriculum[4]; /* start_iter,end_iter,start_value,end_value (:556-557) */
  float    cmd_ranges[4][2];        /* lin_vel_x, lin_vel_y, ang_vel_yaw, heading (go2_config.py:142-146) */
  int32_t  cmd_curriculum_count;    /* command_range_curriculum entries (:433-446) */
  float    cmd_curriculum[4][9];    /* {iter, x0,x1, y0,y1, yaw0,yaw1, h0,h1} */
  float    terrain_max_cmd_ranges[GO2_NUM_TERRAIN_KINDS][4][2]; /* per terrain kind (go2_config.py:129-139) */
  int32_t  cmd_tracking_curriculum; /* commands.curriculum (legged_robot_config.py:44; off in every go2 config): update_command_curriculum
                                     * (legged_robot.py:728-737) widens command_ranges['lin_vel_x'] by 0.5 per reset step whose mean
                                     * tracking_lin_vel episode sum exceeds 80 % of the maximum.  In this fork that list only feeds
                                     * extras['episode']['max_command_x'] (:241-242): _resample_commands samples env_command_ranges, which is
                                     * rebuilt from command_ranges only when a command_range_curriculum stage starts (:433-446), and the stage
                                     * overwrites lin_vel_x first.  episode_info[GO2_NUM_REWARDS+1..+2] carry the list's two ends. */
  float    cmd_max_curriculum;      /* commands.max_curriculum (1.0) */

  /* ---- rewards: go2_config.py:156-205 ---- */
  float    reward_scales[GO2_NUM_REWARDS]; /* RAW scales (before x dt); 0 = inactive */
  int32_t  only_positive_rewards;
  float    tracking_sigma;          /* 0.25 */
  int32_t  dynamic_sigma_enabled;   float dynamic_sigma_vel[4]; /* min_lin,max_lin,min_ang,max_ang */
  float    dynamic_sigma_max[9];
  float    soft_dof_pos_limit;      /* 0.9 */
  float    soft_dof_vel_limit, soft_torque_limit;
  float    base_height_target;      /* 0.38 */
  float    max_contact_force;       /* 147 */
  float    min_legs_distance;       /* 0.1 */
  /* init_state.turn_over (legged_robot_config.py:97-102, commands.turn_over_zero_time :66-69, rewards.turn_over_* :193-212):
   * resets start a share of the robots on their back / side; while |roll| > threshold the turn_over_scales replace
   * reward_scales (legged_robot.py:257-265); base-contact termination is off (:174); commands stay zero for zero_time after
   * such a reset (:586-590) */
  int32_t  turn_over;
  float    turn_over_proportions[3];        /* backflip, sideflip, no flip */
  float    turn_over_init_heights[2][2];    /* [backflip|sideflip][lo|hi] */
  float    turn_over_zero_time[2];          /* [backflip|sideflip] seconds */
  float    turn_over_roll_threshold;        /* pi/4 */
  float    turn_over_scales[GO2_NUM_REWARDS];
  int32_t  reward_curriculum_count; /* curriculum_rewards entries (go2_config.py:161-166) */
  int32_t  reward_curriculum_term[4]; /* GO2_REW_* index */
  float    reward_curriculum[4][4]; /* start_iter,end_iter,start_value,end_value */

  /* ---- observations: legged_robot_config.py:214-234 ---- */
  float    obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
  int32_t  add_noise; float noise_level;
  float    noise_dof_pos, noise_dof_vel, noise_lin_vel, noise_ang_vel, noise_gravity, noise_height;

  float    episode_length_s;        /* 25 -> max_episode_length = ceil(25/0.02) = 1250 */
  int32_t  send_timeouts;
  int32_t  num_steps_per_env;       /* 24: hard-coded in the env (:58), drives every curriculum */
} Go2SimCfg;

/* Raw views of the library-owned buffers (device pointers for the HIP library, host pointers for the
 * oracle).  Shapes are row-major as the reference's torch tensors see them. */
typedef struct Go2SimBuffers {
  /* the four Isaac Gym state tensors (legged_robot.py:779-787) */
  float*   root_states;        /* [N,13] */
  float*   dof_state;          /* [N,12,2]  (pos, vel) */
  float*   contact_forces;     /* [N,19,3]  world frame, last substep */
  float*   rigid_body_states;  /* [N,19,13] pos(3) quat(4) lin(3) ang(3) */
  /* VecEnv buffers (base_task.py:41-49) */
  float*   obs_buf;            /* [N,45] */
  float*   privileged_obs_buf; /* [N,263] */
  float*   rew_buf;            /* [N] */
  uint8_t* reset_buf;          /* [N] bool */
  uint8_t* time_out_buf;       /* [N] bool */
  int64_t* episode_length_buf; /* [N]  (the runner overwrites it: on_policy_runner.py:118) */
  /* per-env state the reference keeps as attributes (legged_robot.py:805-859) */
  float*   torques;            /* [N,12] last substep */
  float*   actions;            /* [N,12] clipped */
  float*   last_actions;       /* [N,12] */
  float*   last_last_actions;  /* [N,12] */
  float*   last_dof_vel;       /* [N,12] */
  float*   last_root_vel;      /* [N,6] */
  float*   commands;           /* [N,4] (writable by the caller: play.py:54-60) */
  float*   commands_resampling_step; /* [N] */
  float*   commands_xy_accumulation; /* [N,2] */
  uint8_t* stop_heading;       /* [N] */
  uint8_t* last_is_limit_vel;  /* [N] */
  float*   base_lin_vel;       /* [N,3] */
  float*   base_ang_vel;       /* [N,3] */
  float*   projected_gravity;  /* [N,3] */
  float*   rpy;                /* [N,3] */
  float*   measured_heights;   /* [N,187] */
  float*   max_move_distance;  /* [N] */
  float*   turn_over_timer;    /* [N] */
  float*   feet_air_time;      /* [N,4] */
  uint8_t* last_contacts;      /* [N,4] */
  uint8_t* last_contacts2;     /* [N,4] */
  float*   motor_strengths;    /* [N,12] */
  float*   motor_zero_offsets; /* [N,12] */
  float*   p_gains_multiplier; /* [N,12] */
  float*   d_gains_multiplier; /* [N,12] */
  float*   env_origins;        /* [N,3] */
  int64_t* terrain_levels;     /* [N] */
  int64_t* terrain_types;      /* [N] */
  float*   episode_sums;       /* [GO2_NUM_REWARDS,N] */
  /* per-env physical DR fixed at creation (legged_robot.py:320-402) */
  float*   friction_coeffs;    /* [N] shape friction */
  float*   restitution_coeffs; /* [N] */
  float*   added_base_mass;    /* [N] */
  float*   added_base_com;     /* [N,3] */
  float*   link_mass_ratio;    /* [N,18] */
  /* extras["episode"] (legged_robot.py:229-242): mean over the envs reset in the latest step that
   * had >= 1 reset, divided by max_episode_length_s; then [GO2_NUM_REWARDS] the number of envs reset in that step, [+1, +2]
   * command_ranges['lin_vel_x'] (lo, hi) as update_command_curriculum keeps it (extras['episode']['max_command_x'] = hi, :241-242),
   * [+3] terrain_level_all = mean(terrain_levels) and [+4 + k] the mean level of the envs on terrain kind k (NaN if none), all as of that
   * same step (:231-237: written by reset_idx only; 0 / NaN on a plane).  GO2_EPISODE_INFO_LEN floats. */
  float*   episode_info;
  /* warm-start impulses of the 4 foot contacts */
  float*   foot_impulse;       /* [N,4,3] */
} Go2SimBuffers;

typedef struct Go2Sim Go2Sim;

/* 1 if this library computes on the GPU (product), 0 if it is the CPU oracle. */
int go2sim_is_device_library(void);
/* Memory layout of the per-env buffers of Go2SimBuffers: 0 = row-major as documented (oracle);
 * 1 = field-major / SoA (HIP library): logical [N,a,b] is stored in C order of the REVERSED dims, element
 * (e,i,j) at ((j*a + i)*N + e), so consecutive envs are consecutive addresses.  In both layouts
 * obs_buf [N,45], privileged_obs_buf [N,263] (the policy's GEMM inputs) and episode_sums [R,N] are
 * row-major exactly as documented. */
int go2sim_buffer_layout(void);
const char* go2sim_last_error(void);
/* Fill `cfg` with the task=go2 defaults (go2_config.py + legged_robot_config.py); plane terrain. */
void go2sim_default_cfg(Go2SimCfg* cfg);

int  go2sim_create(const Go2SimCfg* cfg, int device_id, Go2Sim** out);
void go2sim_destroy(Go2Sim* h);
int  go2sim_get_buffers(Go2Sim* h, Go2SimBuffers* out);

/* ---- fused fast path ------------------------------------------------------------------------- */
/* LeggedRobot.reset_idx(all envs) (base_task.py:82-84) without the following step. */
int  go2sim_reset_all(Go2Sim* h, void* stream);
/* LeggedRobot.reset_idx(env_ids) (legged_robot.py:180-245) called from OUTSIDE a step, for `count` env ids (int32, in the library's memory
 * space like `actions`; duplicates allowed, ids outside [0, num_envs) are ignored): domain-randomisation redraws, terrain curriculum,
 * _reset_dofs / _reset_root_states, the per-env buffers the reference clears, _resample_commands, the extras['episode'] means of the envs
 * reset by this call, episode sums zeroed.  Like the reference it leaves obs_buf / rew_buf / time_out_buf and the derived base
 * velocities of those envs as they were.  count == 0 is a no-op (:189-190).  (The per-env resets of a step happen inside go2sim_step.) */
int  go2sim_reset_idx(Go2Sim* h, const int32_t* env_ids, int32_t count, void* stream);
/* LeggedRobot.step (legged_robot.py:60-100): clip actions, `decimation` x {delay select, PD torque,
 * clip, strength, articulated-body substep with contact}, post_physics_step, clip observations.
 * `actions` is [N,12] in the library's memory space. */
int  go2sim_step(Go2Sim* h, const float* actions, void* stream);

/* go2sim_step with the bookkeeping of ONE policy step of the rollout loop fused in (rsl_rl/runners/on_policy_runner.py:135-153,
 * algorithms/ppo.py:104-114), so that a rollout step is { policy, this call } with no copy / store launches in between:
 *   obs_out / priv_out: where the step's observations go instead of buffers.obs_buf / privileged_obs_buf ([N,45] / [N,263] row-major; the
 *       next row of the rollout storage, which the policy then reads in place; NULL = the library's own buffers, whose content is
 *       unspecified after a call that redirects them);
 *   rewards_out [N] = rew_buf + gamma * values * time_out_buf (the time-out bootstrap, ppo.py:107-108; values [N] = the critic's values of
 *       the step's input observations; NULL values = no bootstrap);  dones_out [N] = reset_buf;
 *   episode_info_out [GO2_EPISODE_INFO_LEN]: a copy of buffers.episode_info as of this step (extras['episode'] ring slot).
 * Every pointer may be NULL.  go2sim_step(h, a, s) == go2sim_step_rollout(h, a, NULL, s). */
typedef struct Go2StepOutputs {
  float*   obs_out;
  float*   priv_out;
  const float* values;
  float*   rewards_out;
  uint8_t* dones_out;
  float*   episode_info_out;
  float    gamma;
} Go2StepOutputs;
int  go2sim_step_rollout(Go2Sim* h, const float* actions, const Go2StepOutputs* out, void* stream);

/* ---- fine-grained operations (the Isaac Gym tensor API the reference drives) ------------------- */
/* legged_robot.py:73-92: only the physics loop of step() (actions must be in buffers.actions). */
int  go2sim_simulate(Go2Sim* h, void* stream);
/* legged_robot.py:102-142: post_physics_step() on whatever the four state tensors currently hold. */
int  go2sim_post_physics(Go2Sim* h, void* stream);
/* gym.set_actor_root_state_tensor_indexed / set_dof_state_tensor_indexed (:632,:705,:722):
 * commit rows `ids` of the API tensors into the simulator's internal state. ids may be NULL = all. */
int  go2sim_set_root_state_indexed(Go2Sim* h, const int32_t* ids, int32_t count, void* stream);
int  go2sim_set_dof_state_indexed(Go2Sim* h, const int32_t* ids, int32_t count, void* stream);

/* ---- host-side scalars the reference keeps in Python ------------------------------------------ */
int     go2sim_set_common_step_counter(Go2Sim* h, int64_t value);  /* train.py:14 */
int64_t go2sim_get_common_step_counter(Go2Sim* h);
/* update_reward_curriculum(force_update) (:144-152) */
int  go2sim_update_reward_curriculum(Go2Sim* h, int force_update);
/* current reward_curriculum_scales / command_ranges / zero_command_proba, for inspection */
int  go2sim_get_curriculum_state(Go2Sim* h, float reward_curriculum_scale[GO2_NUM_REWARDS],
                                 float cmd_ranges[4][2], float* zero_command_proba);

/* A go2sim_step enqueue captured in a HIP graph advances the DEVICE-resident counters when replayed, but not the
 * library's host mirror (go2sim_get_common_step_counter, go2sim_peek_uniforms): tell it that `steps` captured
 * steps were replayed. */
int  go2sim_notify_replayed(Go2Sim* h, int32_t steps);

/* ---- test hooks ------------------------------------------------------------------------------- */
/* Use the caller's uniforms [N][GO2_NUM_UNIFORMS] for the NEXT step/reset only (NULL = Philox). */
int  go2sim_inject_uniforms(Go2Sim* h, const float* uniforms, void* stream);
/* Copy the Philox uniforms the next step would use into `out` [N][GO2_NUM_UNIFORMS]. */
int  go2sim_peek_uniforms(Go2Sim* h, float* out, void* stream);
/* LeggedRobot.step's torque loop (legged_robot.py:67-81, _compute_torques :594-618) on caller-supplied DOF states ("fake physics"):
 * clip the raw actions [N,12], draw the action delay (:71-78), and for each of the `decimation` substeps compute the PD torques from
 * dof [decimation][N][12][2] (row-major) -> out [decimation][N][12].  Leaves the clipped actions / the last substep's torques in the
 * library's buffers like step().  Lets the reference's golden torques be compared with each library's own arithmetic. */
int  go2sim_debug_torque_trace(Go2Sim* h, const float* actions_raw, const float* dof, float* out, void* stream);
/* The individually rounded fp32 operations the height-scan INDEX arithmetic is built from (csrc/go2_math.h go2_*_rn) over arrays:
 * out [6][n] = a*b, a+b, a-b, a/b, sqrt(|a|), a/b[0] (through the double-precision reciprocal) — each must equal the IEEE result. */
int  go2sim_debug_strict_ops(const float* a, const float* b, float* out, int32_t n, void* stream);
/* The simulator's sphere-vs-terrain contact query (what replaces PhysX's mesh / heightfield collision for the robot's collision spheres):
 * pts [n][4] = world centre x, y, z and radius -> out [n][4] = gap (< 0: penetration) and unit contact normal of the deepest contact
 * (facet under the centre; with hf_walls also the vertical faces of the trimesh). */
int  go2sim_debug_contact_query(Go2Sim* h, const float* pts, float* out, int32_t n, void* stream);
/* Measurement aid: a kernel with the step kernel's HBM access pattern (16 environments per 256-thread workgroup, field-major fields
 * [f][N], 4 bytes per environment and field) and a known byte count: reads `nread` fields of `in` [nread][N], writes `nwrite` fields of
 * `out` [nwrite][N].  Profiled with rocprofv3 FETCH_SIZE / WRITE_SIZE it calibrates those counters for this pattern (tools/pmc_pass.sh). */
int  go2sim_debug_traffic_probe(const float* in, float* out, int32_t N, int32_t nread, int32_t nwrite, void* stream);

/* ---- measurement ------------------------------------------------------------------------------ */
/* While enabled, every go2sim_step / go2sim_simulate brackets its main kernel with events ON THE STREAM IT
 * IS LAUNCHED ON; go2sim_kernel_time synchronises, returns the summed kernel time [ms] and the number of
 * launches since the last call, and resets both.  (The oracle reports host wall-clock.) */
int  go2sim_enable_timing(Go2Sim* h, int enable);
int  go2sim_kernel_time(Go2Sim* h, double* total_ms, int64_t* launches);

/* ---- PPO rollout kernels ---------------------------------------------------------------------- */
/* RolloutStorage.compute_returns (rollout_storage.py:123-137) for a [T,N] rollout:
 *   GAE(gamma, lam) reverse scan -> returns, raw advantages; also accumulates
 *   partials[3] = {sum adv, sum adv^2, count} (float64) for the caller to all-reduce across shards.
 * rewards/values/returns/advantages: float [T,N]; dones: uint8 [T,N]; last_values: float [N]. */
int  go2sim_gae(const float* rewards, const uint8_t* dones, const float* values, const float* last_values,
                float* returns, float* advantages, double* partials, int32_t T, int32_t N,
                float gamma, float lam, void* stream);
/* advantages = (advantages - mean) / (std + 1e-8) with mean/std (unbiased) from the (all-reduced)
 * partials (rollout_storage.py:137). */
int  go2sim_normalize_advantages(float* advantages, const double* partials, int32_t count, void* stream);

/* Fused PPO loss head (rsl_rl/rsl_rl/algorithms/ppo.py:131-170 + Normal.log_prob/entropy of modules/actor_critic.py:101-127):
 * per-sample log-prob, ratio, clipped surrogate, clipped value loss, entropy, KL(old || new), and the ANALYTIC gradients
 * of   loss = mean(surrogate) + value_coef * mean(value_loss) - entropy_coef * mean(entropy)
 * w.r.t. mu [B,A], the state-independent std [A] and value [B] in one pass (what autograd spreads over ~150 launches).
 * stats[5] = {surrogate_loss, value_loss, kl_mean, entropy_mean, loss}.  workspace: >= 24*ceil(B/64) floats.  A <= 16.
 * surrogate_split: 0 = plain PPO.  0 < split < B = the Concurrent-Teacher-Student surrogate (algorithms/cts.py:228-231):
 *   mean(surrogate[:split]) + mean(surrogate[split:]) — teacher rows first, student rows after; value loss / entropy / KL
 *   stay means over all B rows.
 * Deterministic: block partials are combined in a fixed order by a second tiny kernel. */
int  go2sim_ppo_loss(const float* mu, const float* std, const float* value, const float* actions, const float* old_mu,
                     const float* old_sigma, const float* old_log_prob, const float* advantages, const float* target_values,
                     const float* returns, float* grad_mu, float* grad_std, float* grad_value, float* stats, float* workspace,
                     int32_t B, int32_t A, float clip_param, float value_loss_coef, float entropy_coef,
                     int32_t use_clipped_value_loss, int32_t surrogate_split, void* stream);

/* Sampling head of PPO.act / CTS.act + its storage writes for ONE rollout step (rsl_rl/algorithms/ppo.py:90-102,
 * modules/actor_critic.py:101-127), replacing ~17 element-wise launches:
 *   a = mu + std * eps                         (Normal(mu, std).sample() with the noise made explicit)
 *   log_prob = sum_j [ -(a_j - mu_j)^2 / (2 std_j^2) - log std_j - log sqrt(2 pi) ]
 * mu, eps: float [N,A]; std: float [A] (state-independent); value: float [N].
 * Writes actions_out [N,A] (handed to env.step) and the rollout-storage rows of this step: actions_st, mu_st,
 * sigma_st [N,A] (sigma = std broadcast), log_prob_st [N], values_st [N].  Any *_st pointer may be NULL. */
int  go2sim_act_head(const float* mu, const float* std, const float* eps, const float* value, float* actions_out,
                     float* actions_st, float* mu_st, float* sigma_st, float* log_prob_st, float* values_st,
                     int32_t N, int32_t A, void* stream);
/* PPO.process_env_step (ppo.py:104-114): rewards_st = rewards + gamma * values_st * time_outs (bootstrap on time-outs; time_outs
 * may be NULL), dones_st = dones.  rewards/values_st/rewards_st: float [N]; dones/time_outs/dones_st: uint8 [N]. */
int  go2sim_store_transition(const float* rewards, const uint8_t* dones, const uint8_t* time_outs, const float* values_st,
                             float* rewards_st, uint8_t* dones_st, float gamma, int32_t N, void* stream);

/* Backward of a hidden layer's activation fused with its bias gradient (the policy MLPs are Linear -> ELU stacks,
 * modules/actor_critic.py:60-90): given the upstream gradient gy [B,C] and the layer OUTPUT y = elu(z) [B,C] (alpha = 1),
 *   gz = gy * (y > 0 ? 1 : y + 1)            (= torch's elu_backward with is_result=True)
 *   gb[c] = sum_b gz[b,c]                    (the Linear's bias gradient; deterministic two-stage reduction)
 * in one pass over the activations instead of two.  workspace: >= C * ceil(B/64) floats.  gz may alias gy. */
int  go2sim_elu_backward_bias(const float* gy, const float* y, float* gz, float* gb, float* workspace, int32_t B, int32_t C, void* stream);

/* The tail of one mini-batch step of PPO.update / CTS.update (rsl_rl/algorithms/ppo.py:140-155,178-181; cts.py:207-286) over a LIST of
 * parameter tensors, in two launches instead of the ~30 element-wise / multi-tensor launches of the eager formulation:
 *   if kl_mean: lr = kl > 2 d ? max(1e-5, lr / 1.5) : (kl < d / 2 && kl > 0 ? min(1e-2, lr * 1.5) : lr)         (:140-155; lr is a device scalar, in place)
 *   coef = min(1, max_norm / (|| all grads ||_2 + 1e-6))                                                         (torch.nn.utils.clip_grad_norm_, :180)
 *   per element, torch.optim.Adam (amsgrad off, weight_decay 0):  g = coef grad;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
 *      t = step + 1;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps);  every step[i] += 1                (:181)
 * The gradient tensors are read, not scaled in place.  The norm is a fixed-order two-stage reduction (deterministic).
 * Go2AdamTensors is a HOST struct of DEVICE pointers (it travels as a kernel argument: a captured launch keeps the addresses it was
 * recorded with).  workspace: >= go2sim_adam_workspace_len(t) floats of device memory. */
#define GO2_ADAM_MAX_TENSORS 48
#define GO2_ADAM_CHUNK 4096
typedef struct Go2AdamTensors {
  int32_t  count;
  int32_t  numel[GO2_ADAM_MAX_TENSORS];
  float*   param[GO2_ADAM_MAX_TENSORS];
  const float* grad[GO2_ADAM_MAX_TENSORS];
  float*   exp_avg[GO2_ADAM_MAX_TENSORS];
  float*   exp_avg_sq[GO2_ADAM_MAX_TENSORS];
  float*   step[GO2_ADAM_MAX_TENSORS];       /* torch keeps one float step counter per parameter; all equal */
} Go2AdamTensors;
int  go2sim_adam_workspace_len(const Go2AdamTensors* t);
int  go2sim_adam_clip_step(const Go2AdamTensors* t, float* lr, const float* kl_mean, float desired_kl, float max_grad_norm,
                           double beta1, double beta2, double eps, float* workspace, void* stream);   /* (doubles, as torch holds them: 1 - beta is formed in fp64) */

/* Observation-history ring of the CTS runner (on_policy_runner_cts.py:155-156), in place:
 *   history[dones > 0] = 0;  history = cat(history[:, 1:], obs[:, None])      history: float [N,H,D], obs: float [N,D],
 * dones: uint8 [N] or NULL (no zeroing: the push before the first step, :129). */
int  go2sim_history_push(float* history, const float* obs, const uint8_t* dones, int32_t N, int32_t H, int32_t D, void* stream);

/* The head of PPO.update: ONE permutation of the rollout for all epochs and the gathers of the storage tensors into mini-batch order
 * (rsl_rl/rsl_rl/storage/rollout_storage.py:147-183: torch.randperm, then obs[b], critic_obs[b], actions[b], ... per mini-batch), as ONE launch instead of
 * a 12-kernel radix sort and one gather per tensor:   dst_j[r, :] = src_j[pi(r), :]   for every job j (rows of row_floats floats, dense) and r < rows.
 *   indices != NULL: pi = the given permutation (int64 [rows]; how the tests replay the reference's torch.randperm draw).
 *   indices == NULL: pi = go2sim_shuffle_index(., rows, key_state[0], key_state[1]) — a keyed pseudo-random BIJECTION of [0, rows): 6 Feistel rounds over the
 *     next even power of two with cycle walking (no sort, no scratch: every output row computes its own source row) — and key_state[1] is advanced by one when
 *     the launch is over, so a replayed HIP graph draws a new permutation every time.  key_state: device uint32 [4] = {seed, counter, internal ticket, 0}.
 * clear / nclear: floats set to zero by the launch (the update's loss accumulators), or NULL.  Up to GO2_GATHER_MAX_JOBS jobs.
 * ABI 8: dst_pitch = floats between two destination rows (0: dense = row_floats).  A wider pitch lets the gathered rows land in a column block of a wider matrix:
 * the CTS update gathers obs / privileged obs straight into columns [32, 32 + 45) / [32, 32 + 263) of the actor's / critic's [latent | obs] input matrices
 * (rsl_rl/rsl_rl/modules/actor_critic_cts.py:146-151,168-176 builds them with torch.cat per mini-batch); `dst` then points at the block's first column. */
#define GO2_GATHER_MAX_JOBS 12
typedef struct Go2GatherJob { const float* src; float* dst; int32_t row_floats; int32_t dst_pitch; } Go2GatherJob;
int  go2sim_shuffle_gather(const Go2GatherJob* jobs, int32_t njobs, int32_t rows, const int64_t* indices, uint32_t* key_state, float* clear, int32_t nclear, void* stream);
/* host-callable statement of the permutation (both libraries; no device work): pi(i) for i < n under (seed, counter) */
uint32_t go2sim_shuffle_index(uint32_t i, uint32_t n, uint32_t seed, uint32_t counter);
/* ABI 8: the index list of a CTS update (rsl_rl/rsl_rl/storage/rollout_storage_cts.py:152-160: teacher and student samples are shuffled SEPARATELY — two torch.randperm —
 * and mini-batch i is [teacher chunk i | student chunk i]), as one launch instead of two radix sorts and ~10 slicing / cat / index launches:
 *   out[i (tb + sb) + j] = map[ j < tb ? pi_t(i tb + j) : nt + pi_s(i sb + j - tb) ]      i < nmb, tb = nt / nmb, sb = ns / nmb
 * pi_t = go2sim_shuffle_index(., nt, key_state[0], key_state[1]), pi_s = go2sim_shuffle_index(., ns, key_state[0] ^ GO2_SHUFFLE_TAIL_SEED, key_state[1]); key_state[1]
 * advances by one when the launch is over (a replayed HIP graph draws new permutations every time).  map: int64 [nt + ns] (the storage position of sample k of the
 * reference's teacher-first layout) or NULL (identity).  out: int64 [nmb (tb + sb)], what go2sim_shuffle_gather takes as `indices`. */
#define GO2_SHUFFLE_TAIL_SEED 0x9E3779B9u
int  go2sim_cts_minibatch_indices(int64_t* out, int32_t nmb, int32_t nt, int32_t ns, const int64_t* map, uint32_t* key_state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GO2SIM_H */
