/* go2sim_defaults.h — the task=go2 configuration as numbers.
 *
 * Values restate legged_gym/envs/go2/go2_config.py:4-208 over the base
 * legged_gym/envs/base/legged_robot_config.py:4-259 (each line cited below).  Shared by the HIP
 * library and the oracle because it is data, not algorithm.  terrain_mode defaults to plane: the
 * BASELINE "go2 flat" workload (go2_config_fast_flat_move.py:98 is how the reference spells flat).
 */
#ifndef GO2SIM_DEFAULTS_H
#define GO2SIM_DEFAULTS_H
#include <string.h>
#include "go2sim.h"

static inline void go2sim_fill_default_cfg(Go2SimCfg* c) {
  memset(c, 0, sizeof(*c));
  c->struct_size = (uint32_t)sizeof(Go2SimCfg);
  c->abi_version = GO2SIM_ABI_VERSION;
  c->num_envs = 4096; c->env_offset = 0; c->num_envs_global = 4096;
  c->seed = 1;                                   /* legged_robot_config.py:262 */
  c->sim_dt = 0.005f;                            /* :243 */
  c->decimation = 4;                             /* go2_config.py:85 */
  c->gravity[0] = 0.f; c->gravity[1] = 0.f; c->gravity[2] = -9.81f; /* :245 */
  c->solver_iterations = 8;                      /* 2 x physx.num_position_iterations (:251): sweeps of the leg-parallel iteration per substep; 4..16 give the same gait/stance, the
                                                    body forces' distance to the converged solve is what moves (profiles/r6_solver_convergence.txt) */
  c->contact_offset = 0.01f;                     /* :253 */
  c->erp = 0.5f;                                 /* own choice */
  c->max_depenetration_velocity = 1.0f;          /* :256 */
  c->bounce_threshold_velocity = 0.5f;           /* :255 */
  c->contact_cfm = 1e-3f;                        /* own choice */
  c->joint_armature = 0.f;                       /* :133 */
  c->max_linear_velocity = 1000.f;               /* :132 */
  c->max_angular_velocity = 1000.f;              /* :131 */
  c->joint_limit_margin = 0.05f;                 /* own choice */
  c->terrain_mode = 0;
  c->terrain_friction = 1.0f;                    /* :21-22 */
  c->terrain_restitution = 0.f;                  /* :23 */
  c->hf_hscale = 0.1f; c->hf_vscale = 0.005f; c->hf_border = 25.f; /* :17-19 */
  c->terrain_num_levels = 10; c->terrain_num_types = 20;           /* :33-34 */
  c->terrain_curriculum = 1;                     /* :20 */
  c->max_init_terrain_level = 5;                 /* go2_config.py:88 */
  c->move_down_by_accumulated_xy_command = 1;    /* go2_config.py:96 */
  c->terrain_length = 8.f;                       /* :31 */
  c->env_spacing = 3.f;                          /* :10 */
  c->measure_heights = 1;                        /* :25 */
  {
    /* go2_config.py:7-22 in DOF order FL,FR,RL,RR x hip,thigh,calf */
    static const float q0[12] = {0.1f, 0.8f, -1.5f, -0.1f, 0.8f, -1.5f, 0.1f, 1.0f, -1.5f, -0.1f, 1.0f, -1.5f};
    for (int i = 0; i < 12; ++i) { c->kp[i] = 20.f; c->kd[i] = 0.5f; c->default_dof_pos[i] = q0[i]; } /* go2_config.py:80-81 */
  }
  c->action_scale = 0.25f;                       /* go2_config.py:83 */
  c->control_type = 0;                           /* 'P', go2_config.py:78 */
  c->clip_actions = 100.f; c->clip_observations = 100.f; /* :222-223 */
  {
    static const float s0[13] = {0.f, 0.f, 0.42f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; /* go2_config.py:6, :91-93 */
    memcpy(c->base_init_state, s0, sizeof(s0));
  }
  c->randomize_friction = 1;      c->friction_range[0] = 0.f;   c->friction_range[1] = 2.f;    /* go2_config.py:43-44 */
  c->randomize_restitution = 1;   c->restitution_range[0] = 0.f; c->restitution_range[1] = 0.5f; /* :55-56 */
  c->randomize_base_mass = 1;     c->added_mass_range[0] = -1.f; c->added_mass_range[1] = 1.f;   /* :46-47 */
  c->randomize_link_mass = 1;     c->link_mass_range[0] = 0.9f;  c->link_mass_range[1] = 1.1f;   /* :49-50 */
  c->randomize_base_com = 1;      c->base_com_range[0] = -0.03f; c->base_com_range[1] = 0.03f;   /* :52-53 */
  c->randomize_pd_gains = 1;      c->stiffness_mult_range[0] = 0.9f; c->stiffness_mult_range[1] = 1.1f;
                                  c->damping_mult_range[0] = 0.9f;   c->damping_mult_range[1] = 1.1f;  /* :59-61 */
  c->randomize_motor_zero_offset = 1; c->motor_zero_offset_range[0] = -0.035f; c->motor_zero_offset_range[1] = 0.035f; /* :63-64 */
  c->randomize_motor_strength = 1;    c->motor_strength_range[0] = 0.8f; c->motor_strength_range[1] = 1.2f;           /* :66-67 */
  c->push_robots = 1; c->push_interval = 200; c->max_push_vel_xy = 0.4f; c->max_push_ang_vel = 0.6f; /* :70-73, legged_robot.py:1106 */
  c->randomize_action_delay = 1;                 /* :75 */
  c->cmd_resampling_time = 5.f;                  /* :102 */
  c->heading_command = 0;                        /* :103 */
  c->dynamic_resample_commands = 1;              /* :111 */
  c->limit_vel_prob = 0.2f;                      /* :107 */
  c->limit_vel_invert_when_continuous = 1;       /* :108 */
  c->stop_heading_at_limit = 1;                  /* :110 */
  c->limit_ang_vel_at_zero_command_prob = 0.2f;  /* :106 */
  c->cmd_tracking_curriculum = 0; c->cmd_max_curriculum = 1.f;   /* go2_config.py:99-100 */
  {
    /* itertools.product([-1,1],[-1,1],[-1,0,1]) (legged_robot.py:827-831, go2_config.py:109) */
    int n = 0;
    for (int a = -1; a <= 1; a += 2) for (int b = -1; b <= 1; b += 2) for (int d = -1; d <= 1; ++d) {
      c->limit_vel_comb[n][0] = (float)a; c->limit_vel_comb[n][1] = (float)b; c->limit_vel_comb[n][2] = (float)d; ++n;
    }
    c->limit_vel_comb_count = n;
  }
  c->zero_cmd_curriculum_enabled = 1;            /* go2_config.py:105 */
  c->zero_cmd_curriculum[0] = 0.f; c->zero_cmd_curriculum[1] = 1500.f; c->zero_cmd_curriculum[2] = 0.f; c->zero_cmd_curriculum[3] = 0.1f;
  c->cmd_ranges[0][0] = -0.5f; c->cmd_ranges[0][1] = 0.5f;   /* go2_config.py:142-146 */
  c->cmd_ranges[1][0] = -0.5f; c->cmd_ranges[1][1] = 0.5f;
  c->cmd_ranges[2][0] = -1.0f; c->cmd_ranges[2][1] = 1.0f;
  c->cmd_ranges[3][0] = -1.57f; c->cmd_ranges[3][1] = 1.57f;
  c->cmd_curriculum_count = 2;                   /* go2_config.py:112-124 */
  {
    static const float cc[2][9] = {{20000.f, -1.f, 1.f, -1.f, 1.f, -1.5f, 1.5f, -1.57f, 1.57f},
                                   {50000.f, -2.f, 2.f, -1.f, 1.f, -2.0f, 2.0f, -1.57f, 1.57f}};
    memcpy(c->cmd_curriculum, cc, sizeof(cc));
  }
  {
    /* go2_config.py:129-139: wave, slope, rough slope, stairs up, stairs down, obstacles, stepping stones, gap, flat */
    static const float lx[9] = {1.5f, 1.5f, 1.5f, 1.f, 1.f, 1.f, 1.f, 1.f, 2.f};
    static const float yw[9] = {1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 2.f};
    for (int k = 0; k < 9; ++k) {
      c->terrain_max_cmd_ranges[k][0][0] = -lx[k]; c->terrain_max_cmd_ranges[k][0][1] = lx[k];
      c->terrain_max_cmd_ranges[k][1][0] = -1.f;   c->terrain_max_cmd_ranges[k][1][1] = 1.f;
      c->terrain_max_cmd_ranges[k][2][0] = -yw[k]; c->terrain_max_cmd_ranges[k][2][1] = yw[k];
      c->terrain_max_cmd_ranges[k][3][0] = -1.57f; c->terrain_max_cmd_ranges[k][3][1] = 1.57f;
    }
  }
  /* go2_config.py:178-194 (the scales class is redefined, so nothing is inherited) */
  c->reward_scales[GO2_REW_TRACKING_LIN_VEL] = 1.0f;
  c->reward_scales[GO2_REW_TRACKING_ANG_VEL] = 0.5f;
  c->reward_scales[GO2_REW_LIN_VEL_Z] = -2.0f;
  c->reward_scales[GO2_REW_ANG_VEL_XY] = -0.05f;
  c->reward_scales[GO2_REW_DOF_ACC] = -2.5e-7f;
  c->reward_scales[GO2_REW_DOF_POWER] = -2e-5f;
  c->reward_scales[GO2_REW_TORQUES] = -1e-4f;
  c->reward_scales[GO2_REW_CORRECT_BASE_HEIGHT] = -1.0f;
  c->reward_scales[GO2_REW_ACTION_RATE] = -0.01f;
  c->reward_scales[GO2_REW_ACTION_SMOOTHNESS] = -0.01f;
  c->reward_scales[GO2_REW_COLLISION] = -1.0f;
  c->reward_scales[GO2_REW_DOF_POS_LIMITS] = -2.0f;
  c->reward_scales[GO2_REW_FEET_REGULATION] = -0.05f;
  c->reward_scales[GO2_REW_HIP_TO_DEFAULT] = -0.05f;
  c->only_positive_rewards = 0;                  /* go2_config.py:159 */
  c->tracking_sigma = 0.25f;                     /* :167 */
  c->dynamic_sigma_enabled = 1;                  /* :168-176 (inert on a plane: legged_robot.py:1303-1304) */
  c->dynamic_sigma_vel[0] = 0.5f; c->dynamic_sigma_vel[1] = 1.5f; c->dynamic_sigma_vel[2] = 1.0f; c->dynamic_sigma_vel[3] = 2.0f;
  {
    static const float ms[9] = {5.f / 12.f, 0.25f, 0.25f, 0.5f, 0.5f, 0.75f, 1.f, 1.f, 0.25f};
    memcpy(c->dynamic_sigma_max, ms, sizeof(ms));
  }
  c->soft_dof_pos_limit = 0.9f;                  /* :157 */
  c->soft_dof_vel_limit = 1.f; c->soft_torque_limit = 1.f; /* legged_robot_config.py:196-197 */
  c->base_height_target = 0.38f;                 /* :158 */
  c->max_contact_force = 147.f;                  /* :160 */
  c->min_legs_distance = 0.1f;                   /* :177 */
  c->turn_over = 0;                                /* go2_config.py:23 */
  c->turn_over_proportions[0] = 0.0f; c->turn_over_proportions[1] = 0.2f; c->turn_over_proportions[2] = 0.8f;     /* :25 */
  c->turn_over_init_heights[0][0] = 0.10f; c->turn_over_init_heights[0][1] = 0.15f;                                /* :26-29 */
  c->turn_over_init_heights[1][0] = 0.16f; c->turn_over_init_heights[1][1] = 0.21f;
  c->turn_over_zero_time[0] = 5.0f; c->turn_over_zero_time[1] = 3.0f;                                              /* :125-128 */
  c->turn_over_roll_threshold = 0.78539816339744830962f;                                                           /* :199 */
  for (int i = 0; i < GO2_NUM_REWARDS; ++i) c->turn_over_scales[i] = 0.0f;
  c->turn_over_scales[GO2_REW_UPRIGHT] = 1.0f;                                                                     /* :200-201 */
  c->reward_curriculum_count = 2;                /* :161-166 */
  c->reward_curriculum_term[0] = GO2_REW_LIN_VEL_Z;
  c->reward_curriculum[0][0] = 0.f; c->reward_curriculum[0][1] = 1500.f; c->reward_curriculum[0][2] = 1.f; c->reward_curriculum[0][3] = 0.f;
  c->reward_curriculum_term[1] = GO2_REW_CORRECT_BASE_HEIGHT;
  c->reward_curriculum[1][0] = 0.f; c->reward_curriculum[1][1] = 5000.f; c->reward_curriculum[1][2] = 1.f; c->reward_curriculum[1][3] = 10.f;
  c->obs_scale_lin_vel = 2.f; c->obs_scale_ang_vel = 0.25f; c->obs_scale_dof_pos = 1.f;
  c->obs_scale_dof_vel = 0.05f; c->obs_scale_height = 2.5f;  /* legged_robot_config.py:216-221 */
  c->add_noise = 1; c->noise_level = 1.f;        /* :226-227 */
  c->noise_dof_pos = 0.01f; c->noise_dof_vel = 1.5f; c->noise_lin_vel = 0.1f;
  c->noise_ang_vel = 0.2f; c->noise_gravity = 0.05f; c->noise_height = 0.1f; /* :229-234 */
  c->episode_length_s = 25.f;                    /* go2_config.py:39 */
  c->send_timeouts = 1;                          /* :11 */
  c->num_steps_per_env = 24;                     /* legged_robot.py:58 */
}
#endif
