"""task=go2 configuration: restates legged_gym/envs/go2/go2_config.py:4-217 on top of the base config, plus
`GO2FlatCfg` — the same robot on `mesh_type='plane'`, which is how the reference spells "flat"
(legged_gym/envs/go2/go2_config_fast_flat_move.py:98; the BASELINE workload "task=go2 flat terrain")."""
import math

from ..base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgACMoECTS, LeggedRobotCfgCTS, LeggedRobotCfgDualMoECTS, LeggedRobotCfgMCPCTS, LeggedRobotCfgMoECTS, LeggedRobotCfgMoENGCTS, LeggedRobotCfgPPO, _max_cmd_table

_LEGS = ("FL", "FR", "RL", "RR")


class GO2Cfg(LeggedRobotCfg):
    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.42]
        default_joint_angles = dict([(f"{l}_hip_joint", 0.1 if l[1] == "L" else -0.1) for l in _LEGS]
                                    + [(f"{l}_thigh_joint", 0.8 if l[0] == "F" else 1.0) for l in _LEGS]
                                    + [(f"{l}_calf_joint", -1.5) for l in _LEGS])
        turn_over = False
        turn_over_proportions = [0.0, 0.2, 0.8]
        turn_over_init_heights = {"backflip": [0.10, 0.15], "sideflip": [0.16, 0.21]}

    class env(LeggedRobotCfg.env):
        num_envs = 8192
        num_observations = 45
        num_privileged_obs = 45 + 3 + 4 + 12 + 12 + 187      # 263
        episode_length_s = 25

    class domain_rand(LeggedRobotCfg.domain_rand):
        randomize_friction = True
        friction_range = [0.0, 2.0]
        randomize_base_mass = True
        added_mass_range = [-1.0, 1.0]
        randomize_link_mass = True
        multiplied_link_mass_range = [0.9, 1.1]
        randomize_base_com = True
        added_base_com_range = [-0.03, 0.03]
        randomize_restitution = True
        restitution_range = [0.0, 0.5]
        randomize_pd_gains = True
        stiffness_multiplier_range = [0.9, 1.1]
        damping_multiplier_range = [0.9, 1.1]
        randomize_motor_zero_offset = True
        motor_zero_offset_range = [-0.035, 0.035]
        randomize_motor_strength = True
        motor_strength_range = [0.8, 1.2]
        push_robots = True
        push_interval_s = 4
        max_push_vel_xy = 0.4
        max_push_ang_vel = 0.6
        randomize_action_delay = True

    class control(LeggedRobotCfg.control):
        control_type = "P"
        stiffness = {"joint": 20.0}
        damping = {"joint": 0.5}
        action_scale = 0.25
        decimation = 4

    class terrain(LeggedRobotCfg.terrain):
        max_init_terrain_level = 5
        terrain_proportions = [0.05, 0.20, 0.05, 0.25, 0.10, 0.20, 0.0, 0.0, 0.15]
        move_down_by_accumulated_xy_command = True

    class commands(LeggedRobotCfg.commands):
        curriculum = False
        max_curriculum = 1.0
        num_commands = 4
        resampling_time = 5.0
        heading_command = False
        zero_command_curriculum = {"start_iter": 0, "end_iter": 1500, "start_value": 0.0, "end_value": 0.1}
        limit_ang_vel_at_zero_command_prob = 0.2
        limit_vel_prob = 0.2
        limit_vel_invert_when_continuous = True
        limit_vel = {"lin_vel_x": [-1, 1], "lin_vel_y": [-1, 1], "ang_vel_yaw": [-1, 0, 1]}
        stop_heading_at_limit = True
        dynamic_resample_commands = True
        command_range_curriculum = [
            {"iter": 20000, "lin_vel_x": [-1.0, 1.0], "lin_vel_y": [-1.0, 1.0], "ang_vel_yaw": [-1.5, 1.5], "heading": [-1.57, 1.57]},
            {"iter": 50000, "lin_vel_x": [-2.0, 2.0], "lin_vel_y": [-1.0, 1.0], "ang_vel_yaw": [-2.0, 2.0], "heading": [-1.57, 1.57]},
        ]
        turn_over_zero_time = {"backflip": 5.0, "sideflip": 3.0}
        terrain_max_command_ranges = _max_cmd_table([1.5, 1.5, 1.5, 1, 1, 1, 1, 1, 2.0], [1.0] * 9, [1.5] * 8 + [2.0])

        class ranges:
            lin_vel_x = [-0.5, 0.5]
            lin_vel_y = [-0.5, 0.5]
            ang_vel_yaw = [-1.0, 1.0]
            heading = [-1.57, 1.57]

    class asset(LeggedRobotCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/go2/urdf/go2.urdf"   # the tables in include/go2_model_data.h are generated from it
        name = "go2"
        foot_name = "foot"
        penalize_contacts_on = ["thigh", "calf"]
        terminate_after_contacts_on = ["base"]
        self_collisions = 1

    class rewards(LeggedRobotCfg.rewards):
        soft_dof_pos_limit = 0.9
        base_height_target = 0.38
        only_positive_rewards = False
        max_contact_force = 147.0
        curriculum_rewards = [
            {"reward_name": "lin_vel_z", "start_iter": 0, "end_iter": 1500, "start_value": 1.0, "end_value": 0.0},
            {"reward_name": "correct_base_height", "start_iter": 0, "end_iter": 5000, "start_value": 1.0, "end_value": 10.0},
        ]
        tracking_sigma = 0.25
        dynamic_sigma = {"min_lin_vel": 0.5, "max_lin_vel": 1.5, "min_ang_vel": 1.0, "max_ang_vel": 2.0,
                         "max_sigma": [5 / 12, 1 / 4, 1 / 4, 1 / 2, 1 / 2, 3 / 4, 1, 1, 1 / 4]}
        min_legs_distance = 0.1
        turn_over_roll_threshold = math.pi / 4

        class scales:                      # redefined without a base on purpose (go2_config.py:178): nothing inherited
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            dof_acc = -2.5e-7
            dof_power = -2e-5
            torques = -1e-4
            correct_base_height = -1.0
            action_rate = -0.01
            action_smoothness = -0.01
            collision = -1.0
            dof_pos_limits = -2.0
            feet_regulation = -0.05
            hip_to_default = -0.05

        class turn_over_scales:
            upright = 1.0

    class noise(LeggedRobotCfg.noise):
        add_noise = True


class GO2FlatCfg(GO2Cfg):
    """task=go2_flat: GO2Cfg on a ground plane (no terrain object, no terrain curriculum)."""
    class terrain(GO2Cfg.terrain):
        mesh_type = "plane"
        curriculum = False


class GO2CfgPPO(LeggedRobotCfgPPO):
    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        run_name = ""
        experiment_name = "go2_ppo"
        max_iterations = 150000
        save_interval = 500


class GO2FlatCfgPPO(GO2CfgPPO):
    class runner(GO2CfgPPO.runner):
        experiment_name = "go2_flat_ppo"


class GO2CfgCTS(LeggedRobotCfgCTS):                 # go2_config.py:219-229
    class runner(LeggedRobotCfgCTS.runner):
        num_steps_per_env = 24
        run_name = ""
        experiment_name = "go2_cts"
        max_iterations = 150000
        save_interval = 500

    class policy(LeggedRobotCfgCTS.policy):
        latent_dim = 32
        norm_type = "l2norm"


class GO2CfgMoECTS(LeggedRobotCfgMoECTS):           # go2_config.py:276-284
    class policy(LeggedRobotCfgMoECTS.policy):
        expert_num = 8

    class runner(LeggedRobotCfgMoECTS.runner):
        run_name = ""
        experiment_name = "go2_moe_cts"
        max_iterations = 150000
        save_interval = 500


class GO2FlatCfgCTS(GO2CfgCTS):
    class runner(GO2CfgCTS.runner):
        experiment_name = "go2_flat_cts"


class GO2FlatCfgMoECTS(GO2CfgMoECTS):
    class runner(GO2CfgMoECTS.runner):
        experiment_name = "go2_flat_moe_cts"


class GO2CfgMoENGCTS(LeggedRobotCfgMoENGCTS):       # go2_config.py:231-243
    class policy(LeggedRobotCfgMoENGCTS.policy):
        obs_no_goal_mask = [True] * 6 + [False] * 3 + [True] * 36      # observation without the command entries
        student_expert_num = 8

    class algorithm(LeggedRobotCfgMoENGCTS.algorithm):
        load_balance_coef = 0.01

    class runner(LeggedRobotCfgMoENGCTS.runner):
        run_name = ""
        experiment_name = "go2_moe_no_goal_cts"
        max_iterations = 150000
        save_interval = 500


class GO2CfgACMoECTS(LeggedRobotCfgACMoECTS):       # go2_config.py:256-264
    class policy(LeggedRobotCfgACMoECTS.policy):
        expert_num = 8

    class runner(LeggedRobotCfgACMoECTS.runner):
        run_name = ""
        experiment_name = "go2_ac_moe_cts"
        max_iterations = 150000
        save_interval = 500


class GO2CfgDualMoECTS(LeggedRobotCfgDualMoECTS):   # go2_config.py:266-274
    class policy(LeggedRobotCfgDualMoECTS.policy):
        expert_num = 8

    class runner(LeggedRobotCfgDualMoECTS.runner):
        run_name = ""
        experiment_name = "go2_dual_moe_cts"
        max_iterations = 150000
        save_interval = 500


class GO2CfgMCPCTS(LeggedRobotCfgMCPCTS):           # go2_config.py:245-254
    class policy(LeggedRobotCfgMCPCTS.policy):
        obs_no_goal_mask = [True] * 6 + [False] * 3 + [True] * 36
        student_expert_num = 8

    class runner(LeggedRobotCfgMCPCTS.runner):
        run_name = ""
        experiment_name = "go2_mcp_cts"
        max_iterations = 150000
        save_interval = 500
