"""Base environment / training configuration.  The attribute tree and every value restate
legged_gym/envs/base/legged_robot_config.py (env :5-13, terrain :15-40, commands :42-89, init_state :91-105,
control :107-115, asset :117-135, domain_rand :137-175, rewards :177-212, normalization :214-223,
noise :225-234, sim :242-259, PPO :261-307, CTS :309-340): the values are the specification of the hot path.
"""
import math

from .base_config import BaseConfig

_TERRAIN_KINDS = ("wave", "slope", "rough slope", "stairs up", "stairs down", "obstacles", "stepping stones", "gap", "flat")


def _max_cmd_table(x, y, yaw):
    return [{"lin_vel_x": [-a, a], "lin_vel_y": [-b, b], "ang_vel_yaw": [-c, c], "heading": [-1.57, 1.57]} for a, b, c in zip(x, y, yaw)]


class LeggedRobotCfg(BaseConfig):
    class env:
        num_envs = 4096
        num_observations = 48
        num_privileged_obs = None
        num_actions = 12
        env_spacing = 3.0
        send_timeouts = True
        episode_length_s = 20
        test = False

    class terrain:
        mesh_type = "trimesh"            # none | plane | heightfield | trimesh
        horizontal_scale = 0.1
        vertical_scale = 0.005
        border_size = 25
        curriculum = True
        static_friction = 1.0
        dynamic_friction = 1.0
        restitution = 0.0
        measure_heights = True
        measured_points_x = [round(-0.8 + 0.1 * i, 1) for i in range(17)]
        measured_points_y = [round(-0.5 + 0.1 * i, 1) for i in range(11)]
        selected = False
        terrain_kwargs = None
        max_init_terrain_level = 5
        terrain_length = 8.0
        terrain_width = 8.0
        num_rows = 10
        num_cols = 20
        terrain_spacing = 0.5
        terrain_proportions = [0.1, 0.1, 0.1, 0.2, 0.2, 0.1, 0.1, 0.1, 0.0]   # order: _TERRAIN_KINDS
        slope_treshold = 0.75
        move_down_by_accumulated_xy_command = False

    class commands:
        curriculum = False
        max_curriculum = 1.0
        num_commands = 4
        resampling_time = 10.0
        heading_command = False
        zero_command_curriculum = None
        limit_ang_vel_at_zero_command_prob = 0.0
        limit_vel_prob = 0.0
        limit_vel_invert_when_continuous = True
        limit_vel = {"lin_vel_x": [-1, 1], "lin_vel_y": [-1, 1], "ang_vel_yaw": [-1, 0, 1]}
        stop_heading_at_limit = True
        dynamic_resample_commands = False
        command_range_curriculum = []
        turn_over_zero_time = {"backflip": 5.0, "sideflip": 3.0}
        terrain_max_command_ranges = _max_cmd_table([1.5, 1.5, 1.5, 1, 1, 1, 1, 1, 2.0], [1.5, 1.5, 1.5, 1, 1, 1, 1, 1, 1.5], [1.5] * 9)

        class ranges:
            lin_vel_x = [-1.0, 1.0]
            lin_vel_y = [-0.5, 0.5]
            ang_vel_yaw = [-1, 1]
            heading = [-3.14, 3.14]

    class init_state:
        pos = [0.0, 0.0, 1.0]
        rot = [0.0, 0.0, 0.0, 1.0]       # x, y, z, w
        lin_vel = [0.0, 0.0, 0.0]
        ang_vel = [0.0, 0.0, 0.0]
        default_joint_angles = {"joint_a": 0.0, "joint_b": 0.0}
        turn_over = False
        turn_over_proportions = [0.0, 0.2, 0.8]
        turn_over_init_heights = {"backflip": [0.10, 0.15], "sideflip": [0.16, 0.21]}

    class control:
        control_type = "P"
        stiffness = {"joint_a": 10.0, "joint_b": 15.0}
        damping = {"joint_a": 1.0, "joint_b": 1.5}
        action_scale = 0.5
        decimation = 4

    class asset:
        file = ""
        name = "legged_robot"
        foot_name = "None"
        penalize_contacts_on = []
        terminate_after_contacts_on = []
        disable_gravity = False
        collapse_fixed_joints = True
        fix_base_link = False
        default_dof_drive_mode = 3
        self_collisions = 0
        replace_cylinder_with_capsule = True
        flip_visual_attachments = True
        density = 0.001
        angular_damping = 0.0
        linear_damping = 0.0
        max_angular_velocity = 1000.0
        max_linear_velocity = 1000.0
        armature = 0.0
        thickness = 0.01

    class domain_rand:
        robot_properties_update = None
        randomize_friction = True
        friction_range = [0.2, 1.25]
        randomize_base_mass = True
        added_mass_range = [-1.0, 1.0]
        randomize_link_mass = True
        multiplied_link_mass_range = [0.9, 1.1]
        randomize_base_com = True
        added_base_com_range = [-0.03, 0.03]
        randomize_restitution = False
        restitution_range = [0.0, 0.2]
        randomize_pd_gains = True
        stiffness_multiplier_range = [0.9, 1.1]
        damping_multiplier_range = [0.9, 1.1]
        randomize_motor_zero_offset = True
        motor_zero_offset_range = [-0.035, 0.035]
        randomize_motor_strength = False
        motor_strength_range = [0.8, 1.2]
        push_robots = True
        push_interval_s = 4
        max_push_vel_xy = 0.4
        max_push_ang_vel = 0.6
        randomize_action_delay = False

    class rewards:
        class scales:
            termination = -0.0
            tracking_lin_vel = 1.0
            tracking_ang_vel = 0.5
            lin_vel_z = -2.0
            ang_vel_xy = -0.05
            orientation = -0.0
            torques = -0.00001
            dof_vel = -0.0
            dof_acc = -2.5e-7
            base_height = -0.0
            feet_air_time = 1.0
            collision = -1.0
            feet_stumble = -0.0
            action_rate = -0.01
            stand_still = -0.0

        class turn_over_scales:
            upright = 1.0

        only_positive_rewards = True
        tracking_sigma = 0.25
        soft_dof_pos_limit = 1.0
        soft_dof_vel_limit = 1.0
        soft_torque_limit = 1.0
        base_height_target = 1.0
        max_contact_force = 100.0
        curriculum_rewards = None
        dynamic_sigma = None
        turn_over_roll_threshold = math.pi / 4
        min_legs_distance = 0.1

    class normalization:
        class obs_scales:
            lin_vel = 2.0
            ang_vel = 0.25
            dof_pos = 1.0
            dof_vel = 0.05
            height_measurements = 2.5
        clip_observations = 100.0
        clip_actions = 100.0

    class noise:
        add_noise = True
        noise_level = 1.0

        class noise_scales:
            dof_pos = 0.01
            dof_vel = 1.5
            lin_vel = 0.1
            ang_vel = 0.2
            gravity = 0.05
            height_measurements = 0.1

    class viewer:
        ref_env = 0
        pos = [10, 0, 6]
        lookat = [11.0, 5, 3.0]

    class sim:
        dt = 0.005
        substeps = 1
        gravity = [0.0, 0.0, -9.81]
        up_axis = 1

        class physx:                       # PhysX knobs the reference sets (:248-259); see `solver` for ours
            num_threads = 10
            solver_type = 1
            num_position_iterations = 4
            num_velocity_iterations = 0
            contact_offset = 0.01
            rest_offset = 0.0
            bounce_threshold_velocity = 0.5
            max_depenetration_velocity = 1.0
            max_gpu_contact_pairs = 2 ** 23
            default_buffer_size_multiplier = 5
            contact_collection = 2

        class solver:                      # parameters of this build's own contact solver
            iterations = 8                 # sweeps per substep = 2 per physx.num_position_iteration (round 6).  Against this model's own converged solve the body forces after one
                                           # policy step are within p90 = 13 % at 4 sweeps, 9.6 % at 5, 6.6 % at 6, 5.2 % at 7, 3.8 % at 8 (profiles/r6_solver_convergence.txt); a sweep costs
                                           # 2.2 us of the step kernel on the plane, 3.7 us on rough terrain (profiles/r6_kbench*.txt)
            erp = 0.5
            cfm = 1e-3
            joint_limit_margin = 0.05


class LeggedRobotCfgPPO(BaseConfig):
    seed = 1
    runner_class_name = "OnPolicyRunner"

    class policy:
        init_noise_std = 1.0
        actor_hidden_dims = [512, 256, 128]
        critic_hidden_dims = [512, 256, 128]
        activation = "elu"

    class algorithm:
        value_loss_coef = 1.0
        use_clipped_value_loss = True
        clip_param = 0.2
        entropy_coef = 0.01
        num_learning_epochs = 5
        num_mini_batches = 4
        learning_rate = 1.0e-3
        schedule = "adaptive"
        gamma = 0.99
        lam = 0.95
        desired_kl = 0.01
        max_grad_norm = 1.0

    class runner:
        policy_class_name = "ActorCritic"
        algorithm_class_name = "PPO"
        num_steps_per_env = 24
        max_iterations = 1500
        save_interval = 50
        experiment_name = "test"
        run_name = ""
        resume = False
        load_run = -1
        checkpoint = -1
        resume_path = None

    class robogauge:
        enabled = False
        port = 9973


class LeggedRobotCfgCTS(BaseConfig):
    """Concurrent Teacher-Student training (legged_robot_config.py:309-359)."""
    seed = 0
    runner_class_name = "OnPolicyRunnerCTS"
    history_length = 5

    class policy:
        init_noise_std = 1.0
        actor_hidden_dims = [512, 256, 128]
        critic_hidden_dims = [512, 256, 128]
        teacher_encoder_hidden_dims = [512, 256]
        student_encoder_hidden_dims = [512, 256]
        activation = "elu"
        latent_dim = 32
        norm_type = "l2norm"

    class algorithm:
        value_loss_coef = 1.0
        use_clipped_value_loss = True
        clip_param = 0.2
        entropy_coef = 0.01
        num_learning_epochs = 5
        num_mini_batches = 4
        learning_rate = 1.0e-3
        student_encoder_learning_rate = 1.0e-3
        schedule = "adaptive"
        gamma = 0.99
        lam = 0.95
        desired_kl = 0.01
        max_grad_norm = 1.0
        teacher_env_ratio = 0.75

    class runner:
        policy_class_name = "ActorCriticCTS"
        algorithm_class_name = "CTS"
        num_steps_per_env = 24
        max_iterations = 1500
        save_interval = 50
        experiment_name = "test"
        run_name = ""
        resume = False
        load_run = -1
        checkpoint = -1
        resume_path = None

    class robogauge:
        enabled = False
        port = 9973


class LeggedRobotCfgMoECTS(LeggedRobotCfgCTS):
    """MoE student encoder (legged_robot_config.py:399-409)."""

    class policy(LeggedRobotCfgCTS.policy):
        expert_num = 8
        student_encoder_hidden_dims = [512, 256, 256]

    class algorithm(LeggedRobotCfgCTS.algorithm):
        load_balance_coef = 0.01

    class runner(LeggedRobotCfgCTS.runner):
        policy_class_name = "ActorCriticMoECTS"
        algorithm_class_name = "MoECTS"


class LeggedRobotCfgMoENGCTS(LeggedRobotCfgCTS):
    """MoE student encoder whose experts do not see the command (legged_robot_config.py:361-371)."""

    class policy(LeggedRobotCfgCTS.policy):
        obs_no_goal_mask = None
        student_expert_num = 8

    class algorithm(LeggedRobotCfgCTS.algorithm):
        load_balance_coef = 0.01

    class runner(LeggedRobotCfgCTS.runner):
        policy_class_name = "ActorCriticMoENGCTS"
        algorithm_class_name = "MoENGCTS"


class LeggedRobotCfgACMoECTS(LeggedRobotCfgCTS):
    """MoE actor + expert critic sharing the actor gate (legged_robot_config.py:382-388)."""

    class policy(LeggedRobotCfgCTS.policy):
        expert_num = 8

    class runner(LeggedRobotCfgCTS.runner):
        policy_class_name = "ActorCriticACMoECTS"
        algorithm_class_name = "ACMoECTS"


class LeggedRobotCfgDualMoECTS(LeggedRobotCfgCTS):
    """AC-MoE heads + MoE student encoder (legged_robot_config.py:390-397)."""

    class policy(LeggedRobotCfgCTS.policy):
        expert_num = 8
        student_encoder_hidden_dims = [512, 256, 256]

    class runner(LeggedRobotCfgCTS.runner):
        policy_class_name = "ActorCriticDualMoECTS"
        algorithm_class_name = "DualMoECTS"


class LeggedRobotCfgMCPCTS(LeggedRobotCfgCTS):
    """Multiplicative-compositional actor (legged_robot_config.py:373-380)."""

    class policy(LeggedRobotCfgCTS.policy):
        obs_no_goal_mask = None
        student_expert_num = 8

    class runner(LeggedRobotCfgCTS.runner):
        policy_class_name = "ActorCriticMCPCTS"
        algorithm_class_name = "MCPCTS"
