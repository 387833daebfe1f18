#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE in the build container.

CONTAINER-ONLY (reads /root/reference; the GPU box has no such path and never runs this).
What it pins: everything of the hot path that lives in the reference's own Python —
torque computation and action delay, post_physics_step (derived base quantities, command resampling,
termination, the reward terms, reset_idx, push, observations, last_* bookkeeping), and
RolloutStorage.compute_returns / one PPO.update.  The physics itself (Isaac Gym) is absent, so the four
simulator tensors are filled with SYNTHETIC states: the fixtures record those inputs, the per-env uniforms
injected at each torch.rand call site (oracle/fake_isaacgym.py) and every output the reference computed.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz + MANIFEST.json)
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import fake_isaacgym as fig  # noqa: E402

fig.install()
sys.path.insert(0, "/root/reference/rsl_rl")
sys.path.insert(0, "/root/reference")
import isaacgym  # noqa: E402,F401  (the stub; same import order as legged_gym/scripts/train.py:6-7)
from legged_gym.envs import *  # noqa: E402,F401,F403
from legged_gym.utils import task_registry  # noqa: E402
from legged_gym.utils.helpers import class_to_dict  # noqa: E402

# reward name -> index in the GO2_REW_* enum of include/go2sim.h
sys.path.insert(0, ROOT)
from go2_rl_gym_amd._abi import Abi  # noqa: E402

ABI = Abi()
NU = ABI.GO2_NUM_UNIFORMS


TERRAIN_SEED = 11


# "alt" sequence: the other branch of the booleans the registered go2 config leaves one way, and EVERY reward term switched on
# (the go2 config uses 14 of the 28).  Applied by dotted path to the reference's config here and, in the tests, to this build's
# config classes — so the host layer's config translation (LeggedRobot._fill_cfg) is part of what the fixture pins.
ALT_OVERRIDES = {
    "commands.heading_command": True, "commands.dynamic_resample_commands": False, "commands.limit_vel_invert_when_continuous": False,
    "commands.stop_heading_at_limit": False, "commands.resampling_time": 4.0,
    "rewards.only_positive_rewards": True, "rewards.tracking_sigma": 0.3, "rewards.base_height_target": 0.34, "rewards.max_contact_force": 60.0,
    "rewards.soft_dof_pos_limit": 0.85, "rewards.soft_dof_vel_limit": 0.3, "rewards.soft_torque_limit": 0.4, "rewards.min_legs_distance": 0.25,
    "rewards.scales.orientation": -0.3, "rewards.scales.base_height": -1.5, "rewards.scales.dof_vel": -1e-4, "rewards.scales.termination": -2.0,
    "rewards.scales.dof_vel_limits": -0.2, "rewards.scales.torque_limits": -0.01, "rewards.scales.feet_air_time": 1.0, "rewards.scales.stumble": -0.3,
    "rewards.scales.stand_still": -0.05, "rewards.scales.feet_contact_forces": -0.01, "rewards.scales.similar_to_default": -0.02,
    "rewards.scales.upright": 0.2, "rewards.scales.legs_distance": -1.5, "rewards.scales.x_command_hip_regular": -0.1,
    "domain_rand.randomize_motor_strength": False, "domain_rand.randomize_pd_gains": False, "domain_rand.randomize_action_delay": False,
    "domain_rand.push_interval_s": 3, "noise.noise_level": 0.5, "normalization.clip_observations": 5.0, "normalization.clip_actions": 3.0,
    "control.action_scale": 0.3,
}


def set_dotted(cfg, path, value):
    obj = cfg
    parts = path.split(".")
    for p_ in parts[:-1]:
        obj = getattr(obj, p_)
    setattr(obj, parts[-1], value)


def make_env(N, seed=1, mesh_type="plane", turn_over=False, overrides=None):
    env_cfg, train_cfg = task_registry.get_cfgs("go2")
    env_cfg.env.num_envs = N
    env_cfg.terrain.mesh_type = mesh_type
    env_cfg.init_state.turn_over = turn_over
    for k_, v_ in (overrides or {}).items():
        set_dotted(env_cfg, k_, v_)
    if turn_over:
        env_cfg.init_state.turn_over_proportions = [0.25, 0.35, 0.4]      # every branch of :654-684 gets exercised
        env_cfg.rewards.turn_over_scales.dof_power = -2e-5                  # a term present in both scale tables, besides `upright`
    torch.manual_seed(seed)
    np.random.seed(TERRAIN_SEED if mesh_type != "plane" else seed)
    fig.GYM.alloc(N)
    env = Go2Robot(env_cfg, fig.SimParams(), 1, "cpu", True)  # noqa: F405
    return env, env_cfg, train_cfg


def synth_state(rng, N, env, wide_roll=False):
    """One synthetic simulator state (the 'fake physics')."""
    root = np.zeros((N, 13), np.float32)
    root[:, 0:2] = env.env_origins[:, :2].numpy() + rng.uniform(-3, 3, (N, 2))
    root[:, 2] = env.env_origins[:, 2].numpy() + rng.uniform(0.2, 0.45, N)
    yaw = rng.uniform(-np.pi, np.pi, N); roll = rng.normal(0, 0.15, N); pitch = rng.normal(0, 0.15, N)
    if wide_roll:   # robots lying on their side / back: |roll| > turn_over_roll_threshold for about a third of the envs
        lying = rng.uniform(size=N) < 0.35
        roll = np.where(lying, rng.uniform(-np.pi, np.pi, N), roll)
    q = fig.quat_from_euler_xyz(torch.tensor(roll), torch.tensor(pitch), torch.tensor(yaw)).numpy()
    root[:, 3:7] = q
    root[:, 7:10] = rng.uniform(-1.2, 1.2, (N, 3))
    root[:, 10:13] = rng.uniform(-1.5, 1.5, (N, 3))
    q0 = env.default_dof_pos.numpy().reshape(1, 12)
    dof = np.zeros((4, N, 12, 2), np.float32)
    base = q0 * rng.uniform(0.4, 1.6, (N, 12))
    # push a few joints past their soft limits
    far = rng.uniform(size=(N, 12)) < 0.05
    base = np.where(far, q0 + rng.uniform(-2.0, 2.0, (N, 12)), base)
    for i in range(4):
        dof[i, :, :, 0] = base + 0.01 * i * rng.normal(size=(N, 12))
        dof[i, :, :, 1] = rng.uniform(-8, 8, (N, 12))
    contact = np.zeros((N, 19, 3), np.float32)
    feet = [6, 10, 14, 18]
    on = rng.uniform(size=(N, 4)) < 0.6
    contact[:, feet, 2] = on * rng.uniform(0, 120, (N, 4))
    contact[:, feet, 0:2] = on[..., None] * rng.normal(0, 15, (N, 4, 2))
    pen = [4, 5, 8, 9, 12, 13, 16, 17]
    hit = rng.uniform(size=(N, 8)) < 0.08
    contact[:, pen, :] = hit[..., None] * rng.normal(0, 5, (N, 8, 3))
    small = rng.uniform(size=(N, 8)) < 0.05
    contact[:, pen, :] += small[..., None] * rng.normal(0, 0.05, (N, 8, 3))
    base_hit = rng.uniform(size=N) < 0.04
    contact[:, 0, :] = base_hit[:, None] * rng.normal(0, 4, (N, 3))
    feet_state = np.zeros((N, 4, 13), np.float32)
    feet_state[:, :, 0:2] = root[:, None, 0:2] + rng.uniform(-0.3, 0.3, (N, 4, 2))
    feet_state[:, :, 2] = rng.uniform(0.0, 0.12, (N, 4))
    feet_state[:, :, 7:10] = rng.uniform(-2, 2, (N, 4, 3))
    return root, dof, contact.astype(np.float32), feet_state


def gen_env_sequence(N=16, T=64, seed=7, mesh_type="plane", turn_over=False, overrides=None):
    rng = np.random.default_rng(seed)
    # (recorded BEFORE the run: update_command_curriculum edits command_ranges['lin_vel_x'] in place, and after a command_range_curriculum
    #  stage has started that list IS the stage's config entry, legged_robot.py:438,736-737)
    overrides_json = json.dumps(overrides) if overrides else None
    env, env_cfg, train_cfg = make_env(N, mesh_type=mesh_type, turn_over=turn_over, overrides=json.loads(overrides_json) if overrides else None)
    hf = mesh_type != "plane"
    fig.patch_torch()
    names_active = list(env.episode_sums.keys())
    rew_index = {n: ABI.reward_names.index(n) for n in names_active}
    feet = [6, 10, 14, 18]
    G = fig.GYM
    start_counter = 24 * 1000
    env.common_step_counter = start_counter
    env.update_reward_curriculum(force_update=True)

    rec = {k: [] for k in ("U", "actions", "root_in", "dof_in", "contact_in", "feet_in", "torques", "obs", "priv", "rew", "reset",
                           "time_out", "commands", "cmd_timer", "cmd_xy_acc", "last_is_limit_vel", "ep_len", "episode_sums",
                           "root_out", "dof_out", "last_actions", "last_last_actions", "last_dof_vel", "base_lin_vel", "base_ang_vel",
                           "projected_gravity", "rpy", "motor_strengths", "motor_zero_offsets", "p_gains_multiplier", "d_gains_multiplier",
                           "max_move_distance", "episode_info", "episode_info_valid", "terrain_level_info", "ep_len_in", "cmd_timer_in", "max_move_in", "terrain_levels", "env_origins_out", "measured_heights", "turn_over_timer")}

    KINDS = ("wave", "slope", "rough_slope", "stairs_up", "stairs_down", "obstacles", "stepping_stones", "gap", "flat")

    def level_info():
        """extras['episode']['terrain_level_all'] and ['terrain_level_<name>'] (legged_robot.py:231-237) -> [all, 9 kinds]; NaN where the reference
        has no such entry (a terrain name that does not occur) or its group is empty"""
        ep = env.extras["episode"]
        out = np.full(10, np.nan, np.float32)
        out[0] = float(ep["terrain_level_all"])
        for k, nm in enumerate(KINDS):
            if "terrain_level_" + nm in ep:
                out[1 + k] = float(ep["terrain_level_" + nm])
        return out

    def snapshot(t_extras_rebuilt):
        rec["obs"].append(env.obs_buf.numpy().copy()); rec["priv"].append(env.privileged_obs_buf.numpy().copy())
        rec["rew"].append(env.rew_buf.numpy().copy()); rec["reset"].append(env.reset_buf.numpy().astype(np.uint8))
        rec["time_out"].append(env.time_out_buf.numpy().astype(np.uint8))
        rec["commands"].append(env.commands.numpy().copy()); rec["cmd_timer"].append(env.commands_resampling_step.numpy().copy())
        rec["cmd_xy_acc"].append(env.commands_xy_accumulation.numpy().copy())
        rec["last_is_limit_vel"].append(env.last_is_limit_vel.numpy().astype(np.uint8)); rec["ep_len"].append(env.episode_length_buf.numpy().copy())
        es = np.zeros((ABI.GO2_NUM_REWARDS, N), np.float32)
        for n, i in rew_index.items():
            es[i] = env.episode_sums[n].numpy()
        rec["episode_sums"].append(es)
        rec["root_out"].append(env.root_states.numpy().copy()); rec["dof_out"].append(env.dof_state.numpy().reshape(N, 12, 2).copy())
        rec["last_actions"].append(env.last_actions.numpy().copy())
        rec["last_last_actions"].append(env.last_last_actions.numpy().copy() if hasattr(env, "last_last_actions") else np.zeros((N, 12), np.float32))
        rec["last_dof_vel"].append(env.last_dof_vel.numpy().copy())
        for k in ("base_lin_vel", "base_ang_vel", "projected_gravity", "rpy", "motor_strengths", "motor_zero_offsets", "p_gains_multiplier", "d_gains_multiplier", "max_move_distance"):
            rec[k].append(getattr(env, k).numpy().copy())
        rec["terrain_levels"].append(env.terrain_levels.numpy().copy() if hf else np.zeros(N, np.int64))
        rec["env_origins_out"].append(env.env_origins.numpy().copy())
        rec["turn_over_timer"].append(env.turn_over_timer.numpy().copy())
        mh = env.measured_heights
        rec["measured_heights"].append(mh.numpy().copy() if torch.is_tensor(mh) else np.zeros((N, 187), np.float32))
        info = np.zeros(ABI.GO2_NUM_REWARDS, np.float32)
        if t_extras_rebuilt:
            for n, i in rew_index.items():
                info[i] = float(env.extras["episode"]["rew_" + n])
        rec["episode_info"].append(info); rec["episode_info_valid"].append(np.uint8(t_extras_rebuilt))
        rec["terrain_level_info"].append(level_info() if t_extras_rebuilt else np.full(10, np.nan, np.float32))

    def new_table():
        return rng.uniform(0, 1, (N, NU)).astype(np.float32)

    track_curr = bool((overrides or {}).get("commands.curriculum", False))
    if track_curr:
        rec["es_track_in"] = []; rec["cmd_x_range"] = []
    rec_levels0 = env.terrain_levels.numpy().copy() if hf else None
    origins0 = env.env_origins.numpy().copy()
    # ---- reset_idx(all) (base_task.py:82-84) with table 0 ------------------------------------------------
    U0 = new_table()
    fig.INJECT.table = torch.from_numpy(U0)
    env.reset_idx(torch.arange(N))
    reset_all_out = dict(root=env.root_states.numpy().copy(), dof=env.dof_state.numpy().reshape(N, 12, 2).copy(), commands=env.commands.numpy().copy(),
                         motor_strengths=env.motor_strengths.numpy().copy(), motor_zero_offsets=env.motor_zero_offsets.numpy().copy(),
                         p_gains_multiplier=env.p_gains_multiplier.numpy().copy(), d_gains_multiplier=env.d_gains_multiplier.numpy().copy(),
                         cmd_timer=env.commands_resampling_step.numpy().copy(), cmd_xy_acc=env.commands_xy_accumulation.numpy().copy(),
                         last_is_limit_vel=env.last_is_limit_vel.numpy().astype(np.uint8))

    counters = dict(resets=0, time_outs=0, pushes=0, resample_cb=0, limit=0, zero=0)
    for t in range(T):
        if t == 1:
            # what OnPolicyRunner.learn(init_at_random_ep_len=True) does (on_policy_runner.py:117-118); a few envs close to time-out
            el = rng.integers(0, 1250, N)
            el[: N // 4] = rng.integers(1225, 1251, N // 4)
            if turn_over:   # time-outs are the only resets in this mode (:174): bring half of the envs close to one
                el[: N // 2] = rng.integers(1215, 1251, N // 2)
            env.episode_length_buf = torch.from_numpy(el.astype(np.int64))
            # stagger the command timers so the post-physics callback resamples during the sequence (:409-410)
            env.commands_resampling_step[:] = torch.from_numpy(rng.integers(1, 60, N).astype(np.float32))
            if hf:   # spread the walked distances so the terrain curriculum moves envs up and down (:1154-1169)
                env.max_move_distance[:] = torch.from_numpy(rng.uniform(0, 7, N).astype(np.float32))
        if track_curr:
            # update_command_curriculum (:728-737) compares the mean tracking_lin_vel episode sum of the envs being reset with 80 % of its
            # maximum (= 0.8 * 0.02 * 1251 = 20.0): spread the sums around that threshold so that both outcomes occur
            env.episode_sums["tracking_lin_vel"][:] = torch.from_numpy(rng.uniform(8.0, 34.0, N).astype(np.float32))
            rec["es_track_in"].append(env.episode_sums["tracking_lin_vel"].numpy().copy())
        rec["ep_len_in"].append(env.episode_length_buf.numpy().copy())
        rec["cmd_timer_in"].append(env.commands_resampling_step.numpy().copy())
        rec["max_move_in"].append(env.max_move_distance.numpy().copy())
        U = new_table()
        root, dof, contact, feet_state = synth_state(rng, N, env, wide_roll=turn_over)
        actions = rng.normal(0, 1.0, (N, 12)).astype(np.float32)
        actions[rng.uniform(size=(N, 12)) < 0.01] *= 300.0      # exercise clip_actions
        if t == 0:
            actions[:] = 0                                        # reset() steps with zeros (base_task.py:85)
        sub = {"i": 0}

        def on_sim():
            i = sub["i"]
            G.dof[:] = torch.from_numpy(dof[i].reshape(N * 12, 2))
            if i == 3:
                G.root[:] = torch.from_numpy(root)
                G.contact[:] = torch.from_numpy(contact.reshape(N * 19, 3))
                G.rigid.view(N, 19, 13)[:, feet, :] = torch.from_numpy(feet_state)
            sub["i"] += 1
        G.on_simulate = on_sim
        G.torque_log.clear()
        fig.INJECT.table = torch.from_numpy(U)
        fig.INJECT.log.clear()
        pre_extras = env.extras.get("episode", None)
        was_limit = env.last_is_limit_vel.clone()
        env.step(torch.from_numpy(actions))
        rebuilt = env.extras.get("episode", None) is not pre_extras
        rec["U"].append(U); rec["actions"].append(actions); rec["root_in"].append(root); rec["dof_in"].append(dof)
        rec["contact_in"].append(contact); rec["feet_in"].append(feet_state)
        rec["torques"].append(np.stack([x.numpy() for x in G.torque_log]))
        snapshot(rebuilt)
        if track_curr:
            rec["cmd_x_range"].append(np.array([float(env.command_ranges["lin_vel_x"][0]), float(env.command_ranges["lin_vel_x"][1])], np.float32))
            if rebuilt:
                assert float(env.extras["episode"]["max_command_x"]) == float(env.command_ranges["lin_vel_x"][1])
        counters["resets"] += int(env.reset_buf.sum()); counters["time_outs"] += int(env.time_out_buf.sum())
        counters["pushes"] += int((env.episode_length_buf % 200 == 0).sum())
        slots = [s for s, _, _ in fig.INJECT.log]
        counters["resample_cb"] += sum(1 for s in slots if s == fig.U["RSA"] + 3)
        counters["limit"] += sum(1 for s in slots if s in (fig.U["RSA"] + 4, fig.U["RSB"] + 4))
        counters["zero"] += sum(1 for s in slots if s in (fig.U["RSA"] + 5, fig.U["RSB"] + 5))
    # ---- reset_idx(subset) called from outside a step (legged_robot.py:180-245), on the state the sequence ended in --------------------
    ids = np.sort(rng.choice(N, size=max(N // 3, 2), replace=False)).astype(np.int64)
    Ur = new_table()
    if hf:
        env.max_move_distance[:] = torch.from_numpy(rng.uniform(0, 7, N).astype(np.float32))
    ri_in = dict(max_move=env.max_move_distance.numpy().copy())
    fig.INJECT.table = torch.from_numpy(Ur)
    env.reset_idx(torch.from_numpy(ids))
    ri = dict(ids=ids, U=Ur, max_move_in=ri_in["max_move"], root=env.root_states.numpy().copy(), dof=env.dof_state.numpy().reshape(N, 12, 2).copy(),
              commands=env.commands.numpy().copy(), cmd_timer=env.commands_resampling_step.numpy().copy(), cmd_xy_acc=env.commands_xy_accumulation.numpy().copy(),
              ep_len=env.episode_length_buf.numpy().copy(), reset=env.reset_buf.numpy().astype(np.uint8), time_out=env.time_out_buf.numpy().astype(np.uint8),
              obs=env.obs_buf.numpy().copy(), priv=env.privileged_obs_buf.numpy().copy(), rew=env.rew_buf.numpy().copy(),
              last_actions=env.last_actions.numpy().copy(), last_dof_vel=env.last_dof_vel.numpy().copy(), actions=env.actions.numpy().copy(),
              feet_air_time=env.feet_air_time.numpy().copy(), base_lin_vel=env.base_lin_vel.numpy().copy(), rpy=env.rpy.numpy().copy(),
              motor_strengths=env.motor_strengths.numpy().copy(), motor_zero_offsets=env.motor_zero_offsets.numpy().copy(),
              p_gains_multiplier=env.p_gains_multiplier.numpy().copy(), d_gains_multiplier=env.d_gains_multiplier.numpy().copy(),
              last_is_limit_vel=env.last_is_limit_vel.numpy().astype(np.uint8), max_move=env.max_move_distance.numpy().copy(),
              env_origins=env.env_origins.numpy().copy(), terrain_levels=env.terrain_levels.numpy().copy() if hf else np.zeros(N, np.int64),
              turn_over_timer=env.turn_over_timer.numpy().copy())
    es = np.zeros((ABI.GO2_NUM_REWARDS, N), np.float32); info = np.zeros(ABI.GO2_NUM_REWARDS, np.float32)
    for n_, i_ in rew_index.items():
        es[i_] = env.episode_sums[n_].numpy(); info[i_] = float(env.extras["episode"]["rew_" + n_])
    ri["episode_sums"] = es; ri["episode_info"] = info; ri["terrain_level_info"] = level_info()
    if track_curr:
        ri["cmd_x_range"] = np.array([float(env.command_ranges["lin_vel_x"][0]), float(env.command_ranges["lin_vel_x"][1])], np.float32)
    fig.INJECT.table = None
    fig.unpatch_torch()
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update({"reset_idx_" + k: v for k, v in ri.items()})
    out.update({"reset_all_" + k: v for k, v in reset_all_out.items()})
    out["U_reset_all"] = U0
    out["start_counter"] = np.int64(start_counter)
    out["env_origins"] = origins0
    out["dof_pos_limits"] = env.dof_pos_limits.numpy().copy()
    out["torque_limits"] = env.torque_limits.numpy().copy()
    out["noise_scale_vec"] = env.noise_scale_vec.numpy().copy()
    scales = np.zeros(ABI.GO2_NUM_REWARDS, np.float32)
    for n, i in rew_index.items():
        scales[i] = env.reward_scales.get(n, 0.0)
    out["reward_scales_dt"] = scales
    if turn_over:
        ts = np.zeros(ABI.GO2_NUM_REWARDS, np.float32)
        for n, i in rew_index.items():
            ts[i] = env.reward_turn_over_scales.get(n, 0.0)
        out["turn_over_scales_dt"] = ts
    out["height_points"] = env.height_points[0].numpy().copy()
    out["base_height_scan_mask"] = env.base_height_scan_mask.numpy().copy()
    out["limit_vel_comb"] = env.limit_vel_comb.numpy().astype(np.float32)
    if hf:
        out["terrain_levels0"] = rec_levels0; out["terrain_types"] = env.terrain_types.numpy().copy()
        out["hf_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(env.terrain.height_field_raw).tobytes()).digest(), np.uint8)
        out["terrain_seed"] = np.int64(TERRAIN_SEED)
    if hf:
        lv = np.stack(rec["terrain_levels"])
        counters["level_changes"] = int((np.diff(np.concatenate([rec_levels0[None], lv]), axis=0) != 0).sum())
        counters["nonzero_heights"] = float((np.abs(np.stack(rec["measured_heights"])) > 0).mean())
    if overrides:
        out["cfg_overrides"] = np.array(overrides_json)
    if turn_over:
        tt = np.stack(rec["turn_over_timer"])
        counters["timer_running"] = int((tt > 0).sum()); counters["rolled"] = int((np.abs(np.stack(rec["rpy"])[..., 0]) > np.pi / 4).sum())
        out["turn_over"] = np.int64(1)
    print("env sequence (%s%s): N=%d T=%d events:" % (mesh_type, ", turn_over" if turn_over else "", N, T), counters, "active rewards:", sorted(names_active))
    return out


def gen_terrain():
    """legged_gym/utils/terrain.py:9-174 run on this build's generators: layout, per-column kinds, origins, placed height field."""
    from legged_gym.utils.terrain import Terrain
    env_cfg, _ = task_registry.get_cfgs("go2")
    env_cfg.terrain.mesh_type = "heightfield"
    np.random.seed(TERRAIN_SEED)
    t = Terrain(env_cfg.terrain, 64)
    hfr = np.ascontiguousarray(t.height_field_raw)
    return dict(seed=np.int64(TERRAIN_SEED), shape=np.array(hfr.shape), sha256=np.frombuffer(hashlib.sha256(hfr.tobytes()).digest(), np.uint8),
                env_origins=t.env_origins.copy(), cols2id=np.array(t.cols2id), tot_rows=np.int64(t.tot_rows), tot_cols=np.int64(t.tot_cols),
                tile_wave=hfr[250:330, 250:330].copy(), tile_stairs=hfr[250 + 9 * 85:330 + 9 * 85, 250 + 8 * 85:330 + 8 * 85].copy(),
                tile_obstacles=hfr[250 + 5 * 85:330 + 5 * 85, 250 + 14 * 85:330 + 14 * 85].copy(), col_sums=hfr.astype(np.int64).sum(0), row_sums=hfr.astype(np.int64).sum(1))


def gen_gae(seed=3):
    from rsl_rl.storage import RolloutStorage
    rng = np.random.default_rng(seed)
    T, N = 24, 48
    st = RolloutStorage(N, T, [45], [263], [12], "cpu")
    st.rewards[:] = torch.from_numpy(rng.normal(0, 0.05, (T, N, 1)).astype(np.float32))
    st.values[:] = torch.from_numpy(rng.normal(0, 1.0, (T, N, 1)).astype(np.float32))
    st.dones[:] = torch.from_numpy((rng.uniform(size=(T, N, 1)) < 0.06).astype(np.uint8))
    last = torch.from_numpy(rng.normal(0, 1.0, (N, 1)).astype(np.float32))
    st.compute_returns(last, 0.99, 0.95)
    return dict(rewards=st.rewards.numpy()[..., 0], values=st.values.numpy()[..., 0], dones=st.dones.numpy()[..., 0], last_values=last.numpy()[:, 0],
                returns=st.returns.numpy()[..., 0], advantages=st.advantages.numpy()[..., 0], gamma=np.float32(0.99), lam=np.float32(0.95))


def gen_ppo(seed=5):
    """One PPO.update (ppo.py:120-187) on a tiny actor-critic; also PPO.act statistics (ppo.py:90-102)."""
    from rsl_rl.algorithms import PPO
    from rsl_rl.modules import ActorCritic
    torch.manual_seed(seed)
    T, N = 6, 32
    ac = ActorCritic(45, 263, 12, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", init_noise_std=1.0)
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=2, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device="cpu")
    alg.init_storage(N, T, [45], [263], [12])
    sd0 = {k: v.detach().numpy().copy() for k, v in ac.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T + 1, N, 45, generator=g); cobs = torch.randn(T + 1, N, 263, generator=g)
    rew = torch.randn(T, N, generator=g) * 0.05; dones = torch.rand(T, N, generator=g) < 0.1; touts = dones & (torch.rand(T, N, generator=g) < 0.5)
    noise = torch.randn(T, N, 12, generator=g)
    acts, vals, logps = [], [], []
    for t in range(T):
        # PPO.act with the sampled noise made explicit: a = mu + std * eps
        ac.update_distribution(obs[t])
        a = (ac.action_mean + ac.action_std * noise[t]).detach()
        alg.transition.actions = a
        alg.transition.values = ac.evaluate(cobs[t]).detach()
        alg.transition.actions_log_prob = ac.get_actions_log_prob(a).detach()
        alg.transition.action_mean = ac.action_mean.detach(); alg.transition.action_sigma = ac.action_std.detach()
        alg.transition.observations = obs[t]; alg.transition.critic_observations = cobs[t]
        acts.append(a.numpy().copy()); vals.append(alg.transition.values.numpy().copy()); logps.append(alg.transition.actions_log_prob.numpy().copy())
        alg.process_env_step(rew[t], dones[t], {"time_outs": touts[t]})
    alg.compute_returns(cobs[T])
    returns = alg.storage.returns.numpy().copy(); adv = alg.storage.advantages.numpy().copy(); srew = alg.storage.rewards.numpy().copy()
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(seed + 1))
    orig_randperm = torch.randperm
    torch.randperm = lambda n, **kw: perm
    try:
        mvl, msl = alg.update()
    finally:
        torch.randperm = orig_randperm
    sd1 = {k: v.detach().numpy().copy() for k, v in ac.state_dict().items()}
    out = dict(obs=obs.numpy(), cobs=cobs.numpy(), rew=rew.numpy(), dones=dones.numpy().astype(np.uint8), time_outs=touts.numpy().astype(np.uint8), noise=noise.numpy(),
               actions=np.stack(acts), values=np.stack(vals), logp=np.stack(logps), returns=returns, advantages=adv, stored_rewards=srew, perm=perm.numpy(),
               mean_value_loss=np.float64(mvl), mean_surrogate_loss=np.float64(msl), final_lr=np.float64(alg.learning_rate))
    for k, v in sd0.items(): out["w0_" + k] = v
    for k, v in sd1.items(): out["w1_" + k] = v
    import tempfile
    from legged_gym.utils.exporter import export_policy_as_jit
    d = tempfile.mkdtemp()
    export_policy_as_jit(ac, d, filename="p.pt")
    jit = torch.jit.load(os.path.join(d, "p.pt"))
    out["jit_actions"] = jit(obs[0][:3]).detach().numpy().copy()
    return out


def full_size_weights(state_dict_keys_shapes, seed=9):
    """Initial weights of the FULL-SIZE go2 actor-critic (45-512-256-128-12 / 263-512-256-128-1) as a pure function of numpy's PCG64 stream, so
    that a test can rebuild them bit for bit instead of the fixture carrying 2 MB of them: U(-b, b), b = 1 / sqrt(fan_in), tensor after tensor
    in state_dict order; std = 1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k, shp in state_dict_keys_shapes:
        if k == "std":
            out[k] = np.ones(shp, np.float32)
        else:
            fan_in = shp[1] if len(shp) == 2 else None
            b = 1.0 / np.sqrt(fan_in if fan_in else (512 if shp[0] == 512 else shp[0]))
            out[k] = rng.uniform(-b, b, shp).astype(np.float32)
    return out


def gen_ppo_full(seed=9):
    """One PPO.update (ppo.py:120-187) of the reference on the FULL-SIZE networks of task go2 (go2_config.py:219-221: 512-256-128), 4 Adam steps on
    96-row mini-batches.  The fixture holds the inputs, the rollout statistics, and of the 488 857 final weights a fixed sample (every element of
    the small tensors, 20 000 per large one) plus each tensor's sum — a test compares those element by element."""
    from rsl_rl.algorithms import PPO
    from rsl_rl.modules import ActorCritic
    torch.manual_seed(seed)
    T, N = 6, 32
    ac = ActorCritic(45, 263, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    w0 = full_size_weights([(k, tuple(v.shape)) for k, v in ac.state_dict().items()], seed)
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=2, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device="cpu")
    alg.init_storage(N, T, [45], [263], [12])
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T + 1, N, 45, generator=g); cobs = torch.randn(T + 1, N, 263, generator=g)
    rew = torch.randn(T, N, generator=g) * 0.05; dones = torch.rand(T, N, generator=g) < 0.1; touts = dones & (torch.rand(T, N, generator=g) < 0.5)
    noise = torch.randn(T, N, 12, generator=g)
    acts, vals, logps = [], [], []
    for t in range(T):
        ac.update_distribution(obs[t])
        a = (ac.action_mean + ac.action_std * noise[t]).detach()
        alg.transition.actions = a
        alg.transition.values = ac.evaluate(cobs[t]).detach()
        alg.transition.actions_log_prob = ac.get_actions_log_prob(a).detach()
        alg.transition.action_mean = ac.action_mean.detach(); alg.transition.action_sigma = ac.action_std.detach()
        alg.transition.observations = obs[t]; alg.transition.critic_observations = cobs[t]
        acts.append(a.numpy().copy()); vals.append(alg.transition.values.numpy().copy()); logps.append(alg.transition.actions_log_prob.numpy().copy())
        alg.process_env_step(rew[t], dones[t], {"time_outs": touts[t]})
    alg.compute_returns(cobs[T])
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(seed + 1))
    orig_randperm = torch.randperm
    torch.randperm = lambda n, **kw: perm
    try:
        mvl, msl = alg.update()
    finally:
        torch.randperm = orig_randperm
    out = dict(seed=np.int64(seed), obs=obs.numpy(), cobs=cobs.numpy(), rew=rew.numpy(), dones=dones.numpy().astype(np.uint8), time_outs=touts.numpy().astype(np.uint8),
               noise=noise.numpy(), actions=np.stack(acts), values=np.stack(vals), logp=np.stack(logps), returns=alg.storage.returns.numpy().copy(),
               advantages=alg.storage.advantages.numpy().copy(), perm=perm.numpy(), mean_value_loss=np.float64(mvl), mean_surrogate_loss=np.float64(msl),
               final_lr=np.float64(alg.learning_rate))
    pick = np.random.Generator(np.random.PCG64(seed + 2))
    for k, v in ac.state_dict().items():
        w1 = v.detach().numpy().reshape(-1)
        idx = np.arange(w1.size) if w1.size <= 20000 else np.sort(pick.choice(w1.size, 20000, replace=False))
        out["idx_" + k] = idx.astype(np.int32); out["w1_" + k] = w1[idx].copy(); out["sum1_" + k] = np.float64(w1.astype(np.float64).sum())
        out["step_" + k] = np.float32(np.abs(w1 - w0[k].reshape(-1)).max())       # how far the 4 Adam steps moved the tensor (a test must see the same scale)
    return out


def gen_pretrained(seed=31):
    """I/O vectors of the reference's own pretrained deployment policy (deploy/pre_train/go2/go2_cts_150k.pt, a TorchScript export of
    a CTS student) plus its tensors, so the build's loader/exporter and the behavioural walking test can run where the reference
    tree is absent.  DATA of the reference (weights + input/output vectors), not source."""
    path = "/root/reference/deploy/pre_train/go2/go2_cts_150k.pt"
    jit = torch.jit.load(path)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in jit.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(12, 1, 45, generator=g) * 0.5
    acts, lats = [], []
    for t in list(range(8)) + ["reset"] + list(range(8, 12)):
        if t == "reset":
            jit.reset()
            continue
        o = jit(obs[t])
        a, l = (o[0], o[1][1]) if isinstance(o, tuple) else (o, torch.zeros(1, 0))
        acts.append(a.detach().numpy().copy()); lats.append(l.detach().numpy().copy())
    out = dict(obs=obs.numpy(), actions=np.stack(acts), latent=np.stack(lats), sha256=np.frombuffer(hashlib.sha256(open(path, "rb").read()).digest(), np.uint8))
    for k, v in sd.items():
        out["w_" + k] = v
    print("pretrained:", {k: v.shape for k, v in sd.items()})
    return out


class _ScriptedEnv:
    """The VecEnv surface OnPolicyRunnerCTS consumes (rsl_rl/env/vec_env.py), replaying pre-generated tensors: the runner-level
    fixtures need the reference's RUNNER + ALGORITHM + STORAGE + MODULES, not its simulator."""

    class _Cfg:
        class env:
            test = True

    def __init__(self, obs, priv, rew, dones, touts):
        self.obs_seq, self.priv_seq, self.rew_seq, self.done_seq, self.tout_seq = obs, priv, rew, dones, touts
        self.num_envs, self.num_obs, self.num_privileged_obs, self.num_actions = obs.shape[1], obs.shape[2], priv.shape[2], 12
        self.max_episode_length = 1000
        self.episode_length_buf = torch.zeros(self.num_envs, dtype=torch.long)
        self.cfg = self._Cfg()
        self.t = 0
        self.actions = []

    def reset(self):
        return self.obs_seq[0], self.priv_seq[0]

    def get_observations(self):
        return self.obs_seq[self.t]

    def get_privileged_observations(self):
        return self.priv_seq[self.t]

    def step(self, actions):
        self.actions.append(actions.clone().numpy())
        t = self.t
        self.t += 1
        # 'episode': what LeggedRobot.reset_idx reports (:229-242) - a reward term and a terrain key, for the runners' scalar tags
        return self.obs_seq[t + 1], self.priv_seq[t + 1], self.rew_seq[t], self.done_seq[t], {
            "time_outs": self.tout_seq[t], "episode": {"rew_tracking_lin_vel": torch.tensor(0.25), "terrain_level": 1.5}}


class _TagRecorder:
    """stands in for the SummaryWriter: the scalar tags a runner writes, in order"""

    def __init__(self):
        self.tags = []

    def add_scalar(self, tag, *a, **k):
        self.tags.append(tag)


def gen_cts(kind, seed=21):
    """ONE iteration of the reference's OnPolicyRunnerCTS.learn (on_policy_runner_cts.py:123-202) on a scripted env with a small
    network: history ring, CTS.act / process_env_step, compute_returns, the two update loops, checkpoint keys."""
    import tempfile
    from rsl_rl.runners import OnPolicyRunnerCTS
    import rsl_rl.modules.actor_critic_cts as ref_ac_cts
    torch.manual_seed(seed)
    T, N, H = 6, 32, 5
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T + 1, N, 45, generator=g); priv = torch.randn(T + 1, N, 263, generator=g)
    rew = torch.randn(T, N, generator=g) * 0.05; dones = torch.rand(T, N, generator=g) < 0.12; touts = dones & (torch.rand(T, N, generator=g) < 0.5)
    noise = torch.randn(T, N, 12, generator=g)
    env = _ScriptedEnv(obs, priv, rew, dones, touts)
    policy = dict(init_noise_std=1.0, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], teacher_encoder_hidden_dims=[32, 16],
                  student_encoder_hidden_dims=[32, 16] if kind == "CTS" else [32, 16, 8], activation="elu", latent_dim=8, norm_type="l2norm")
    algorithm = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=2, num_mini_batches=2,
                     learning_rate=1e-3, student_encoder_learning_rate=1e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0,
                     teacher_env_ratio=0.75)
    if kind == "MoECTS":
        policy["expert_num"] = 4
        algorithm["load_balance_coef"] = 0.01
    if kind in ("ACMoECTS", "DualMoECTS"):
        policy["expert_num"] = 4
        policy["student_encoder_hidden_dims"] = [32, 16] if kind == "ACMoECTS" else [32, 16, 8]
        policy["actor_hidden_dims"] = [32, 16, 8]; policy["critic_hidden_dims"] = [32, 16, 8]
    if kind == "MCPCTS":
        policy.pop("init_noise_std")
        policy.update(actor_hidden_dims=[32, 16], student_expert_num=4, obs_no_goal_mask=[True] * 6 + [False] * 3 + [True] * 36)
    if kind == "MoENGCTS":
        policy["student_encoder_hidden_dims"] = [32, 16]
        policy["student_expert_num"] = 4
        policy["obs_no_goal_mask"] = [True] * 6 + [False] * 3 + [True] * 36
        algorithm["load_balance_coef"] = 0.01
    train_cfg = {"runner": dict(policy_class_name="ActorCritic" + kind, algorithm_class_name=kind, num_steps_per_env=T, max_iterations=1, save_interval=1000,
                                experiment_name="golden", run_name=""),
                 "algorithm": algorithm, "policy": policy, "history_length": H, "robogauge": {"enabled": False, "port": 0}}
    # the reference allocates the model's deployment history on 'cuda' unconditionally (actor_critic_cts.py:48): run it on the CPU
    real_zeros = torch.zeros
    ref_ac_cts.torch.zeros = lambda *a, **k: real_zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        runner = OnPolicyRunnerCTS(env, train_cfg, log_dir=tempfile.mkdtemp(), device="cpu")
    finally:
        ref_ac_cts.torch.zeros = real_zeros
    alg = runner.alg
    ti, si = alg.teacher_env_idxs, alg.student_env_idxs
    sd0 = {k: v.detach().numpy().copy() for k, v in alg.model.state_dict().items()}
    # sampling noise made explicit: a = mu + std * eps with eps in ENV order; the reference samples teacher rows, then student rows
    state = {"t": 0, "rollout": True}
    from torch.distributions import Normal
    real_sample = Normal.sample

    def sample(self, sample_shape=torch.Size()):
        if not state["rollout"]:
            return self.mean.detach()
        rows = ti if self.mean.shape[0] == len(ti) else si
        eps = noise[state["t"]][rows]
        if rows is si:
            state["t"] += 1
        return (self.mean + self.stddev * eps).detach()

    perms = {len(ti) * T: torch.randperm(len(ti) * T, generator=torch.Generator().manual_seed(seed + 1)),
             len(si) * T: torch.randperm(len(si) * T, generator=torch.Generator().manual_seed(seed + 2))}
    real_randperm, real_update = torch.randperm, alg.update
    rec = {}

    def update():
        state["rollout"] = False
        st = alg.storage
        order = torch.cat([ti, si])
        inv = torch.empty_like(order); inv[order] = torch.arange(N)
        for k in ("returns", "advantages", "values", "rewards", "actions_log_prob", "history", "observations", "mu"):
            rec["storage_" + k] = getattr(st, k)[:, inv].numpy().copy()                 # back to env order
        rec["history_after_rollout"] = runner.history.numpy().copy()
        return real_update()

    Normal.sample = sample
    torch.randperm = lambda n, **kw: perms[n]
    alg.update = update
    runner.writer = _TagRecorder()
    try:
        runner.learn(1, init_at_random_ep_len=False)
    finally:
        Normal.sample, torch.randperm = real_sample, real_randperm
    sd1 = {k: v.detach().numpy().copy() for k, v in alg.model.state_dict().items()}
    ckpt = torch.load(os.path.join(runner.log_dir, "model_1.pt"), weights_only=False)
    out = dict(obs=obs.numpy(), priv=priv.numpy(), rew=rew.numpy(), dones=dones.numpy().astype(np.uint8), time_outs=touts.numpy().astype(np.uint8), noise=noise.numpy(),
               actions=np.stack(env.actions), perm_teacher=perms[len(ti) * T].numpy(), perm_student=perms[len(si) * T].numpy(),
               teacher_env_idxs=ti.numpy(), student_env_idxs=si.numpy(), final_lr=np.float64(alg.learning_rate),
               checkpoint_keys=np.array(sorted(ckpt.keys())), optimizer1_groups=np.array([len(g["params"]) for g in ckpt["optimizer1_state_dict"]["param_groups"]]),
               optimizer2_groups=np.array([len(g["params"]) for g in ckpt["optimizer2_state_dict"]["param_groups"]]),
               log_tags=np.array(runner.writer.tags), **rec)
    # act_inference on a fresh model copy state: deployment path (history shift inside the module)
    alg.model.history[:] = 0
    inf = [alg.model.act_inference(obs[t]).detach().numpy().copy() for t in range(3)]
    out["act_inference"] = np.stack(inf)
    # deployment export through the reference's exporter (legged_gym/utils/exporter.py:13-23,67-193)
    from legged_gym.utils.exporter import export_policy_as_jit
    d = tempfile.mkdtemp()
    export_policy_as_jit(alg.model, d, filename="p.pt")
    jit = torch.jit.load(os.path.join(d, "p.pt"))
    acts, lats, wts = [], [], []
    for t in list(range(4)) + ["reset", 0, 1]:
        if t == "reset":
            jit.reset()
            continue
        a, extra = jit(obs[t][:1])
        w, l = extra[0], extra[-1]                      # (weights | None, [actor weights,] latent)
        acts.append(a.detach().numpy().copy()); lats.append(l.detach().numpy().copy())
        if w is not None:
            wts.append(np.concatenate([x.detach().numpy() for x in extra[:-1]], axis=-1))
    out["jit_actions"], out["jit_latent"] = np.stack(acts), np.stack(lats)
    if wts:
        out["jit_weights"] = np.stack(wts)
    for k, v in sd0.items(): out["w0_" + k] = v
    for k, v in sd1.items(): out["w1_" + k] = v
    print("cts golden (%s): dones=%d, lr=%g, params=%d" % (kind, int(dones.sum()), alg.learning_rate, sum(v.size for v in sd0.values())))
    return out


def gen_ppo_runner(seed=41):
    """ONE iteration of the reference's OnPolicyRunner.learn (on_policy_runner.py:113-172) on the scripted env: the scalar tags it logs
    (:185-207), the checkpoint layout (:243-250) and the parameter names."""
    import tempfile
    from rsl_rl.runners import OnPolicyRunner
    g = torch.Generator().manual_seed(seed)
    N, T = 8, 6
    obs, priv = torch.randn(T + 1, N, 45, generator=g), torch.randn(T + 1, N, 263, generator=g)
    rew = torch.randn(T, N, generator=g) * 0.05
    dones = torch.rand(T, N, generator=g) < 0.3
    touts = dones & (torch.rand(T, N, generator=g) < 0.5)
    env = _ScriptedEnv(obs, priv, rew, dones, touts)
    train_cfg = {"runner": dict(policy_class_name="ActorCritic", algorithm_class_name="PPO", num_steps_per_env=T, max_iterations=1, save_interval=50,
                                experiment_name="golden", run_name=""),
                 "algorithm": dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=2, num_mini_batches=2,
                                   learning_rate=1e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0),
                 "policy": dict(init_noise_std=1.0, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu"),
                 "robogauge": {"enabled": False, "port": 0}}
    torch.manual_seed(seed)
    runner = OnPolicyRunner(env, train_cfg, log_dir=tempfile.mkdtemp(), device="cpu")
    runner.writer = _TagRecorder()
    runner.learn(1, init_at_random_ep_len=False)
    ckpt = torch.load(os.path.join(runner.log_dir, "model_1.pt"), weights_only=False)
    return dict(log_tags=np.array(runner.writer.tags), checkpoint_keys=np.array(sorted(ckpt.keys())), checkpoint_iter=np.int64(ckpt["iter"]),
                state_dict_keys=np.array(list(ckpt["model_state_dict"].keys())), saved_files=np.array(sorted(f for f in os.listdir(runner.log_dir) if f.startswith("model_"))),
                obs=obs.numpy(), priv=priv.numpy(), rew=rew.numpy(), dones=dones.numpy().astype(np.uint8), time_outs=touts.numpy().astype(np.uint8))


def _save(files, name, data):
    """Write tests/golden/<name> unless it already holds exactly these arrays (zip timestamps would churn the git history)."""
    path = os.path.join(OUT, name)
    files[name] = None
    if os.path.exists(path):
        old = dict(np.load(path))
        same = lambda a, b: a.shape == b.shape and (np.allclose(a, b, atol=4e-6, rtol=0) if a.dtype.kind == "f" else np.array_equal(a, b))
        if set(old) == set(data) and all(same(old[k], np.asarray(data[k])) for k in data):
            return          # (torch's threaded CPU reductions make the last bit of a few float outputs vary from run to run)
    np.savez_compressed(path, **data)


def main():
    os.makedirs(OUT, exist_ok=True)
    files = {}
    seq = gen_env_sequence()
    _save(files, "go2_plane_sequence.npz", seq)
    hfseq = gen_env_sequence(N=12, T=40, seed=9, mesh_type="heightfield")
    _save(files, "go2_heightfield_sequence.npz", hfseq)
    _save(files, "go2_turn_over_sequence.npz", gen_env_sequence(N=12, T=40, seed=13, turn_over=True))
    _save(files, "go2_alt_sequence.npz", gen_env_sequence(N=12, T=48, seed=17, mesh_type="heightfield", overrides=ALT_OVERRIDES))
    # (task_registry.get_cfgs returns the REGISTERED config instance, task_registry.py:24-28, so the ALT overrides above stay applied in this
    #  process: they are part of these sequences' configuration and are recorded as such)
    # the branches of the cited line ranges that no go2 task switches on: _compute_torques control types 'V' / 'T' (:612-615) and
    # commands.curriculum (:225-226,:241-242,:728-737) across the start of a command_range_curriculum stage (:433-446)
    _save(files, "go2_control_v_sequence.npz", gen_env_sequence(N=8, T=24, seed=19, overrides={**ALT_OVERRIDES, "control.control_type": "V"}))
    _save(files, "go2_control_t_sequence.npz", gen_env_sequence(N=8, T=24, seed=23, overrides={**ALT_OVERRIDES, "control.control_type": "T"}))
    _save(files, "go2_cmd_curriculum_sequence.npz", gen_env_sequence(N=12, T=56, seed=29, overrides={
        **ALT_OVERRIDES, "control.control_type": "P", "commands.curriculum": True, "commands.max_curriculum": 1.5,
        "commands.command_range_curriculum": [{"iter": 1001, "lin_vel_x": [-1.0, 1.0], "lin_vel_y": [-1.0, 1.0], "ang_vel_yaw": [-1.5, 1.5], "heading": [-1.57, 1.57]}]}))
    _save(files, "terrain.npz", gen_terrain())
    _save(files, "gae.npz", gen_gae())
    _save(files, "ppo_update.npz", gen_ppo())
    _save(files, "ppo_update_full.npz", gen_ppo_full())
    _save(files, "ppo_runner_log.npz", gen_ppo_runner())
    _save(files, "pretrained_go2_cts_150k.npz", gen_pretrained())
    _save(files, "cts_iteration.npz", gen_cts("CTS"))
    _save(files, "moe_cts_iteration.npz", gen_cts("MoECTS"))
    _save(files, "moe_ng_cts_iteration.npz", gen_cts("MoENGCTS"))
    _save(files, "ac_moe_cts_iteration.npz", gen_cts("ACMoECTS"))
    _save(files, "dual_moe_cts_iteration.npz", gen_cts("DualMoECTS"))
    _save(files, "mcp_cts_iteration.npz", gen_cts("MCPCTS"))
    for f in files:
        files[f] = hashlib.sha256(open(os.path.join(OUT, f), "rb").read()).hexdigest()
    try:
        ref = subprocess.run(["git", "-C", "/root/reference", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or "snapshot 2026-02-20 (no git metadata)"
    except Exception:
        ref = "snapshot 2026-02-20"
    notes = {"pretrained_go2_cts_150k.npz": "holds the reference's TRAINED policy weights (deploy/pre_train/go2/go2_cts_150k.pt: student_encoder + actor tensors) next to its "
             "input/output vectors — fixture DATA for the behavioural test of the physics model (tests/test_export.py, tests/test_gpu_parity.py: a PhysX-trained policy walks "
             "in this simulator); read only by tests, never by anything under go2_rl_gym_amd/, bench.py's timed region or smoke()'s product path"}
    json.dump({"generator": "oracle/gen_golden.py", "reference": "wty-yy/go2_rl_gym @ " + ref, "torch": torch.__version__, "numpy": np.__version__, "files": files, "notes": notes},
              open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)
    for f in files:
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
