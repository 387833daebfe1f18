#!/usr/bin/env python3
"""TOOL (GPU): what are the rare episodes of reward ~ -1e4 that tools/long_run.py go2 shows (profiles/r6_long_run_go2.txt)?  Rolls the policy trained there (build/long_run_go2.pt, or
zero-mean noise without it) on the rough curriculum map with the curriculum's final command ranges and exploration noise, watches every env-step for a reward below -5 or a base speed above
20 m/s and prints the state one step BEFORE and at the event.      python tools/outlier_probe.py [steps] [noise std]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from go2_rl_gym_amd.envs import task_registry
from go2_rl_gym_amd.utils import get_args
from go2_rl_gym_amd.utils.terrain import KIND_NAMES

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
std = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
task = "go2"
args = get_args(["--task", task, "--num_envs", "4096", "--headless", "--seed", "1"])
env_cfg, train_cfg = task_registry.get_cfgs(task)
env_cfg.commands.command_range_curriculum = []
env_cfg.commands.ranges.lin_vel_x, env_cfg.commands.ranges.lin_vel_y, env_cfg.commands.ranges.ang_vel_yaw = [-2.0, 2.0], [-1.0, 1.0], [-2.0, 2.0]
env_cfg.terrain.max_init_terrain_level = 9
env, _ = task_registry.make_env(task, args, env_cfg=env_cfg)
runner, _ = task_registry.make_alg_runner(env, None, args, train_cfg=train_cfg, log_root=None)
model = runner.alg.actor_critic
ck = os.path.join(ROOT, "build", "long_run_go2.pt")
if os.path.exists(ck):
    model.load_state_dict(torch.load(ck, map_location=env.device)); print("policy: build/long_run_go2.pt")
else:
    print("policy: untrained")
env.common_step_counter = 0
env.update_reward_curriculum(force_update=True)
obs = env.get_observations()
g = torch.Generator(device=env.device).manual_seed(5)
keys = ("root_states", "dof_vel", "dof_pos", "contact_forces", "base_lin_vel", "projected_gravity", "torques")
prev = None
events, by_kind, n_steps = [], {}, 0
kinds = env.terrain_ids if hasattr(env, "terrain_ids") else None
for it in range(steps):
    with torch.no_grad():
        a = model.act_inference(obs)
        a = a + std * torch.randn(a.shape, device=a.device, generator=g)
    snap = {k: getattr(env, k).clone() for k in keys}
    snap["actions"] = a.clone(); snap["levels"] = env.terrain_levels.clone(); snap["types"] = env.terrain_types.clone(); snap["ep"] = env.episode_length_buf.clone()
    obs, _, rew, dones, infos = env.step(a)
    n_steps += env.num_envs
    speed = env.root_states[:, 7:10].norm(dim=1)
    bad = (rew < -5.0) | (speed > 20.0) | (snap["root_states"][:, 7:10].norm(dim=1) > 20.0)
    if bad.any():
        for e in torch.nonzero(bad).flatten().tolist()[:4]:
            kd = int(env.terrain_ids[env.terrain_types[e]]) if hasattr(env, "terrain_ids") else -1
            by_kind[kd] = by_kind.get(kd, 0) + 1
            if len(events) < 14:
                p, c = snap, env
                cf = p["contact_forces"][e].norm(dim=1)
                events.append("step %d env %d | terrain %s level %d | episode step %d | reward %.1f done %d\n"
                              "    before: z %.3f (origin z %.3f) v %s w %s gravity_b %s | max |qd| %.1f max |tau| %.1f max |a| %.1f | contact |F|: base %.0f max other %.0f (body %d)\n"
                              "    after : z %.3f v %s w %s | max |qd| %.1f | contact |F| max %.0f"
                              % (it, e, KIND_NAMES[kd] if kd >= 0 else "?", int(p["levels"][e]), int(p["ep"][e]), float(rew[e]), int(dones[e]),
                                 float(p["root_states"][e, 2]), float(env.env_origins[e, 2]), np.round(p["root_states"][e, 7:10].tolist(), 2), np.round(p["root_states"][e, 10:13].tolist(), 2),
                                 np.round(p["projected_gravity"][e].tolist(), 2), float(p["dof_vel"][e].abs().max()), float(p["torques"][e].abs().max()), float(p["actions"][e].abs().max()),
                                 float(cf[0]), float(cf[1:].max()), int(cf[1:].argmax()) + 1,
                                 float(c.root_states[e, 2]), np.round(c.root_states[e, 7:10].tolist(), 2), np.round(c.root_states[e, 10:13].tolist(), 2), float(c.dof_vel[e].abs().max()),
                                 float(c.contact_forces[e].norm(dim=1).max())))
torch.cuda.synchronize()
print("%d env-steps (%d robots x %d steps), noise std %.2f: %d env-steps with reward < -5 or base speed > 20 m/s; by terrain kind %s" % (n_steps, env.num_envs, steps, std, sum(by_kind.values()), {KIND_NAMES[k] if k >= 0 else "?": v for k, v in by_kind.items()}))
for ev in events:
    print(ev)
