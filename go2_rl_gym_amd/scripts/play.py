"""python -m go2_rl_gym_amd.scripts.play --task go2_flat  (legged_gym/scripts/play.py:15-62, headless).

Loads the latest checkpoint of the task's experiment through the runner, exports the deployment policy
(TorchScript / pkl, and ONNX when the `onnx` package is present) and rolls it out with a fixed 1 m/s command."""
import os

import torch

from go2_rl_gym_amd.envs import *  # noqa: F401,F403
from go2_rl_gym_amd.utils import get_args
from go2_rl_gym_amd.utils.exporter import export_policy_as_jit, export_policy_as_onnx, export_policy_as_pkl
from go2_rl_gym_amd.utils.task_registry import ROOT_DIR, task_registry

EXPORT_POLICY = True
FIX_COMMAND = True


def play(args, steps=None, log_root="default", export_policy=None):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 100)
    env_cfg.terrain.num_rows = 7
    env_cfg.terrain.num_cols = 7
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = False
    dr = env_cfg.domain_rand
    dr.randomize_friction = dr.push_robots = dr.randomize_base_mass = dr.randomize_link_mass = False
    dr.randomize_base_com = dr.randomize_pd_gains = dr.randomize_motor_zero_offset = False
    env_cfg.env.test = True
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = True
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg, log_root=log_root)
    policy = runner.get_inference_policy(device=env.device)
    exported = None
    if EXPORT_POLICY if export_policy is None else export_policy:
        root = os.path.join(ROOT_DIR, "logs", train_cfg.runner.experiment_name) if log_root == "default" else log_root
        path = os.path.join(root, "exported", "policies")
        model = runner.alg.actor_critic
        exported = [export_policy_as_jit(model, path), export_policy_as_pkl(model, path)]
        try:
            exported.append(export_policy_as_onnx(model, path))
        except Exception as e:      # the ONNX exporter needs the `onnx` package
            print("ONNX export skipped:", type(e).__name__, e)
        print("Exported policy to: ", path)
    n = steps if steps is not None else 10 * int(env.max_episode_length)
    with torch.inference_mode():
        for _ in range(n):
            actions = policy(obs.detach())
            if FIX_COMMAND:
                env.commands[:, 0] = 1.0
                env.commands[:, 1] = 0.0
                env.commands[:, 2] = 0.0
            obs, _, rews, dones, infos = env.step(actions.detach())
    return env, exported


if __name__ == "__main__":
    play(get_args())
