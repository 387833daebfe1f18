"""python -m go2_rl_gym_amd.scripts.play --task go2_flat  (legged_gym/scripts/play.py:15-62, headless)."""
import torch

from go2_rl_gym_amd.envs import *  # noqa: F401,F403
from go2_rl_gym_amd.utils import get_args
from go2_rl_gym_amd.utils.task_registry import task_registry


def play(args, steps=None):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 100)
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = False
    env_cfg.domain_rand.randomize_friction = False
    env_cfg.domain_rand.push_robots = False
    env_cfg.env.test = True
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = True
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
    policy = runner.get_inference_policy(device=env.device)
    n = steps if steps is not None else 10 * int(env.max_episode_length)
    for _ in range(n):
        env.commands[:, 0] = 1.0
        env.commands[:, 1] = 0.0
        env.commands[:, 2] = 0.0
        actions = policy(obs.detach())
        obs, _, rews, dones, infos = env.step(actions.detach())
    return env


if __name__ == "__main__":
    play(get_args())
