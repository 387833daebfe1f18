"""python -m go2_rl_gym_amd.scripts.train --task go2_flat --num_envs 4096 --headless
Same three-call flow as legged_gym/scripts/train.py:11-16."""
from go2_rl_gym_amd.envs import *  # noqa: F401,F403
from go2_rl_gym_amd.utils import get_args
from go2_rl_gym_amd.utils.task_registry import task_registry


def train(args):
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args)
    env.common_step_counter = runner.current_learning_iteration * env.num_steps_per_env   # resume the env's curriculum clock
    env.update_reward_curriculum(force_update=True)
    runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)


if __name__ == "__main__":
    train(get_args())
