"""Task registry with the reference's surface (legged_gym/utils/task_registry.py:15-128)."""
import copy
import os
from datetime import datetime

from ..rsl_rl.runners import OnPolicyRunner, OnPolicyRunnerCTS
from .helpers import class_to_dict, get_args, get_load_path, parse_sim_params, set_seed, update_cfg_from_args

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_RUNNERS = {"OnPolicyRunner": OnPolicyRunner, "OnPolicyRunnerCTS": OnPolicyRunnerCTS}


class TaskRegistry:
    def __init__(self):
        self.task_classes, self.env_cfgs, self.train_cfgs = {}, {}, {}

    def register(self, name, task_class, env_cfg, train_cfg):
        self.task_classes[name], self.env_cfgs[name], self.train_cfgs[name] = task_class, env_cfg, train_cfg

    def get_task_class(self, name):
        return self.task_classes[name]

    def get_cfgs(self, name):
        # copies: the reference hands out the registered instances themselves (task_registry.py:24-28), so what one caller changes
        # (play.py: resume, noise off, fewer envs; --resume from the CLI) silently carries over to every later make_env /
        # make_alg_runner of the process.  Scripts that edit the returned objects and pass them back in behave the same.
        train_cfg, env_cfg = copy.deepcopy(self.train_cfgs[name]), copy.deepcopy(self.env_cfgs[name])
        env_cfg.seed = train_cfg.seed
        return env_cfg, train_cfg

    def make_env(self, name, args=None, env_cfg=None, **env_kwargs):
        if args is None:
            args = get_args()
        if name not in self.task_classes:
            raise ValueError(f"Task with name: {name} was not registered")
        task_class = self.get_task_class(name)
        if env_cfg is None:
            env_cfg, _ = self.get_cfgs(name)
        env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
        set_seed(env_cfg.seed)
        sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
        env = task_class(cfg=env_cfg, sim_params=sim_params, physics_engine=args.physics_engine, sim_device=args.sim_device, headless=args.headless, **env_kwargs)
        return env, env_cfg

    def make_alg_runner(self, env, name=None, args=None, train_cfg=None, log_root="default", **runner_kwargs):
        if args is None:
            args = get_args()
        if train_cfg is None:
            if name is None:
                raise ValueError("Either 'name' or 'train_cfg' must be not None")
            _, train_cfg = self.get_cfgs(name)
        elif name is not None:
            print(f"'train_cfg' provided -> Ignoring 'name={name}'")
        _, train_cfg = update_cfg_from_args(None, train_cfg, args)
        if log_root == "default":
            log_root = os.path.join(ROOT_DIR, "logs", train_cfg.runner.experiment_name)
            log_dir = os.path.join(log_root, datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg.runner.run_name)
        elif log_root is None:
            log_dir = None
        else:
            log_dir = os.path.join(log_root, datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg.runner.run_name)
        runner = _RUNNERS[train_cfg.runner_class_name](env, class_to_dict(train_cfg), log_dir, device=args.rl_device, **runner_kwargs)
        if train_cfg.runner.resume:
            resume_path = get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
            print(f"Loading model from: {resume_path}")
            runner.load(resume_path)
        return runner, train_cfg


task_registry = TaskRegistry()
