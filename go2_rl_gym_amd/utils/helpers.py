"""CLI / config helpers with the reference's surface (legged_gym/utils/helpers.py): class_to_dict :12-27,
set_seed :38-48, parse_sim_params :50-72, get_load_path :74-97, update_cfg_from_args :99-126, get_args :128-157
(including the flags Isaac Gym's gymutil.parse_arguments adds: --sim_device --pipeline --graphics_device_id
--physx/--flex --num_threads --subscenes --slices)."""
import argparse
import os
import random
from pathlib import Path

import numpy as np
import torch


def _public_attrs(obj):
    return ((k, getattr(obj, k)) for k in dir(obj) if not k.startswith("_"))


def class_to_dict(obj) -> dict:
    """Nested config classes -> plain nested dicts (what the runner dumps to config.yaml).  Leaves (anything without a __dict__) pass
    through; lists are converted element-wise.  Reference surface: legged_gym/utils/helpers.py class_to_dict."""
    if not hasattr(obj, "__dict__"):
        return obj
    conv = lambda v: list(map(class_to_dict, v)) if isinstance(v, list) else class_to_dict(v)
    return {k: conv(v) for k, v in _public_attrs(obj)}


def update_class_from_dict(obj, d):
    """Inverse direction: write the entries of a (nested) dict into the matching (nested) config classes."""
    for key, val in d.items():
        target = getattr(obj, key, None)
        if isinstance(target, type):
            update_class_from_dict(target, val)
            continue
        setattr(obj, key, val)


def set_seed(seed):
    """Seed every generator a run touches (python, numpy, torch CPU + all GPUs); -1 draws one.  -> the seed used."""
    seed = int(np.random.randint(0, 10000)) if seed == -1 else seed
    print(f"Setting seed: {seed}")
    os.environ["PYTHONHASHSEED"] = str(seed)
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


class SimParams:
    """What the reference passes around as gymapi.SimParams: only dt / substeps / gravity / pipeline flag matter here."""

    def __init__(self, dt=0.005, substeps=1, use_gpu_pipeline=True, gravity=(0.0, 0.0, -9.81)):
        self.dt, self.substeps, self.use_gpu_pipeline, self.gravity = dt, substeps, use_gpu_pipeline, list(gravity)


def parse_sim_params(args, cfg):
    sp = SimParams(use_gpu_pipeline=getattr(args, "use_gpu_pipeline", True))
    sim = cfg.get("sim", {})
    sp.dt = sim.get("dt", sp.dt)
    sp.substeps = sim.get("substeps", sp.substeps)
    sp.gravity = list(sim.get("gravity", sp.gravity))
    return sp


def _checkpoint_number(path):
    stem = path.stem.split("_", 1)[-1]
    return int(stem) if stem.isdigit() else -1


def get_load_path(root, load_run=-1, checkpoint=-1):
    """logs/<experiment>/<run>/model_<it>.pt to resume from: the last run directory (by name) that holds checkpoints unless `load_run`
    names one, and its highest-numbered checkpoint unless `checkpoint` names one (legged_gym/utils/helpers.py get_load_path)."""
    root = Path(root) if root is not None else None
    if load_run == -1:
        runs = sorted(d.name for d in root.iterdir() if d.is_dir() and d.name != "exported" and any(d.glob("model_*.pt"))) if root is not None and root.is_dir() else []
        if not runs:
            raise ValueError("No runs in this directory: %s" % (root,))
        run_dir = root / runs[-1]
    else:
        run_dir = root / load_run
    if checkpoint != -1:
        return str(run_dir / ("model_%s.pt" % checkpoint))
    saved = sorted(run_dir.glob("model*"), key=lambda f: (_checkpoint_number(f), f.name))
    if not saved:
        raise ValueError("No checkpoints in %s" % (run_dir,))
    return str(saved[-1])


# command-line argument -> attribute of train_cfg.runner it overrides when given
_RUNNER_OVERRIDES = ("max_iterations", "experiment_name", "run_name", "load_run", "checkpoint")


def update_cfg_from_args(env_cfg, cfg_train, args):
    """Command-line overrides onto the config objects (legged_gym/utils/helpers.py update_cfg_from_args): --num_envs onto the env config;
    --seed, --resume and the runner fields above onto the train config; the RoboGauge switches when the train config has that section."""
    if env_cfg is not None and args.num_envs is not None:
        env_cfg.env.num_envs = args.num_envs
    if cfg_train is None:
        return env_cfg, cfg_train
    if args.seed is not None:
        cfg_train.seed = args.seed
    if args.resume:
        cfg_train.runner.resume = True
    for name in _RUNNER_OVERRIDES:
        given = getattr(args, name)
        if given is not None:
            setattr(cfg_train.runner, name, given)
    gauge = getattr(cfg_train, "robogauge", None)
    if gauge is not None:
        for arg_name, field in (("robogauge", "enabled"), ("robogauge_port", "port")):
            given = getattr(args, arg_name, None)
            if given is not None:
                setattr(gauge, field, given)
    return env_cfg, cfg_train


def get_args(argv=None):
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="go2_flat")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False, help="accepted and ignored, as in the reference")
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    p.add_argument("--robogauge", action="store_true", default=False)
    p.add_argument("--robogauge_port", type=int, default=9973)
    # flags contributed by isaacgym.gymutil.parse_arguments in the reference
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    g = p.add_mutually_exclusive_group()
    g.add_argument("--flex", action="store_true")
    g.add_argument("--physx", action="store_true")
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--slices", type=int, default=0)
    args = p.parse_args(argv)
    dev = args.sim_device
    args.sim_device_type = dev.split(":")[0]
    args.compute_device_id = int(dev.split(":")[1]) if ":" in dev else 0
    args.use_gpu = args.sim_device_type == "cuda"
    args.use_gpu_pipeline = args.pipeline in ("gpu", "cuda") and args.use_gpu
    args.physics_engine = 1                      # SIM_PHYSX in the reference; this build has one engine
    args.sim_device_id = args.compute_device_id  # name alignment (:153-156)
    args.sim_device = args.sim_device_type + (f":{args.sim_device_id}" if args.sim_device_type == "cuda" else "")
    return args
