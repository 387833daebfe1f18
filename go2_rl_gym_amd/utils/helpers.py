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


def class_to_dict(obj) -> dict:
    if not hasattr(obj, "__dict__"):
        return obj
    result = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        result[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return result


def update_class_from_dict(obj, d):
    for key, val in d.items():
        attr = getattr(obj, key, None)
        if isinstance(attr, type):
            update_class_from_dict(attr, val)
        else:
            setattr(obj, key, val)


def set_seed(seed):
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
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


def get_load_path(root, load_run=-1, checkpoint=-1):
    try:
        runs = [r for r in os.listdir(root) if len(list((Path(root) / r).glob("model_*.pt"))) > 0]
        runs.sort()
        if "exported" in runs:
            runs.remove("exported")
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: %s" % (root,))
    load_run = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = [f for f in os.listdir(load_run) if "model" in f]
        models.sort(key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)


def update_cfg_from_args(env_cfg, cfg_train, args):
    if env_cfg is not None and args.num_envs is not None:
        env_cfg.env.num_envs = args.num_envs
    if cfg_train is not None:
        if args.seed is not None:
            cfg_train.seed = args.seed
        if args.max_iterations is not None:
            cfg_train.runner.max_iterations = args.max_iterations
        if args.resume:
            cfg_train.runner.resume = args.resume
        if args.experiment_name is not None:
            cfg_train.runner.experiment_name = args.experiment_name
        if args.run_name is not None:
            cfg_train.runner.run_name = args.run_name
        if args.load_run is not None:
            cfg_train.runner.load_run = args.load_run
        if args.checkpoint is not None:
            cfg_train.runner.checkpoint = args.checkpoint
        if getattr(args, "robogauge", None) is not None and hasattr(cfg_train, "robogauge"):
            cfg_train.robogauge.enabled = args.robogauge
        if getattr(args, "robogauge_port", None) is not None and hasattr(cfg_train, "robogauge"):
            cfg_train.robogauge.port = args.robogauge_port
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
