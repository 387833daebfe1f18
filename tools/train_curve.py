#!/usr/bin/env python3
"""Learning-curve check: train a task for N iterations on the GPU and print reward / episode length / tracking terms every K iterations.
   python tools/train_curve.py [task] [iterations] [every] [seed]"""
import os, sys, tempfile, io, contextlib, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from go2_rl_gym_amd.envs import task_registry
from go2_rl_gym_amd.utils import get_args
task = sys.argv[1] if len(sys.argv) > 1 else "go2_flat"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
every = int(sys.argv[3]) if len(sys.argv) > 3 else 50
seed = sys.argv[4] if len(sys.argv) > 4 else "1"
args = get_args(["--task", task, "--num_envs", "4096", "--headless", "--seed", seed])
# (as in the reference, --seed only reaches the TRAIN config — update_cfg_from_args, legged_gym/utils/helpers.py:99-126 — while make_env seeds every generator and the
#  simulator from env_cfg.seed, which get_cfgs copied from the registered train config before the CLI was read: a seed study has to set both itself)
env_cfg, train_cfg = task_registry.get_cfgs(task)
env_cfg.seed = train_cfg.seed = int(seed)
env, _ = task_registry.make_env(task, args, env_cfg=env_cfg)
runner, _ = task_registry.make_alg_runner(env, None, args, train_cfg=train_cfg, log_root=tempfile.mkdtemp())
env.common_step_counter = 0
env.update_reward_curriculum(force_update=True)
done = 0
while done < iters:
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runner.learn(every, init_at_random_ep_len=(done == 0))
    done += every
    txt = buf.getvalue()
    last = txt[txt.rfind("Learning iteration"):]
    pick = lambda key: (re.findall(key + r"\s*(-?[\d.]+)", last) or ["nan"])[-1]
    names = ("Mean reward:", "Mean teacher reward:", "Mean student reward:", "Mean episode length:", "Mean teacher episode length:", "Mean student episode length:", "Mean episode rew_tracking_lin_vel:", "Mean episode rew_tracking_ang_vel:", "Mean episode terrain_level_all:", "Mean episode terrain_level_stairs_up:", "Mean action noise std:", "Computation:")
    print("it %4d | " % done + " | ".join("%s %s" % (n.replace("Mean ", "").replace("episode ", "").rstrip(":"), pick(re.escape(n))) for n in names if pick(re.escape(n)) != "nan"), flush=True)
env.close()
