#!/usr/bin/env python3
"""A training run over the reference's own horizon (go2_config.py: max_iterations = 150000) on one MI355X, then the trained student policy next to the policy the
reference ships (trained for 150 k iterations in Isaac Gym; tests/golden/pretrained_go2_cts_150k.npz) under the walking protocol of tests/test_export.py:
1 m/s forward command, plane and rough curriculum map, 80 robots, 8 s.
   python tools/long_run.py [task] [iterations] [every] [seed] [wall budget, s]      -> stdout; the final checkpoint goes to gpurun_out/long_run_<task>.pt"""
import contextlib
import io
import math
import os
import re
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from go2_rl_gym_amd.envs import task_registry
from go2_rl_gym_amd.utils import get_args

task = sys.argv[1] if len(sys.argv) > 1 else "go2_cts"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
seed = sys.argv[4] if len(sys.argv) > 4 else "1"
budget = float(sys.argv[5]) if len(sys.argv) > 5 else float("inf")          # wall seconds for the training part (a gpurun call is limited to 60 minutes): stop at the next checkpoint line past it
args = get_args(["--task", task, "--num_envs", "4096", "--headless", "--seed", seed])
env_cfg, train_cfg = task_registry.get_cfgs(task)
env_cfg.seed = train_cfg.seed = int(seed)
train_cfg.runner.save_interval = 10 ** 9          # (one checkpoint, at the end)
env, _ = task_registry.make_env(task, args, env_cfg=env_cfg)
logdir = tempfile.mkdtemp()
runner, _ = task_registry.make_alg_runner(env, None, args, train_cfg=train_cfg, log_root=logdir)
env.common_step_counter = 0
env.update_reward_curriculum(force_update=True)
model = runner.alg.model if hasattr(runner.alg, "model") else runner.alg.actor_critic          # (CTS family / PPO)
print("task %s, %d iterations of 24 x 4096 env-steps, seed %s" % (task, iters, seed), flush=True)
done, t0 = 0, time.time()
names = ("Mean reward:", "Mean teacher reward:", "Mean student reward:", "Mean episode length:", "Mean teacher episode length:", "Mean student episode length:",
         "Mean episode rew_tracking_lin_vel:", "Mean episode rew_tracking_ang_vel:", "Mean episode terrain_level_all:", "Mean episode terrain_level_stairs_up:", "Mean action noise std:")
while done < iters:
    n = min(every, iters - done)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runner.learn(n, init_at_random_ep_len=(done == 0))
    done += n
    txt = buf.getvalue()
    last = txt[txt.rfind("Learning iteration"):]
    pick = lambda key: (re.findall(key + r"\s*(-?[\d.]+|nan|inf)", last) or ["-"])[-1]
    vals = {k: pick(re.escape(k)) for k in names}
    finite = all(torch.isfinite(p).all().item() for p in model.parameters())
    print("it %6d | " % done + " | ".join("%s %s" % (k.replace("Mean ", "").replace("episode ", "").rstrip(":"), v) for k, v in vals.items() if v != "-")
          + " | wall %.0f s | weights finite: %s" % (time.time() - t0, finite), flush=True)
    if not finite:
        print("STOP: non-finite weights"); break
    if time.time() - t0 > budget:
        print("(wall budget of %.0f s reached: stopping the training here)" % budget, flush=True); break
wall = time.time() - t0
print("trained %d iterations = %.3g env-steps in %.1f s wall (%.2f M env-steps/s including logging and the curve's parsing)" % (done, done * 24 * 4096.0, wall, done * 24 * 4096 / wall / 1e6), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
ck = os.path.join(ROOT, "gpurun_out", "long_run_%s.pt" % task)
torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, ck)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
import copy
model_cpu = copy.deepcopy(model).cpu()
env.close()

# ---- the walking protocol (tests/test_export.py:run_pretrained_walk) for both policies on the HIP simulator --------------------------------------------------------
from helpers import DeviceSim, heightfield_overrides, load_hip
from test_export import pretrained_policy
from go2_rl_gym_amd.rsl_rl.modules.actor_critic_cts import ActorCriticCTS


def walk(m, sim, seconds=8.0):
    N = sim.N
    m.history = torch.zeros(N, 5, 45)
    sim.reset_all()
    a = np.zeros((N, 12), np.float32)
    sim.step(a)
    resets, speeds, zmin = 0, [], 1.0
    steps = int(seconds / 0.02)
    for it in range(steps):
        sim.commands[:, :3] = np.tile(np.array([1.0, 0.0, 0.0], np.float32), (N, 1))
        obs = np.asarray(sim.obs_buf).copy()
        obs[:, 6:9] = np.array([2.0, 0.0, 0.0], np.float32)
        with torch.no_grad():
            a = m.act_inference(torch.from_numpy(obs)).numpy()
        sim.step(a)
        resets += int(np.asarray(sim.reset_buf).sum())
        if it > steps // 3:
            speeds.append(np.asarray(sim.base_lin_vel)[:, 0].mean()); zmin = min(zmin, float(np.asarray(sim.root_states)[:, 2].min()))
    return float(np.mean(speeds)), zmin, resets


if True:
    ref, _ = pretrained_policy()
    if task in ("go2_cts", "go2_flat_cts"):
        ours = ActorCriticCTS(45, 263, 12, 1, 5)
        missing, unexpected = ours.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
    else:
        ours = model_cpu          # (a plain ActorCritic: act_inference(obs) = the actor's mean)
    hip = load_hip()
    for terrain in ("plane", "rough curriculum map"):
        for name, m in (("the reference's shipped policy (150 k iterations in Isaac Gym)", ref), ("the policy trained here (%d iterations)" % done, ours)):
            N = 80
            ov = heightfield_overrides(N)[1] if terrain != "plane" else {}
            s = DeviceSim(hip, num_envs=N, push_robots=0, add_noise=0, **ov)
            v, zmin, resets = walk(m, s)
            s.close()
            print("walk, 1 m/s command, %-22s %-66s mean forward speed %.3f m/s, lowest base height %.3f m, resets %d of %d robots x 8 s" % (terrain + ":", name, v, zmin, resets, N), flush=True)
