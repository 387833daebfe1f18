#!/usr/bin/env python3
"""How often do non-foot contacts / joint-limit rows occur (per env and per 16-env workgroup)?  Guides the solver fast path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from go2_rl_gym_amd.envs import task_registry
from go2_rl_gym_amd.utils import get_args
task = sys.argv[1] if len(sys.argv) > 1 else "go2_flat"
args = get_args(["--task", task, "--num_envs", "4096", "--headless"])
env, _ = task_registry.make_env(task, args)
runner, _ = task_registry.make_alg_runner(env, task, args, log_root=None, use_graphs=False)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
runner.learn(iters, init_at_random_ep_len=True)
lo = torch.tensor([-1.0472, -1.5708, -2.7227] * 2 + [-1.0472, -0.5236, -2.7227] * 2, device=env.device)
hi = torch.tensor([1.0472, 3.4907, -0.83776] * 2 + [1.0472, 4.5379, -0.83776] * 2, device=env.device)
feet = [6, 10, 14, 18]; others = [i for i in range(19) if i not in feet]
acc = {"other_env": [], "other_wg": [], "lim_env": [], "lim_wg": [], "foot_env": []}
with torch.inference_mode():
    for _ in range(48):
        env.step(runner.alg.actor_critic.act(env.get_observations()))
        cf = env.contact_forces
        oth = (cf[:, others].norm(dim=-1) > 0).any(1)
        lim = ((env.dof_pos - lo < 0.05) | (hi - env.dof_pos < 0.05)).any(1)
        acc["other_env"].append(oth.float().mean().item()); acc["other_wg"].append(oth.view(-1, 16).any(1).float().mean().item())
        acc["lim_env"].append(lim.float().mean().item()); acc["lim_wg"].append(lim.view(-1, 16).any(1).float().mean().item())
        acc["foot_env"].append((cf[:, feet, 2] > 0).any(1).float().mean().item())
print(task, "after", iters, "iterations:", {k: round(float(np.mean(v)), 3) for k, v in acc.items()})
