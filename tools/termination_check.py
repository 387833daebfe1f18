#!/usr/bin/env python3
"""Does a fallen robot get terminated?  The contact model gives each leg TWO contact slots (foot + the deepest of the leg's other spheres and
its share of the base / head spheres), so a base sphere can be shadowed by a deeper thigh / calf sphere of the same slot.
This check steps 4096 envs under N(0, 1) actions for 600 steps and counts robots that stay low (base z < 0.12 m) without being reset:
on an MI355X none stays low for more than ~20 steps (0.4 s) — the base contact does win a slot once the robot is down.
   python tools/termination_check.py   (GPU)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch, ctypes as C
from helpers import DeviceSim, load_hip
hip = load_hip()
N = 4096
s = DeviceSim(hip, num_envs=N, seed=5)
s.reset_all()
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
low_steps = torch.zeros(N, device="cuda:0")
worst = 0
hist = []
for t in range(600):
    a = torch.randn(N, 12, device="cuda:0", generator=g) * 1.0
    hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
    root = s.t["root_states"]; z = root[:, 2]
    reset = s.t["reset_buf"].bool()
    low = (z < 0.12)
    low_steps = torch.where(low & ~reset, low_steps + 1, torch.zeros_like(low_steps))
    if t % 100 == 99:
        cf = s.t["contact_forces"]
        basef = cf[:, 0].norm(dim=1)
        print("t %d: resets/step %.1f, envs with base z<0.12: %d, lying >25 steps: %d (max %d), mean z %.3f, base force>1 among low: %d" %
              (t, float(reset.float().sum()), int(low.sum()), int((low_steps > 25).sum()), int(low_steps.max()), float(z.mean()), int(((basef > 1) & low).sum())))
