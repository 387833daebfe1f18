#!/usr/bin/env python3
"""Does a fallen robot get terminated, and how soon?  (check_termination, legged_robot.py:170-173: ||F_base|| > 1.)  Steps 4096 envs under
N(0, 1) actions for 600 steps on the plane and records, for every episode that ends by termination, for how many consecutive steps the base
had been LOW (z < 0.12 m: the trunk box, half height 0.057, touches or is about to) before the reset — the latency a contact model that
under-reports base contacts would show — plus how many robots lie low without a reset, and how many of the 8 penalised bodies
(thigh / calf, _reward_collision :1277-1279) report a force among the low robots.
The last line counts, over ALL env-steps, how often at least one penalised body reports more than 0.1 N (the threshold of _reward_collision) — the
number a change of the collision samples (round 4: mid-segment samples of the leg capsules / thigh box edges) moves.
   python tools/termination_check.py [tag [library.so|- [rough]]]   (GPU; rough: the task=go2 trimesh terrain instead of the plane)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes as C
from helpers import DeviceSim, load_hip
if len(sys.argv) > 2 and sys.argv[2] != "-":      # another build of the library (e.g. build/variants/r2model_1wave.so: the round-2 contact model, two slots per leg)
    from go2_rl_gym_amd import _abi
    hip = _abi.bind(os.path.abspath(sys.argv[2]), C.c_float)
else:
    hip = load_hip()
N = 4096
TERRAIN = {}
if len(sys.argv) > 3 and sys.argv[3] == "rough":
    from helpers import heightfield_overrides
    TERRAIN = heightfield_overrides(N, mesh_type="trimesh")[1]
s = DeviceSim(hip, num_envs=N, seed=5, **TERRAIN)
s.reset_all()
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
low_steps = torch.zeros(N, device="cuda:0")
lat, ncoll_low, nbody_low = [], [], []
pen_any = torch.zeros((), device="cuda:0"); pen_sum = torch.zeros((), device="cuda:0")
PEN = [4, 5, 8, 9, 12, 13, 16, 17]
for t in range(600):
    a = torch.randn(N, 12, device="cuda:0", generator=g) * 1.0
    hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
    root = s.t["root_states"]; z = root[:, 2]
    reset = s.t["reset_buf"].bool(); tout = s.t["time_out_buf"].bool()
    cf = s.t["contact_forces"]
    term = reset & ~tout
    hit = cf[:, PEN].norm(dim=-1) > 0.1
    pen_any += hit.any(1).float().sum(); pen_sum += hit.float().sum()
    # (root_states of a reset env is already the new pose: the low-step counter of the steps BEFORE is what counts)
    lat.append(low_steps[term].cpu().numpy())
    low = (z < 0.12) & ~reset
    low_steps = torch.where(low, low_steps + 1, torch.zeros_like(low_steps))
    if low.any():
        ncoll_low.append((cf[low][:, PEN].norm(dim=-1) > 0.1).sum(1).float().cpu().numpy())
        nbody_low.append((cf[low].norm(dim=-1) > 0.1).sum(1).float().cpu().numpy())
    if t % 100 == 99:
        basef = cf[:, 0].norm(dim=1)
        print("t %d: resets/step %.1f, envs with base z<0.12: %d, lying >25 steps: %d (max %d), mean z %.3f, base force>1 among low: %d" %
              (t, float(reset.float().sum()), int(low.sum()), int((low_steps > 25).sum()), int(low_steps.max()), float(z.mean()), int(((basef > 1) & low).sum())))
lat = np.concatenate(lat); nc = np.concatenate(ncoll_low) if ncoll_low else np.zeros(1); nb = np.concatenate(nbody_low) if nbody_low else np.zeros(1)
q = lambda x, p: float(np.quantile(x, p)) if len(x) else float("nan")
print("%s: %d terminations in 600 steps x %d envs; steps the base was low before its termination: p50 %.0f  p90 %.0f  p99 %.0f  max %.0f (x 0.02 s)" %
      (sys.argv[1] if len(sys.argv) > 1 else "library", len(lat), N, q(lat, .5), q(lat, .9), q(lat, .99), lat.max() if len(lat) else 0))
print("   among low robots: penalised bodies (of 8) with ||F|| > 0.1: mean %.2f  p90 %.0f  max %.0f;  bodies (of 19) in contact: mean %.2f  max %.0f" %
      (nc.mean(), q(nc, .9), nc.max(), nb.mean(), nb.max()))
print("   over all %d env-steps: %.1f per 1000 have at least one penalised body (thigh / calf) over 0.1 N; %.1f penalised-body contacts per 1000 env-steps" %
      (600 * N, float(pen_any) / (600 * N) * 1000, float(pen_sum) / (600 * N) * 1000))
