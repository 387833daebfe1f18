#!/usr/bin/env python3
"""HIP vs oracle one-step differences under reduced configurations (localise a device-only discrepancy)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import STEP_STATE, DeviceSim, HostSim, load_hip, load_oracle
import ctypes as C
from go2_rl_gym_amd import _abi
N = 64
def lib_():
    import torch; torch.cuda.is_available()      # torch's HIP runtime first (go2_rl_gym_amd/_lib.py)
    if os.environ.get("GO2_DEBUG_LIB"):
        return _abi.bind(os.environ["GO2_DEBUG_LIB"], C.c_float)
    return load_hip()
for name, kw in (("default", {}),):
    so, sd = HostSim(load_oracle(), num_envs=N, **kw), DeviceSim(lib_(), num_envs=N, **kw)
    so.reset_all(); sd.reset_all()
    rng = np.random.default_rng(0)
    worst = {}
    for it in range(40):
        a = rng.normal(0, 1, (N, 12)).astype(np.float32)
        for k in STEP_STATE:
            getattr(sd, k)[...] = np.asarray(getattr(so, k))
        so.step(a); sd.step(a)
        for k in ("root_states", "dof_state", "torques", "contact_forces", "obs_buf"):
            d = np.abs(np.asarray(getattr(so, k), np.float64) - np.asarray(getattr(sd, k), np.float64)).reshape(N, -1).max(1)
            if it in (0, 1, 2, 5, 10, 39):
                worst.setdefault(k, []).append("%d:%.1e(env %d)" % (it, d.max(), int(d.argmax())))
    print(name)
    for k, v in worst.items():
        print("   %-16s %s" % (k, "  ".join(v)))
    so.close(); sd.close()
