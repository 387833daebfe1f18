#!/usr/bin/env python3
"""The one-step parity protocol of tests/test_gpu_parity.py::test_one_step_parity_vs_oracle (each step starts from the fp32 oracle's state) at several solver sweep
counts: per tensor the largest HIP-vs-oracle difference over the well-conditioned env-steps, the env-step it occurs at and how far the fp32 oracle is from the
fp64 oracle there.     python tools/sweeps_parity.py [sweeps ...]  > profiles/r6_sweeps_parity.txt      (GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import PLANE_BOUND, STEP_STATE, DeviceSim, HostSim, ill_conditioned_envs, load_hip, load_hip_precise, load_oracle

N, steps = 64, int(os.environ.get("STEPS", 100))
KEYS = ("root_states", "dof_state", "torques", "obs_buf", "rew_buf")
print("one-step parity, %d env-steps per row; bounds %s" % (N * steps, {k: PLANE_BOUND[k] for k in KEYS}))
for it_ in [int(x) for x in sys.argv[1:]] or [4, 6, 8]:
    kw = dict(solver_iterations=it_)
    so, s64 = HostSim(load_oracle(), num_envs=N, **kw), HostSim(load_oracle(f64=True), num_envs=N, **kw)
    sims = {"hip": DeviceSim(load_hip(), num_envs=N, **kw), "hip_precise": DeviceSim(load_hip_precise(), num_envs=N, **kw)}
    for s in [so, s64] + list(sims.values()):
        s.reset_all()
    rng = np.random.default_rng(0)
    worst = {b: {k: (0.0, -1, -1, 0.0) for k in KEYS} for b in sims}
    skipped = 0
    for it in range(steps):
        a = rng.normal(0, 1, (N, 12)).astype(np.float32)
        for k in STEP_STATE:
            v = np.asarray(getattr(so, k))
            getattr(s64, k)[...] = v
            for sd in sims.values():
                getattr(sd, k)[...] = v
        so.step(a); s64.step(a.astype(np.float64))
        for sd in sims.values():
            sd.step(a)
        ok = ~ill_conditioned_envs(so, s64); skipped += int((~ok).sum())
        for k in KEYS:
            ref = np.asarray(getattr(so, k), np.float64)
            d64 = np.abs(ref - np.asarray(getattr(s64, k), np.float64)).reshape(N, -1).max(1)
            for b, sd in sims.items():
                d = np.abs(ref - np.asarray(getattr(sd, k), np.float64)).reshape(N, -1).max(1)
                d = np.where(ok, d, 0.0)
                e = int(d.argmax())
                if d[e] > worst[b][k][0]:
                    worst[b][k] = (float(d[e]), it, e, float(d64[e]))
    print("sweeps %d   (%d ill-conditioned env-steps left out)" % (it_, skipped))
    for b in sims:
        print("   %-12s" % b + "  ".join("%s %.2e (step %d env %d; oracle32-64 there %.1e)" % (k, *worst[b][k]) for k in KEYS))
    for s in [so, s64] + list(sims.values()):
        s.close()
