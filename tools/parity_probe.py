#!/usr/bin/env python3
"""Error distribution of the one-step physics parity protocol (each step starts from the fp32 oracle's state): the shipped HIP build,
the HIP build without -ffast-math, and the fp32 oracle itself, all against the fp64 oracle / the fp32 oracle.  Plane and rough terrain.
   python tools/parity_probe.py [N] [steps]     -> gpurun_out/parity_probe.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import STEP_STATE, DeviceSim, HostSim, heightfield_overrides, load_hip, load_hip_precise, load_oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
KEYS = ("root_states", "dof_state", "torques", "obs_buf", "privileged_obs_buf", "rew_buf", "contact_forces", "measured_heights")
out = {}
for terrain in ("plane", "heightfield"):
    ov = heightfield_overrides(N)[1] if terrain == "heightfield" else {}
    so, s64 = HostSim(load_oracle(), num_envs=N, **ov), HostSim(load_oracle(f64=True), num_envs=N, **ov)
    sims = {"hip_fast_math": DeviceSim(load_hip(), num_envs=N, **ov), "hip_precise": DeviceSim(load_hip_precise(), num_envs=N, **ov)}
    for s in [so, s64] + list(sims.values()):
        s.reset_all()
    rng = np.random.default_rng(0)
    acc = {b: {k: [] for k in KEYS} for b in list(sims) + ["oracle32_vs_oracle64"]}
    resets = 0
    for it in range(steps):
        a = rng.normal(0, 1.0 if terrain == "plane" else 0.6, (N, 12)).astype(np.float32)
        for k in STEP_STATE:
            v = np.asarray(getattr(so, k))
            getattr(s64, k)[...] = v
            for sd in sims.values():
                getattr(sd, k)[...] = v
        so.step(a); s64.step(a.astype(np.float64))
        for sd in sims.values():
            sd.step(a)
        resets += int(np.asarray(so.reset_buf).sum())
        same_reset = {b: bool((np.asarray(sd.reset_buf) == np.asarray(so.reset_buf)).all()) for b, sd in sims.items()}
        assert all(same_reset.values()), (it, same_reset)
        for k in KEYS:
            ref = np.asarray(getattr(so, k), np.float64)
            acc["oracle32_vs_oracle64"][k].append(np.abs(ref - np.asarray(getattr(s64, k), np.float64)).reshape(N, -1).max(1))
            for b, sd in sims.items():
                acc[b][k].append(np.abs(ref - np.asarray(getattr(sd, k), np.float64)).reshape(N, -1).max(1))
    rep = {}
    for b, d in acc.items():
        rep[b] = {}
        for k, v in d.items():
            v = np.concatenate(v)
            rep[b][k] = {"max": float(v.max()), "p50": float(np.quantile(v, 0.5)), "p99": float(np.quantile(v, 0.99)), "p999": float(np.quantile(v, 0.999)),
                         "top5": [float(x) for x in np.sort(v)[-5:]]}
    rep["env_steps"], rep["resets"] = N * steps, resets
    out[terrain] = rep
    for s in [so, s64] + list(sims.values()):
        s.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_probe.json"), "w"), indent=1)
for t, rep in out.items():
    print(t, "env-steps", rep["env_steps"], "resets", rep["resets"])
    for b in ("hip_fast_math", "hip_precise", "oracle32_vs_oracle64"):
        print("  ", b)
        for k, v in rep[b].items():
            print("     %-20s p50 %.2e  p99 %.2e  p99.9 %.2e  max %.2e" % (k, v["p50"], v["p99"], v["p999"], v["max"]))
