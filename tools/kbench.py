#!/usr/bin/env python3
"""Kernel micro-benchmark: time go2_step_kernel alone (HIP events inside the library) for a few configurations.
   python tools/kbench.py [N]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import DeviceSim, load_hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hip = load_hip()


def run(label, steps=300, settle=80, **kw):
    s = DeviceSim(hip, num_envs=N, **kw)
    s.reset_all()
    a = torch.randn(N, 12, device="cuda:0") * 0.5
    for _ in range(settle):
        hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
    torch.cuda.synchronize()
    hip.go2sim_enable_timing(s.h, 1)
    for _ in range(steps):
        hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
    ms, n = C.c_double(), C.c_int64()
    hip.go2sim_kernel_time(s.h, C.byref(ms), C.byref(n))
    k = ms.value / n.value
    fz = float((s.t["contact_forces"][:, [6, 10, 14, 18], 2] > 1).float().mean())
    print("%-34s kernel %.1f us  -> %.2f M env-steps/s kernel-only, %.1f GB/s algorithmic (%.3f%% of 8 TB/s); feet in contact %.2f" %
          (label, k * 1e3, N / k / 1e3, 2936 * N / k / 1e6, 2936 * N / k / 1e6 / 80.0, fz))
    s.close()
    return k


def run_split(steps=300, **kw):
    s = DeviceSim(hip, num_envs=N, **kw)
    s.reset_all()
    a = torch.randn(N, 12, device="cuda:0") * 0.5
    for _ in range(80):
        hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
    for name, fn in (("PHYS only (go2sim_simulate)", hip.go2sim_simulate), ("POST only (go2sim_post_physics)", hip.go2sim_post_physics)):
        torch.cuda.synchronize(); hip.go2sim_enable_timing(s.h, 1)
        for _ in range(steps):
            fn(s.h, s._st())
        ms, n = C.c_double(), C.c_int64(); hip.go2sim_kernel_time(s.h, C.byref(ms), C.byref(n))
        print("%-34s kernel %.1f us" % (name, 1e3 * ms.value / n.value))
    s.close()


if __name__ == "__main__":
    run_split()
    run("default (8 PGS sweeps, 4 substeps)")
    run("no PGS sweeps", solver_iterations=0)
    run("4 PGS sweeps", solver_iterations=4)
    run("1 substep", decimation=1)
    run("1 substep, no PGS", decimation=1, solver_iterations=0)
