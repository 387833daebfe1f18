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


def run_split(steps=300, label="", **kw):
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
        print("%-34s kernel %.1f us   %s" % (name, 1e3 * ms.value / n.value, label))
    s.close()


if __name__ == "__main__":
    run_split()
    run_split(label="measure_heights=0", measure_heights=0)
    run_split(label="add_noise=0", add_noise=0)
    run_split(label="no DR/push/delay", push_robots=0, randomize_action_delay=0, add_noise=0, measure_heights=0)
    run("default (4 PGS sweeps, 4 substeps)")
    run("no PGS sweeps", solver_iterations=0)
    run("4 PGS sweeps", solver_iterations=4)
    run("1 substep", decimation=1)
    run("1 substep, no PGS", decimation=1, solver_iterations=0)


def phase_clocks(steps=50, **kw):
    """Per-wave phase timestamps (wall_clock64, 100 MHz) of the fused kernel: mean and max over the waves."""
    s = DeviceSim(hip, num_envs=N, **kw)
    s.reset_all()
    a = torch.randn(N, 12, device="cuda:0") * 0.5
    for _ in range(80):
        hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
    nb = 4 * ((N + 15) // 16)               # one row of 8 stamps per WAVE (4 waves per 16-env workgroup)
    buf = torch.zeros(nb, 8, dtype=torch.int64, device="cuda:0")
    hip.go2sim_debug_clock.argtypes = [C.c_void_p, C.c_void_p]
    hip.go2sim_debug_clock(s.h, C.c_void_p(buf.data_ptr()))
    acc = []
    for _ in range(steps):
        hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
        torch.cuda.synchronize()
        acc.append(buf[:, :6].cpu().numpy().astype(np.float64))
    hip.go2sim_debug_clock(s.h, None)
    t = np.stack(acc)                       # [steps, blocks, 6]
    d = np.diff(t, axis=2) / 100.0          # us (100 MHz constant clock)
    names = ["load", "4 substeps", "finish/FK", "postA", "postB"]
    span = (t[:, :, 5].max(1) - t[:, :, 0].min(1)) / 100.0
    print("phase           mean-over-waves   max-over-waves (mean over steps)")
    for i, nm in enumerate(names):
        print("%-14s %10.1f us %14.1f us" % (nm, d[:, :, i].mean(), d[:, :, i].max(1).mean()))
    print("first start -> last end: %.1f us" % span.mean())
    s.close()


if __name__ == "__main__":
    phase_clocks()
