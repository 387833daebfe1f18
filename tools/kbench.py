#!/usr/bin/env python3
"""Kernel micro-benchmark: time go2_step_kernel alone (HIP events inside the library) for a few configurations.
   python tools/kbench.py [N] [rough]        (rough: the task=go2 trimesh terrain instead of the plane)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import DeviceSim, load_hip, load_hip_kbench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hip = load_hip()
TERRAIN = {}
if len(sys.argv) > 2 and sys.argv[2] == "rough":
    from helpers import heightfield_overrides
    TERRAIN = heightfield_overrides(N, mesh_type="trimesh")[1]


def run(label, steps=300, settle=80, **kw):
    s = DeviceSim(hip, num_envs=N, **dict(TERRAIN, **kw))
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
    s = DeviceSim(hip, num_envs=N, **dict(TERRAIN, **kw))
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
    run("default (8 PGS sweeps, 4 substeps)")
    run("no PGS sweeps", solver_iterations=0)
    run("4 PGS sweeps", solver_iterations=4)
    run("6 PGS sweeps", solver_iterations=6)
    run("8 PGS sweeps", solver_iterations=8)
    run("16 PGS sweeps", solver_iterations=16)
    run("1 substep", decimation=1)
    run("1 substep, no PGS", decimation=1, solver_iterations=0)


def phase_clocks(steps=50, **kw):
    """Per-wave phase timestamps (wall_clock64, 100 MHz) of the fused kernel: mean and max over the waves.  Uses the STAMPED build of the
    library (build.py:build_hip_kbench, -DGO2_KBENCH_STAMPS); the timings above are of the product build, which carries no stamps."""
    hip = load_hip_kbench()
    s = DeviceSim(hip, num_envs=N, **dict(TERRAIN, **kw))
    s.reset_all()
    a = torch.randn(N, 12, device="cuda:0") * 0.5
    for _ in range(80):
        hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
    nb = 4 * ((N + 15) // 16)               # one row of 32 stamps per WAVE (4 waves per 16-env workgroup)
    buf = torch.zeros(nb, 32, dtype=torch.int64, device="cuda:0")
    hip.go2sim_debug_clock.argtypes = [C.c_void_p, C.c_void_p]
    hip.go2sim_debug_clock(s.h, C.c_void_p(buf.data_ptr()))
    acc = []
    for _ in range(steps):
        hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
        torch.cuda.synchronize()
        acc.append(buf[:, :32].cpu().numpy().astype(np.float64)); buf[:, 8:].zero_()
    hip.go2sim_debug_clock(s.h, None)
    t = np.stack(acc)                       # [steps, blocks, 6]
    d = np.diff(t[:, :, :6], axis=2) / 100.0          # us (100 MHz constant clock)
    names = ["load", "4 substeps", "finish/FK", "postA", "postB"]
    span = (t[:, :, 5].max(1) - t[:, :, 0].min(1)) / 100.0
    print("phase           mean-over-waves   max-over-waves (mean over steps)")
    for i, nm in enumerate(names):
        print("%-14s %10.1f us %14.1f us" % (nm, d[:, :, i].mean(), d[:, :, i].max(1).mean()))
    print("first start -> last end: %.1f us" % span.mean())
    # inside postB (stamps 6 / 7 of go2_post.h): rewards + termination | reset_idx + push | observations + write-back
    r, x, o = (t[:, :, 6] - t[:, :, 4]) / 100.0, (t[:, :, 7] - t[:, :, 6]) / 100.0, (t[:, :, 5] - t[:, :, 7]) / 100.0
    hot = x > 1.0                            # waves that took the reset / push branch
    print("postB: rewards %.2f us | reset+push %.2f us mean, %.2f us over the %.1f%% of waves that took it (%.2f otherwise) | obs+store %.2f us"
          % (r.mean(), x.mean(), x[hot].mean() if hot.any() else 0.0, 100.0 * hot.mean(), x[~hot].mean(), o.mean()))
    tot = (t[:, :, 5] - t[:, :, 0]) / 100.0
    print("wave total: mean %.1f us, p99 %.1f us, max %.1f us (mean over steps of the per-step max); waves with reset/push: mean %.1f us"
          % (tot.mean(), np.quantile(tot, 0.99), tot.max(1).mean(), tot[hot].mean() if hot.any() else 0.0))
    full = hot & (t[:, :, 8] > 0) & (t[:, :, 13] > 0)        # waves whose FIRST env reset (the stamps inside the reset branch are that env's)
    if full.any():
        seq = [6, 8, 9, 10, 11, 12, 13, 7]
        lab = ["philox fill x2", "DOF tables", "terrain curriculum", "dofs + root state", "resample", "episode atomics", "push (fill + draw)"]
        print("reset path (%d samples): " % full.sum() + ", ".join("%s %.2f" % (lab[j], ((t[:, :, seq[j + 1]] - t[:, :, seq[j]]) / 100.0)[full].mean()) for j in range(7)))
    ss = np.diff(t[:, :, 16:23], axis=2) / 100.0
    print("substep 1 (mean / p99 over waves): " + ", ".join("%s %.2f / %.2f" % (nm, ss[:, :, j].mean(), np.quantile(ss[:, :, j], 0.99))
          for j, nm in enumerate(["pd+phaseA", "leg sums", "phaseB", "phaseC (contacts, rows)", "contact / limit solve", "gather+phaseD"])))
    sub_ = d[:, :, 1]
    print("substeps over waves: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.quantile(sub_, q) for q in (0.5, 0.9, 0.99, 1.0)))
    s.close()


if __name__ == "__main__":
    phase_clocks()
