#!/usr/bin/env python3
"""go2_step_kernel alone over batch sizes: HIP events inside the library, 200 timed steps after 60 settling steps under N(0, 0.5) actions.
   python tools/kscale.py [rough] [sizes...]        (profiles/r3_kernel_scaling.txt)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import DeviceSim, load_hip

rough = "rough" in sys.argv[1:]
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 4096, 8192, 16384, 32768]
hip = load_hip()
print("N envs   kernel us   env-steps/s kernel-only   (%s)" % ("task go2 trimesh terrain" if rough else "plane"))
for N in sizes:
    ov = {}
    if rough:
        from helpers import heightfield_overrides
        ov = heightfield_overrides(N, mesh_type="trimesh")[1]
    for _ in (0,):
        s = DeviceSim(hip, num_envs=N, **ov)
        s.reset_all()
        a = torch.randn(N, 12, device="cuda:0") * 0.5
        for _ in range(60):
            hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
        torch.cuda.synchronize()
        hip.go2sim_enable_timing(s.h, 1)
        for _ in range(200):
            hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
        ms, n = C.c_double(), C.c_int64()
        hip.go2sim_kernel_time(s.h, C.byref(ms), C.byref(n))
        k = ms.value / n.value
        print("%6d   %9.1f   %8.2f M" % (N, k * 1e3, N / k / 1e3), flush=True)
        s.close()
