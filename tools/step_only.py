#!/usr/bin/env python3
"""Run N fused env steps (nothing else) — the target of the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) that give
roofline.traffic for go2_step_kernel.   python tools/step_only.py [num_envs] [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import DeviceSim, load_hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
hip = load_hip()
s = DeviceSim(hip, num_envs=N)
s.reset_all()
a = torch.randn(N, 12, device="cuda:0") * 0.5
for _ in range(steps):
    hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
torch.cuda.synchronize()
print("done", N, steps)
s.close()
