#!/usr/bin/env python3
"""Run N fused env steps (nothing else) — the target of the rocprofv3 passes (FETCH_SIZE / WRITE_SIZE, SQ counters) that give
roofline.traffic for go2_step_kernel — and, with --probe, the calibration kernel of known byte count in the same access pattern.
   python tools/step_only.py [num_envs] [steps] [--task go2_flat|go2] [--probe]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import DeviceSim, heightfield_overrides, load_hip

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(argv[0]) if len(argv) > 0 else 4096
steps = int(argv[1]) if len(argv) > 1 else 100
task = sys.argv[sys.argv.index("--task") + 1] if "--task" in sys.argv else "go2_flat"
if "--task" in sys.argv:
    argv = [a for a in argv if a != task]
hip = load_hip()
if "--probe" in sys.argv:
    # known traffic: 200 fields read (= 800 B per env) and 540 fields written (= 2160 B per env): the step kernel's algorithmic mix
    NR, NW = 200, 540
    src, dst = torch.randn(NR, N, device="cuda:0"), torch.zeros(NW, N, device="cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(steps):
        hip.go2sim_debug_traffic_probe(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), N, NR, NW, st)
    torch.cuda.synchronize()
    print("probe", N, steps, NR * N * 4, NW * N * 4)
    sys.exit(0)
ov = heightfield_overrides(N)[1] if task == "go2" else {}
s = DeviceSim(hip, num_envs=N, **ov)
s.reset_all()
a = torch.randn(N, 12, device="cuda:0") * 0.5
for _ in range(steps):
    hip.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
torch.cuda.synchronize()
print("done", N, steps, task)
s.close()
