#!/usr/bin/env python3
"""Time go2_step_kernel for several pre-built variants of the library (experiments):  python tools/kvariants.py lib1.so lib2.so ...
Each library is loaded in a fresh subprocess (one HIP code object per process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
from go2_rl_gym_amd import _abi
from helpers import DeviceSim
lib = _abi.bind(sys.argv[1], C.c_float)
N = int(sys.argv[2])
s = DeviceSim(lib, num_envs=N); s.reset_all()
a = torch.randn(N, 12, device="cuda:0") * 0.5
for _ in range(80): lib.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
torch.cuda.synchronize(); lib.go2sim_enable_timing(s.h, 1)
for _ in range(300): lib.go2sim_step(s.h, C.c_void_p(a.data_ptr()), s._st())
ms, n = C.c_double(), C.c_int64(); lib.go2sim_kernel_time(s.h, C.byref(ms), C.byref(n))
print("%%-60s N=%%d kernel %%.1f us" %% (os.path.basename(sys.argv[1]), N, 1e3 * ms.value / n.value))
''' % (ROOT, ROOT)
N = os.environ.get("KV_N", "4096")
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), N])
