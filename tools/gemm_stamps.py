#!/usr/bin/env python3
"""Where a learner GEMM's workgroups spend their time: shader-clock stamps written by the -DGM_STAMPS build of libgo2nn (build/variants/libgo2nn_stamps.so).
usage: gemm_stamps.py f|i|w K N [M]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from go2_rl_gym_amd import _nn

what, K, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 24576
torch.cuda.is_available()
nn = _nn.bind(os.path.join(ROOT, "build", "variants", "libgo2nn_stamps.so"))
dev = "cuda:0"
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
y, gz = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
yp, gzp, gbp, dw = torch.nn.functional.elu(torch.randn(M, K, device=dev)), torch.empty(M, K, device=dev), torch.empty(K, device=dev), torch.empty(N, K, device=dev)
ws = torch.empty(int(nn.go2nn_linear_backward_workspace(M, N, K)), device=dev)
stamps = torch.zeros(8192, 8, dtype=torch.int64, device=dev)
nn.go2nn_debug_gemm_stamps.argtypes = [C.c_void_p]
nn.go2nn_debug_gemm_stamps(p(stamps))
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(4):
    stamps.zero_()
    torch.cuda.synchronize()
    a.record()
    if what == "f":
        nn.go2nn_linear_elu_forward(p(x), p(w), p(b), p(y), M, K, N, st())
    elif what == "i":
        nn.go2nn_linear_backward_input(p(gz), p(w), p(yp), p(gzp), p(gbp), p(ws), M, N, K, st())
    else:
        nn.go2nn_linear_backward_weight(p(gz), p(x), p(dw), p(ws), M, N, K, st())
    e.record()
    torch.cuda.synchronize()
s = stamps.cpu().numpy()
s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
span = s[:, 3].max() - t0
us = a.elapsed_time(e) * 1e3
print("%s K=%d N=%d M=%d: %d workgroups, launch %.1f us by HIP events, first start -> last end %d ticks (%.3f us per kilo-tick if that is the kernel)" % (what, K, N, M, len(s), us, span, us / span * 1e3))
q = lambda v: "mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max())
print("start after first start   ", q(s[:, 0] - t0))
print("prologue (first tile in)  ", q(s[:, 1] - s[:, 0]))
print("k-loop                    ", q(s[:, 2] - s[:, 1]))
print("  load issue              ", q(s[:, 4]))
print("  fragment reads + MFMAs  ", q(s[:, 5]))
print("  wait for loads + LDS st ", q(s[:, 6]))
print("  barrier                 ", q(s[:, 7]))
print("epilogue                  ", q(s[:, 3] - s[:, 2]))
print("workgroup total           ", q(s[:, 3] - s[:, 0]))
print("end after first start     ", q(s[:, 3] - t0))
