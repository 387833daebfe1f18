#!/usr/bin/env python3
"""One learner GEMM (include/go2nn.h go2nn_linear_*) launched `iters` times — the target of rocprofv3 counter passes (tools/gemm_pmc.sh).
usage: gemm_one.py f|i|w K N [M] [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from go2_rl_gym_amd import _nn

what, K, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 24576
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
nn = _nn.load_nn()
dev = "cuda:0"
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
y, gz = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
yp, gzp, gbp, dw = torch.nn.functional.elu(torch.randn(M, K, device=dev)), torch.empty(M, K, device=dev), torch.empty(K, device=dev), torch.empty(N, K, device=dev)
ws = torch.empty(int(nn.go2nn_linear_backward_workspace(M, N, K)), device=dev)
for _ in range(iters):
    if what == "f":
        nn.go2nn_linear_elu_forward(p(x), p(w), p(b), p(y), M, K, N, st())
    elif what == "i":
        nn.go2nn_linear_backward_input(p(gz), p(w), p(yp), p(gzp), p(gbp), p(ws), M, N, K, st())
    else:
        nn.go2nn_linear_backward_weight(p(gz), p(x), p(dw), p(ws), M, N, K, st())
torch.cuda.synchronize()
