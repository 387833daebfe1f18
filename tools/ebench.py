#!/usr/bin/env python3
"""Micro-benchmark of go2sim_elu_backward_bias against the two torch kernels it replaces (elu_backward + column sum), at the PPO
mini-batch shapes.   python tools/ebench.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import load_hip

hip = load_hip()
B = 24576
for Cn in (512, 256, 128):
    gy, y = torch.randn(B, Cn, device="cuda"), torch.randn(B, Cn, device="cuda")
    gz, gb = torch.empty_like(y), torch.empty(Cn, device="cuda")
    ws = torch.empty(Cn * ((B + 63) // 64), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fused():
        assert hip.go2sim_elu_backward_bias(p(gy), p(y), p(gz), p(gb), p(ws), B, Cn, st) == 0

    def plain():
        g = torch.ops.aten.elu_backward(gy, 1.0, 1.0, 1.0, True, y)
        return g, g.sum(0)

    for name, fn in (("torch elu_backward + sum", plain), ("go2sim_elu_backward_bias", fused)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print("[%5d x %3d] %-28s %7.1f us   %6.0f GB/s (12 B/element)" % (B, Cn, name, us, 12.0 * B * Cn / us / 1e3))
    g, s = plain(); fused()
    print("   max |gz - ref| %.2e   max |gb - ref| %.2e (rel %.1e)" % ((gz - g).abs().max().item(), (gb - s).abs().max().item(), ((gb - s).abs().max() / s.abs().max()).item()))
