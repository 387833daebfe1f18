#!/usr/bin/env python3
"""Does the row pitch of the critic / actor input matter to the fp32 GEMMs?  263 / 45 columns (rows not 16-byte aligned) against the same
problem padded with zero columns to 264 / 48.   python tools/gemm_align.py"""
import torch

B = 24576


def t(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, Kp in ((263, 264), (45, 48)):
    for k in (K, Kp):
        x, w, b = torch.randn(B, k, device="cuda"), torch.randn(512, k, device="cuda"), torch.randn(512, device="cuda")
        gz = torch.randn(B, 512, device="cuda")
        print("K=%3d  forward addmm %6.1f us   weight-grad gz^T x %6.1f us   input-grad gz W %6.1f us" %
              (k, t(lambda: torch.addmm(b, x, w.t())), t(lambda: gz.t().mm(x)), t(lambda: gz.mm(w))))
    # a strided view of a padded buffer: same logical K, aligned pitch
    xp = torch.randn(B, Kp, device="cuda")[:, :K]
    w, b = torch.randn(512, K, device="cuda"), torch.randn(512, device="cuda")
    print("K=%3d in a pitch-%d buffer: forward addmm %6.1f us   weight-grad %6.1f us" % (K, Kp, t(lambda: torch.addmm(b, xp, w.t())), t(lambda: gz.t().mm(xp))))
