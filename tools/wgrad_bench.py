#!/usr/bin/env python3
"""Weight-gradient GEMMs of the PPO mini-batch (dW = gz^T x: tiny output, reduction over 24576 rows): one mm (hipBLASLt picks a 3-way
split-K => ~60 workgroups on 256 CUs) against an explicit S-way split over the rows as a batched GEMM + a sum over S.
   python tools/wgrad_bench.py"""
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


for M, N in ((512, 263), (512, 45), (256, 512), (128, 256), (12, 128), (1, 128)):
    gz, x = torch.randn(B, M, device="cuda"), torch.randn(B, N, device="cuda")
    ref = gz.t().mm(x)
    line = "dW[%3d x %3d]  mm %6.1f us (%5.1f TF/s)" % (M, N, t(lambda: gz.t().mm(x)), 2.0 * B * M * N / t(lambda: gz.t().mm(x)) / 1e6)
    for S in (4, 8, 16, 32, 64):
        f = lambda: torch.bmm(gz.view(S, B // S, M).transpose(1, 2), x.view(S, B // S, N)).sum(0)
        err = (f() - ref).abs().max().item() / ref.abs().max().item()
        line += "   S=%d %6.1f us" % (S, t(f))
    print(line, "  rel err %.1e" % err)
