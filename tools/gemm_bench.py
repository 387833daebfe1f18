#!/usr/bin/env python3
"""A/B of the learner's hidden-layer kernels (include/go2nn.h go2nn_linear_*, csrc/go2nn_gemm.h) against what they replace, at the update's shapes
(M = 24576 rows per mini-batch; actor 45-512-256-128, critic 263-512-256-128), one MI355X:
  forward        go2nn_linear_elu_forward             vs  torch.addmm (hipBLASLt) + elu_
  input grad     go2nn_linear_backward_input          vs  mm + go2sim_elu_backward_bias (two launches)
  weight grad    go2nn_linear_backward_weight         vs  the row-split bmm + sum(0) of modules/fused.py
each alone on the chip and as the actor/critic PAIR on two streams (how the update runs them).  Times: HIP events around 30 launches."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from go2_rl_gym_amd import _lib, _nn
from go2_rl_gym_amd.rsl_rl.modules import fused

M = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
nn, sim = _nn.load_nn(), _lib.load_hip()
dev = "cuda:0"
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, n=30, streams=None):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def pair(f1, f2):
    s2 = torch.cuda.Stream()
    def run():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            f2()
        f1()
        cur.wait_stream(s2)
    return run


def layer(K, N):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
    y, gz = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
    yp, gzp, gbp, dw = F.elu(torch.randn(M, K, device=dev)), torch.empty(M, K, device=dev), torch.empty(K, device=dev), torch.empty(N, K, device=dev)
    ws = torch.empty(int(nn.go2nn_linear_backward_workspace(M, N, K)), device=dev)
    ws2 = torch.empty(K * ((M + 63) // 64), device=dev)
    own_f = lambda: nn.go2nn_linear_elu_forward(p(x), p(w), p(b), p(y), M, K, N, st())
    ref_f = lambda: F.elu_(torch.addmm(b, x, w.t()))
    own_i = lambda: nn.go2nn_linear_backward_input(p(gz), p(w), p(yp), p(gzp), p(gbp), p(ws), M, N, K, st())
    def ref_i():
        g = gz.mm(w)
        sim.go2sim_elu_backward_bias(p(g), p(yp), p(gzp), p(gbp), p(ws2), M, K, st())
    own_w = lambda: nn.go2nn_linear_backward_weight(p(gz), p(x), p(dw), p(ws), M, N, K, st())
    ref_w = lambda: torch.bmm(gz.reshape(8, gz.shape[0] // 8, -1).transpose(1, 2), x.reshape(8, x.shape[0] // 8, -1)).sum(0)          # (the row-split vendor formulation of rounds 1-3)
    return dict(f=(own_f, ref_f), i=(own_i, ref_i), w=(own_w, ref_w)), 2.0 * M * K * N


LAYERS = {"A1": (45, 512), "C1": (263, 512), "L2": (512, 256), "L3": (256, 128)}
print("M = %d; us per launch (TF/s)           own      torch+hipBLASLt" % M)
fns = {}
for name, (K, N) in LAYERS.items():
    fns[name], fl = layer(K, N)
    for what, label in (("f", "forward + ELU"), ("i", "input grad + ELU' + bias grad"), ("w", "weight grad")):
        if what == "i" and name in ("A1", "C1"):
            continue        # the first layer has no input gradient
        o, r = timed(fns[name][what][0]), timed(fns[name][what][1])
        print("%s %-3dx%-3d %-30s %7.1f (%5.1f)   %7.1f (%5.1f)" % (name, K, N, label, o, fl / o / 1e6, r, fl / r / 1e6))
print("pairs on two streams (actor | critic):")
for what, label, a, b in (("f", "layer 1 forward", "A1", "C1"), ("f", "layer 2 forward", "L2", "L2"), ("f", "layer 3 forward", "L3", "L3"),
                          ("i", "layer 2 input grad", "L2", "L2"), ("i", "layer 3 input grad", "L3", "L3"),
                          ("w", "layer 1 weight grad", "A1", "C1"), ("w", "layer 2 weight grad", "L2", "L2"), ("w", "layer 3 weight grad", "L3", "L3")):
    o = timed(pair(fns[a][what][0], fns[b][what][0]))
    r = timed(pair(fns[a][what][1], fns[b][what][1]))
    mx = timed(pair(fns[a][what][0], fns[b][what][1]))          # the actor's on the own kernel, the critic's on the vendor path
    print("%-24s own %7.1f   torch %7.1f   own | torch %7.1f" % (label, o, r, mx))
