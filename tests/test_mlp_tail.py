"""Learner-side kernels of include/go2nn.h (go2_rl_gym_amd/csrc/go2nn_train.h) — here the host build of the same entry points and the autograd
node of modules/fused.py that uses them, against plain PyTorch autograd.  GPU twin (the HIP kernels): tests/test_gpu_mlp_tail.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import load_nn_emu, load_oracle

SHAPES = [(1, 1, 4), (37, 12, 128), (300, 1, 128), (257, 16, 512), (100, 5, 36), (513, 3, 260)]


def head_backward(lib, gy, y, w):
    """-> gz, dW, gb, db through go2nn_head_backward"""
    B, K = y.shape
    Cn = w.shape[0]
    n = lib.go2nn_head_backward_workspace(B, Cn, K)
    assert n > 0
    gz, sums, ws = torch.empty_like(y), torch.full(((Cn + 1) * K + Cn,), float("nan"), device=y.device), torch.full((int(n),), float("nan"), device=y.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if y.is_cuda else None
    rc = lib.go2nn_head_backward(p(gy), p(y), p(w), p(gz), p(sums), p(ws), B, Cn, K, stream)
    assert rc == 0, lib.go2nn_last_error().decode()
    return gz, sums[:Cn * K].view(Cn, K), sums[Cn * K:(Cn + 1) * K], sums[(Cn + 1) * K:]


def head_backward_reference(gy, y, w):
    """the same quantities in float64 with torch ops (mm, elu_backward in result form, column sums)"""
    gy, y, w = gy.double(), y.double(), w.double()
    gz = gy.mm(w) * torch.where(y > 0, torch.ones_like(y), y + 1.0)
    return gz, gy.t().mm(y), gz.sum(0), gy.sum(0)


def check_head_backward(lib, B, Cn, K, device="cpu"):
    g = torch.Generator().manual_seed(B * 1000 + Cn * 10 + K)
    gy, w = torch.randn(B, Cn, generator=g).to(device), (torch.randn(Cn, K, generator=g) * 0.3).to(device)
    y = torch.nn.functional.elu(torch.randn(B, K, generator=g)).to(device)          # an ELU output: both branches of the derivative occur
    got, ref = head_backward(lib, gy, y, w), head_backward_reference(gy, y, w)
    scale = float(np.sqrt(B))                                                      # fp32 sums of B terms of unit scale
    for a, b, tol in zip(got, ref, (2e-6 * Cn, 2e-6 * scale, 2e-6 * scale * Cn, 2e-6 * scale)):
        np.testing.assert_allclose(a.cpu().double().numpy(), b.cpu().numpy(), atol=tol * 4, rtol=2e-5)
    return got


@pytest.mark.parametrize("B,Cn,K", SHAPES)
def test_head_backward_matches_torch(B, Cn, K):
    check_head_backward(load_nn_emu(), B, Cn, K)


def test_head_backward_refuses_unsupported_shapes():
    lib = load_nn_emu()
    assert lib.go2nn_head_backward_workspace(10, 17, 128) < 0 and b"head backward" in lib.go2nn_last_error()
    assert lib.go2nn_head_backward_workspace(10, 4, 130) < 0 and lib.go2nn_head_backward_workspace(10, 4, 516) < 0 and lib.go2nn_head_backward_workspace(0, 4, 128) < 0
    t = torch.zeros(4)
    assert lib.go2nn_head_backward(None, C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), 1, 1, 4, None) < 0


def tail_vs_autograd(nn_lib, sim_lib, device, B=200, dims=(45, 64, 32, 12), atol=2e-6, node=True, own="auto"):
    """FusedSequential against the same parameters under plain autograd: outputs and every gradient.  node: the whole-MLP autograd node (_FusedMLP)
    or the per-layer nodes (_LinearELU + _LinearELUHead); own: which products go to the go2nn GEMMs (auto / all / none)."""
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.actor_critic import _mlp
    torch.manual_seed(3)
    net = _mlp(dims[0], list(dims[1:-1]), dims[-1], "elu").to(device)
    x, tgt = torch.randn(B, dims[0], device=device, requires_grad=True), torch.randn(B, dims[-1], device=device)
    res, saved = [], (fused._MLP_NODE, dict(fused._OWN))
    fused._MLP_NODE = node
    fused._OWN.update(f=own, i=own, w=own)
    try:
        for on in (False, True):
            fused.set_library(sim_lib if on else None)
            fused.set_nn_library(nn_lib if on else None)
            try:
                net.zero_grad(); x.grad = None
                out = net(x)
                assert (type(out.grad_fn).__name__ == ("_FusedMLPBackward" if node else "_LinearELUHeadBackward")) == on
                ((out - tgt) ** 2).mean().backward()
                res.append((out.detach().clone(), [p.grad.clone() for p in net.parameters()] + [x.grad.clone()]))
            finally:
                fused.set_library(None); fused.set_nn_library(None)
    finally:
        fused._MLP_NODE = saved[0]; fused._OWN.update(saved[1])
    np.testing.assert_allclose(res[1][0].cpu().numpy(), res[0][0].cpu().numpy(), atol=1e-5)
    for a, b in zip(res[0][1], res[1][1]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), atol=atol, rtol=2e-3)


@pytest.mark.parametrize("dims", [(45, 64, 32, 12), (263, 48, 1), (20, 16, 16, 16, 3)])
@pytest.mark.parametrize("node,own", [(False, "auto"), (True, "auto"), (True, "all"), (True, "none")])
def test_fused_tail_matches_autograd(dims, node, own):
    tail_vs_autograd(load_nn_emu(), load_oracle(), "cpu", dims=dims, node=node, own=own)


def test_tail_is_not_taken_without_the_library_or_for_wide_outputs():
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.actor_critic import _mlp
    net, wide = _mlp(8, [16], 4, "elu"), _mlp(8, [16], 32, "elu")
    x = torch.randn(5, 8)
    fused.set_library(load_oracle())
    try:
        assert type(net(x).grad_fn).__name__ != "_LinearELUHeadBackward"          # go2sim library alone: the Linear -> ELU pair path as before
        fused.set_nn_library(load_nn_emu())
        assert type(net(x).grad_fn).__name__ == "_FusedMLPBackward"
        assert type(wide(x).grad_fn).__name__ not in ("_LinearELUHeadBackward", "_FusedMLPBackward")
        with torch.no_grad():
            assert net(x).grad_fn is None
    finally:
        fused.set_library(None); fused.set_nn_library(None)


# ---- the hidden layers: go2nn_linear_* (go2_rl_gym_amd/csrc/go2nn_gemm.h on the GPU; plain loops in the host build) --------------------------
LINEAR_SHAPES = [(1, 1, 1), (65, 8, 33), (70, 37, 70), (130, 45, 64), (96, 64, 40)]      # (M, K, N)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream) if t.is_cuda else None


def linear_elu_forward(lib, x, w, b):
    y = torch.full((x.shape[0], w.shape[0]), float("nan"), device=x.device)
    rc = lib.go2nn_linear_elu_forward(_p(x), _p(w), _p(b), _p(y), x.shape[0], x.shape[1], w.shape[0], _stream(x))
    assert rc == 0, lib.go2nn_last_error().decode()
    return y


def linear_backward(lib, gz, w, x, y_prev):
    """-> gz_prev, gb_prev, dw of a layer with weight w [C, Kin], input x = y_prev [M, Kin], pre-activation gradient gz [M, C]"""
    M, Cn = gz.shape
    Kin = w.shape[1]
    n = lib.go2nn_linear_backward_workspace(M, Cn, Kin)
    assert n > 0
    ws = torch.full((int(n),), float("nan"), device=gz.device)
    gzp, gbp, dw = torch.full((M, Kin), float("nan"), device=gz.device), torch.full((Kin,), float("nan"), device=gz.device), torch.full((Cn, Kin), float("nan"), device=gz.device)
    assert lib.go2nn_linear_backward_input(_p(gz), _p(w), _p(y_prev), _p(gzp), _p(gbp), _p(ws), M, Cn, Kin, _stream(gz)) == 0, lib.go2nn_last_error().decode()
    assert lib.go2nn_linear_backward_weight(_p(gz), _p(x), _p(dw), _p(ws), M, Cn, Kin, _stream(gz)) == 0, lib.go2nn_last_error().decode()
    return gzp, gbp, dw


def check_linear(lib, M, K, N, device="cpu"):
    """forward and both backward products of one layer against float64 torch; the tolerance is fp32 round-off of sums of K (resp. M) products"""
    g = torch.Generator().manual_seed(M * 7 + K * 3 + N)
    x, w, b = torch.randn(M, K, generator=g).to(device), (torch.randn(N, K, generator=g) / np.sqrt(K)).to(device), torch.randn(N, generator=g).to(device)
    y = linear_elu_forward(lib, x, w, b)
    ref = torch.nn.functional.elu(x.double().mm(w.double().t()) + b.double())
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.cpu().numpy(), atol=3e-6 * np.sqrt(K) + 2e-6, rtol=2e-6)
    # this layer's backward: gz [M, N] against its input x — with x standing for an ELU output (y_prev) of the layer before
    gz, yp = torch.randn(M, N, generator=g).to(device), torch.nn.functional.elu(torch.randn(M, K, generator=g)).to(device)
    gzp, gbp, dw = linear_backward(lib, gz, w, yp, yp)
    r_gzp = gz.double().mm(w.double()) * torch.where(yp > 0, torch.ones_like(yp), yp + 1.0).double()
    np.testing.assert_allclose(gzp.cpu().double().numpy(), r_gzp.cpu().numpy(), atol=3e-6 * np.sqrt(N) + 2e-6, rtol=2e-6)
    np.testing.assert_allclose(gbp.cpu().double().numpy(), r_gzp.sum(0).cpu().numpy(), atol=4e-6 * np.sqrt(M * N) + 1e-5, rtol=2e-5)
    np.testing.assert_allclose(dw.cpu().double().numpy(), gz.double().t().mm(yp.double()).cpu().numpy(), atol=4e-6 * np.sqrt(M) + 1e-5, rtol=2e-5)
    return y, gzp, gbp, dw


@pytest.mark.parametrize("M,K,N", LINEAR_SHAPES)
def test_linear_layer_matches_torch(M, K, N):
    check_linear(load_nn_emu(), M, K, N)


def check_sum_rows(lib, device="cpu"):
    """go2nn_sum_rows: several jobs of both block shapes (<= 32 rows: one thread per column; more: 16 x 16 tree) in one launch, against float64 sums"""
    from go2_rl_gym_amd._nn import Go2nnSumJob
    g = torch.Generator().manual_seed(5)
    shapes = [(8, 131072), (256, 1676), (384, 512), (1, 7), (33, 17), (32, 300), (192, 128)]
    parts = [torch.randn(r, c, generator=g).to(device) for r, c in shapes]
    outs = [torch.full((c,), float("nan"), device=device) for _, c in shapes]
    arr = (Go2nnSumJob * len(shapes))(*[Go2nnSumJob(p.data_ptr(), o.data_ptr(), r, c) for p, o, (r, c) in zip(parts, outs, shapes)])
    assert lib.go2nn_sum_rows(arr, len(shapes), _stream(parts[0])) == 0, lib.go2nn_last_error().decode()
    for p, o, (r, c) in zip(parts, outs, shapes):
        np.testing.assert_allclose(o.cpu().double().numpy(), p.double().sum(0).cpu().numpy(), atol=3e-6 * np.sqrt(r) * 4, rtol=1e-6)
    assert lib.go2nn_sum_rows(arr, 17, None) < 0 and lib.go2nn_sum_rows(None, 1, None) < 0
    return outs


def test_sum_rows_matches_torch():
    check_sum_rows(load_nn_emu())


def test_whole_mlp_node_edge_cases():
    """one hidden layer, an input that needs no gradient, a non-contiguous input, a second backward through the same parameters (gradient
    accumulation): the whole-MLP node behaves like plain autograd"""
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.actor_critic import _mlp
    torch.manual_seed(7)
    net = _mlp(10, [8], 2, "elu")
    xw = torch.randn(33, 20)
    x = xw[:, ::2]                                     # non-contiguous view
    assert not x.is_contiguous()
    ref = torch.nn.Sequential(*net.children())        # the same modules under plain autograd
    ref(x).pow(2).sum().backward()
    ref(2 * x).pow(2).sum().backward()                 # accumulates
    want = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    fused.set_library(load_oracle()); fused.set_nn_library(load_nn_emu())
    try:
        out = net(x)
        assert type(out.grad_fn).__name__ == "_FusedMLPBackward"
        out.pow(2).sum().backward()
        net(2 * x).pow(2).sum().backward()
    finally:
        fused.set_library(None); fused.set_nn_library(None)
    for a, b in zip(want, [p.grad for p in net.parameters()]):
        np.testing.assert_allclose(b.numpy(), a.numpy(), atol=2e-5, rtol=1e-4)
