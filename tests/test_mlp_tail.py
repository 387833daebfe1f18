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


def tail_vs_autograd(nn_lib, sim_lib, device, B=200, dims=(45, 64, 32, 12), atol=2e-6, split=True):
    """FusedSequential (the whole-MLP autograd node of modules/fused.py: _FusedMLP) against the same parameters under plain autograd: output and every gradient.
    split: the split-operand kernels (default) or the fp32-MFMA ones (GO2_GEMM_SPLIT=0) — on the host build the same loops either way"""
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.actor_critic import _mlp
    torch.manual_seed(3)
    net = _mlp(dims[0], list(dims[1:-1]), dims[-1], "elu").to(device)
    x, tgt = torch.randn(B, dims[0], device=device, requires_grad=True), torch.randn(B, dims[-1], device=device)
    res, saved = [], fused._SPLIT
    fused._SPLIT = split
    try:
        for on in (False, True):
            fused.set_library(sim_lib if on else None)
            fused.set_nn_library(nn_lib if on else None)
            try:
                net.zero_grad(); x.grad = None
                out = net(x)
                assert (type(out.grad_fn).__name__ == "_FusedMLPBackward") == on
                ((out - tgt) ** 2).mean().backward()
                res.append((out.detach().clone(), [p.grad.clone() for p in net.parameters()] + [x.grad.clone()]))
            finally:
                fused.set_library(None); fused.set_nn_library(None)
    finally:
        fused._SPLIT = saved
    np.testing.assert_allclose(res[1][0].cpu().numpy(), res[0][0].cpu().numpy(), atol=1e-5)
    for a, b in zip(res[0][1], res[1][1]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), atol=atol, rtol=2e-3)


@pytest.mark.parametrize("dims", [(45, 64, 32, 12), (263, 48, 1), (20, 16, 16, 16, 3), (60, 32, 16, 8), (225, 40, 24, 32), (12, 8, 40)])
@pytest.mark.parametrize("split", [True, False])
def test_fused_tail_matches_autograd(dims, split):
    """narrow heads (12 / 1 / 3 / 8 wide: go2nn_head_backward) and wide ones (an encoder's 32-wide latent, a 40-wide layer: the last Linear on the GEMM kernels)"""
    if not split and dims[0] % 4 and False:
        pytest.skip()
    tail_vs_autograd(load_nn_emu(), load_oracle(), "cpu", dims=dims, split=split)


def test_node_is_taken_only_with_both_libraries_and_runs_the_rest_as_modules():
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.actor_critic import _mlp
    from go2_rl_gym_amd.rsl_rl.modules.utils import L2Norm
    net, enc = _mlp(8, [16], 4, "elu"), _mlp(8, [16], 8, "elu")
    enc.append(L2Norm())
    odd = torch.nn.Sequential(torch.nn.Linear(8, 6), torch.nn.ELU(), torch.nn.Linear(6, 2))          # a hidden width that is not a multiple of 4
    x = torch.randn(5, 8)
    fused.set_library(load_oracle())
    try:
        assert type(net(x).grad_fn).__name__ != "_FusedMLPBackward"          # the go2sim library alone: plain modules
        fused.set_nn_library(load_nn_emu())
        assert type(net(x).grad_fn).__name__ == "_FusedMLPBackward"
        y = enc(x)                                                             # the stack as a node, the normaliser behind it as the module it is
        assert "FusedMLP" not in type(y.grad_fn).__name__ and torch.allclose(y.norm(dim=1), torch.ones(5), atol=1e-6)
        ref = torch.nn.functional.normalize(torch.nn.Sequential(*list(enc)[:-1])(x), dim=-1)
        np.testing.assert_allclose(y.detach().numpy(), ref.detach().numpy(), atol=2e-6)
        assert type(fused.FusedSequential(*odd)(x).grad_fn).__name__ != "_FusedMLPBackward"
        with torch.no_grad():
            assert net(x).grad_fn is None
            with fused.own_forward():          # the same kernels without a node (the update's once-per-update latents)
                np.testing.assert_allclose(net(x).numpy(), torch.nn.Sequential(*net)(x).numpy(), atol=2e-6)
                np.testing.assert_allclose(enc(x).numpy(), ref.detach().numpy(), atol=2e-6)
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
    # ABI 4: a job may ADD its first nacc column sums to a running vector (PPO.update's loss sums over the mini-batches): two launches add twice
    accs = [torch.full((5,), 10.0, device=device), torch.full((3,), -1.0, device=device)]
    arr2 = (Go2nnSumJob * 2)(Go2nnSumJob(parts[1].data_ptr(), outs[1].data_ptr(), shapes[1][0], shapes[1][1], accs[0].data_ptr(), 5, 0),
                             Go2nnSumJob(parts[5].data_ptr(), outs[5].data_ptr(), shapes[5][0], shapes[5][1], accs[1].data_ptr(), 2, 0))
    for _ in range(2):
        assert lib.go2nn_sum_rows(arr2, 2, _stream(parts[0])) == 0, lib.go2nn_last_error().decode()
    np.testing.assert_allclose(accs[0].cpu().double().numpy(), 10.0 + 2 * parts[1].double().sum(0)[:5].cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(accs[1].cpu().double().numpy()[:2], -1.0 + 2 * parts[5].double().sum(0)[:2].cpu().numpy(), atol=2e-4)
    assert float(accs[1][2]) == -1.0          # beyond nacc: untouched
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


# ---- ABI 3: the grouped layer calls (one launch for the actor's and the critic's layer; go2_rl_gym_amd/csrc/go2nn_gemm3.h on the GPU) -----------
GROUP_SHAPES = [(70, (45, 96), (263, 96)), (130, (37, 70), (64, 70)), (65, (130, 33), (8, 33)), (96, (64, 40), (64, 72))]      # M, (K, N) of job 0, of job 1


def check_linear_group(lib, M, s0, s1, device="cpu", split=False, with_refs=False):
    """forward, input gradient and weight gradient of two layers as ONE grouped call each, against float64 torch (same bounds as check_linear).
    split: the ABI 4 split-operand kernels (3 x bf16 planes per fp32 value, six MFMA terms) — held to the SAME bounds as the fp32-MFMA kernels."""
    from go2_rl_gym_amd._nn import Go2nnBwdInJob, Go2nnBwdWJob, Go2nnFwdJob, Go2nnSplitJob, Go2nnSumJob
    g = torch.Generator().manual_seed(M * 7 + s0[0] * 3 + s1[0])
    nan = lambda *shape: torch.full(shape, float("nan"), device=device)
    jobs = []
    for K, N in (s0, s1):
        x, w, b = torch.randn(M, K, generator=g).to(device), (torch.randn(N, K, generator=g) / np.sqrt(K)).to(device), torch.randn(N, generator=g).to(device)
        gz, yp = torch.randn(M, N, generator=g).to(device), torch.nn.functional.elu(torch.randn(M, K, generator=g)).to(device)
        jobs.append(dict(K=K, N=N, x=x, w=w, b=b, gz=gz, yp=yp, y=nan(M, N), gzp=nan(M, K), gbp=nan(K), dw=nan(N, K)))
    st = _stream(jobs[0]["x"])
    p = lambda t: t.data_ptr()
    for j in jobs:
        j["img"] = None
        if split:
            n = lib.go2nn_split_weights_bytes(j["N"], j["K"])
            assert n > 0, lib.go2nn_last_error().decode()
            j["img_t"] = torch.full((int(n),), 0xff, dtype=torch.uint8, device=device)
            j["img"] = j["img_t"].data_ptr()
    if split:
        sj = (Go2nnSplitJob * 2)(*[Go2nnSplitJob(p(j["w"]), j["img"], j["N"], j["K"]) for j in jobs])
        assert lib.go2nn_split_weights(sj, 2, st) == 0, lib.go2nn_last_error().decode()
    fj = (Go2nnFwdJob * 2)(*[Go2nnFwdJob(p(j["x"]), p(j["w"]), p(j["b"]), p(j["y"]), M, j["K"], j["N"], 0, j["img"]) for j in jobs])
    assert lib.go2nn_linear_elu_forward_group(fj, 2, st) == 0, lib.go2nn_last_error().decode()
    sums, keep = [], []
    ij = (Go2nnBwdInJob * 2)()
    for k, j in enumerate(jobs):
        r = lib.go2nn_linear_backward_input_group_rows(M, j["N"], j["K"])
        assert r > 0
        ws = nan(r * j["K"]); keep.append(ws)
        ij[k] = Go2nnBwdInJob(p(j["gz"]), p(j["w"]), p(j["yp"]), p(j["gzp"]), p(ws), M, j["N"], j["K"], 0, j["img"])
        sums.append((ws, j["gbp"], r, j["K"]))
    assert lib.go2nn_linear_backward_input_group(ij, 2, st) == 0, lib.go2nn_last_error().decode()
    wj = (Go2nnBwdWJob * 2)(*[Go2nnBwdWJob(p(j["gz"]), p(j["yp"]), None, M, j["N"], j["K"], 1 if split else 0) for j in jobs])
    rows = lib.go2nn_linear_backward_weight_group_rows(wj, 2)
    assert rows > 0, lib.go2nn_last_error().decode()
    for k, j in enumerate(jobs):
        ws = nan(rows * j["N"] * j["K"]); keep.append(ws)
        wj[k].workspace = p(ws)
        sums.append((ws, j["dw"], rows, j["N"] * j["K"]))
    assert lib.go2nn_linear_backward_weight_group(wj, 2, st) == 0, lib.go2nn_last_error().decode()
    arr = (Go2nnSumJob * len(sums))(*[Go2nnSumJob(t[0].data_ptr(), t[1].data_ptr(), t[2], t[3]) for t in sums])
    assert lib.go2nn_sum_rows(arr, len(sums), st) == 0
    out, refs = [], []
    for j in jobs:
        K, N, x, w, b, gz, yp = (j[k] for k in ("K", "N", "x", "w", "b", "gz", "yp"))
        ref = torch.nn.functional.elu(x.double().mm(w.double().t()) + b.double())
        np.testing.assert_allclose(j["y"].cpu().double().numpy(), ref.cpu().numpy(), atol=3e-6 * np.sqrt(K) + 2e-6, rtol=2e-6)
        r_gzp = gz.double().mm(w.double()) * torch.where(yp > 0, torch.ones_like(yp), yp + 1.0).double()
        np.testing.assert_allclose(j["gzp"].cpu().double().numpy(), r_gzp.cpu().numpy(), atol=3e-6 * np.sqrt(N) + 2e-6, rtol=2e-6)
        np.testing.assert_allclose(j["gbp"].cpu().double().numpy(), r_gzp.sum(0).cpu().numpy(), atol=4e-6 * np.sqrt(M * N) + 1e-5, rtol=2e-5)
        np.testing.assert_allclose(j["dw"].cpu().double().numpy(), gz.double().t().mm(yp.double()).cpu().numpy(), atol=4e-6 * np.sqrt(M) + 1e-5, rtol=2e-5)
        out += [j["y"], j["gzp"], j["gbp"], j["dw"]]
        refs += [ref, r_gzp, r_gzp.sum(0), gz.double().t().mm(yp.double())]
    return (out, refs) if with_refs else out


@pytest.mark.parametrize("M,s0,s1", GROUP_SHAPES)
def test_linear_group_matches_torch(M, s0, s1):
    check_linear_group(load_nn_emu(), M, s0, s1)
    check_linear_group(load_nn_emu(), M, s0, s1, split=True)          # (the host build reads the fp32 weights: the call path and the struct layout)


# ---- ABI 6: the weight gradient of the layer below out of the input gradient's epilogue (Go2nnBwdInJob.x_in; go2_rl_gym_amd/csrc/go2nn_bx3.h EPI_DELU_WG) -----------
BELOW_SHAPES = [(70, (96, 40), 45, 48, True), (130, (37, 70), 64, 7, False), (257, (8, 33), 33, 32, True), (128, (64, 128), 1, 45, False)]      # M, (C, Kin), Kx job 0, Kx job 1, also store gz_prev


def check_wgrad_below(lib, M, s, kx0, kx1, keep_gz, device="cpu"):
    """two jobs of one input-gradient launch that also leave dW_below = gz_prev^T x_in: against float64 torch, bounds of check_linear_group; gz_prev optional"""
    from go2_rl_gym_amd._nn import Go2nnBwdInJob, Go2nnSplitJob, Go2nnSumJob
    C_, Kin = s
    g = torch.Generator().manual_seed(M * 5 + C_ * 3 + Kin + kx0)
    nan = lambda *shape: torch.full(shape, float("nan"), device=device)
    p = lambda t: t.data_ptr()
    jobs = []
    for kx in (kx0, kx1):
        w = (torch.randn(C_, Kin, generator=g) / np.sqrt(C_)).to(device)
        gz, yp = torch.randn(M, C_, generator=g).to(device), torch.nn.functional.elu(torch.randn(M, Kin, generator=g)).to(device)
        xw = torch.randn(M, kx + (5 if len(jobs) else 0), generator=g).to(device)          # job 1: x_in is columns [3, 3 + kx) of a wider matrix (Go2nnBwdInJob.ldx)
        x = xw[:, 3:3 + kx] if len(jobs) else xw
        n = lib.go2nn_split_weights_bytes(C_, Kin)
        assert n > 0, lib.go2nn_last_error().decode()
        jobs.append(dict(kx=kx, w=w, gz=gz, yp=yp, x=x, img=torch.full((int(n),), 0xff, dtype=torch.uint8, device=device), gzp=nan(M, Kin) if keep_gz else None, gbp=nan(Kin), dw=nan(Kin, kx)))
    st = _stream(jobs[0]["x"])
    sj = (Go2nnSplitJob * 2)(*[Go2nnSplitJob(p(j["w"]), p(j["img"]), C_, Kin) for j in jobs])
    assert lib.go2nn_split_weights(sj, 2, st) == 0, lib.go2nn_last_error().decode()
    ij, sums, keep = (Go2nnBwdInJob * 2)(), [], []
    rows = lib.go2nn_linear_backward_input_fused_rows(M)
    r = lib.go2nn_linear_backward_input_group_rows(M, C_, Kin)
    assert rows > 0 and r > 0
    for k, j in enumerate(jobs):
        ws, dws = nan(r * Kin), nan(rows * Kin * j["kx"]); keep += [ws, dws]
        ij[k] = Go2nnBwdInJob(p(j["gz"]), p(j["w"]), p(j["yp"]), p(j["gzp"]) if keep_gz else None, p(ws), M, C_, Kin, 0, p(j["img"]), 0, j["kx"], p(j["x"]), p(dws), j["x"].stride(0) if k else 0)
        sums += [(ws, j["gbp"], r, Kin), (dws, j["dw"], rows, Kin * j["kx"])]
    assert lib.go2nn_linear_backward_input_group(ij, 2, st) == 0, lib.go2nn_last_error().decode()
    arr = (Go2nnSumJob * len(sums))(*[Go2nnSumJob(t[0].data_ptr(), t[1].data_ptr(), t[2], t[3]) for t in sums])
    assert lib.go2nn_sum_rows(arr, len(sums), st) == 0
    out = []
    for j in jobs:
        yp = j["yp"]
        r_gzp = j["gz"].double().mm(j["w"].double()) * torch.where(yp > 0, torch.ones_like(yp), yp + 1.0).double()
        if keep_gz:
            np.testing.assert_allclose(j["gzp"].cpu().double().numpy(), r_gzp.cpu().numpy(), atol=3e-6 * np.sqrt(C_) + 2e-6, rtol=2e-6)
            out.append(j["gzp"])
        np.testing.assert_allclose(j["gbp"].cpu().double().numpy(), r_gzp.sum(0).cpu().numpy(), atol=4e-6 * np.sqrt(M * C_) + 1e-5, rtol=2e-5)
        np.testing.assert_allclose(j["dw"].cpu().double().numpy(), r_gzp.t().mm(j["x"].double()).cpu().numpy(), atol=4e-6 * np.sqrt(M * C_) + 1e-5, rtol=2e-5)
        out += [j["gbp"], j["dw"]]
    return out


@pytest.mark.parametrize("M,s,kx0,kx1,keep_gz", BELOW_SHAPES)
def test_wgrad_below_matches_torch(M, s, kx0, kx1, keep_gz):
    check_wgrad_below(load_nn_emu(), M, s, kx0, kx1, keep_gz)


def test_wgrad_below_refuses_bad_arguments():
    from go2_rl_gym_amd._nn import Go2nnBwdInJob
    lib = load_nn_emu()
    t = torch.zeros(4096)
    p = t.data_ptr()
    ok = lambda **kw: Go2nnBwdInJob(**{**dict(gz=p, w=p, y_prev=p, gz_prev=None, workspace=p, M=8, C=8, Kin=8, plain=0, w_split=p, ld=0, Kx=4, x_in=p, dw_workspace=p), **kw})
    one = lambda j: lib.go2nn_linear_backward_input_group((Go2nnBwdInJob * 1)(j), 1, None)
    assert one(ok()) == 0
    for bad in (dict(Kx=65), dict(Kx=0), dict(dw_workspace=None), dict(w_split=None), dict(plain=1), dict(ld=16)):
        assert one(ok(**bad)) < 0, bad
    assert lib.go2nn_linear_backward_input_group((Go2nnBwdInJob * 2)(ok(), ok(x_in=None, gz_prev=p)), 2, None) < 0          # x_in on every job or on none
    assert one(ok(x_in=None)) < 0          # no gz_prev without x_in
    assert lib.go2nn_linear_backward_input_fused_rows(0) < 0


def test_group_calls_refuse_bad_arguments():
    from go2_rl_gym_amd._nn import Go2nnBwdWJob, Go2nnFwdJob
    lib = load_nn_emu()
    t = torch.zeros(64)
    j = (Go2nnFwdJob * 3)(*[Go2nnFwdJob(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 2, 2, 2)] * 3)
    assert lib.go2nn_linear_elu_forward_group(j, 3, None) < 0 and lib.go2nn_linear_elu_forward_group(None, 1, None) < 0 and lib.go2nn_linear_elu_forward_group(j, 0, None) < 0
    j[0].x = None
    assert lib.go2nn_linear_elu_forward_group(j, 1, None) < 0 and b"job 0" in lib.go2nn_last_error()
    w = (Go2nnBwdWJob * 2)(Go2nnBwdWJob(t.data_ptr(), t.data_ptr(), None, 4, 4, 4), Go2nnBwdWJob(t.data_ptr(), t.data_ptr(), None, 4, 4, 4))
    assert lib.go2nn_linear_backward_weight_group(w, 2, None) < 0          # no workspace


def test_pair_lins_refuses_what_the_grouped_launches_do_not_cover():
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.actor_critic import _mlp
    fused.set_library(load_oracle()); fused.set_nn_library(load_nn_emu())
    try:
        a, c, c2, c3 = _mlp(8, [16, 8], 4, "elu"), _mlp(6, [16, 8], 1, "elu"), _mlp(6, [16, 12], 1, "elu"), _mlp(6, [16], 1, "elu")
        assert fused.pair_lins(a, c) is not None
        assert fused.pair_lins(a, c2) is None and fused.pair_lins(a, c3) is None and fused.pair_lins(a, a) is None          # other widths, other depth, a 4-wide "value"
        assert fused.pair_lins(_mlp(8, [16, 8], 4, "relu"), c) is None and fused.pair_lins(_mlp(8, [16, 8], 17, "elu"), c) is None
    finally:
        fused.set_library(None); fused.set_nn_library(None)


def ppo_grads_vs_autograd(nn_lib, sim_lib, device, B=300, dims_a=(45, 64, 32, 12), dims_c=(263, 64, 32, 1), clipv=True, atol=3e-6):
    """modules/fused.py:ppo_pair_grads (no autograd: grouped hidden layers + go2nn_ppo_heads + one go2nn_sum_rows) against the reference's eager loss
    (rsl_rl/algorithms/ppo.py:131-170 as PPO._losses states it) differentiated by autograd: statistics and every parameter gradient, including std's"""
    from go2_rl_gym_amd.rsl_rl.algorithms.ppo import PPO
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.actor_critic import ActorCritic
    torch.manual_seed(11)
    A = dims_a[-1]
    ac = ActorCritic(dims_a[0], dims_c[0], A, actor_hidden_dims=list(dims_a[1:-1]), critic_hidden_dims=list(dims_c[1:-1]), init_noise_std=0.8).to(device)
    with torch.no_grad():
        ac.std.mul_(torch.linspace(0.7, 1.3, A, device=device))
    alg = PPO(ac, device=device, lib=None, use_graphs=False, fused_loss=False, entropy_coef=0.01, use_clipped_value_loss=clipv, clip_param=0.2, value_loss_coef=1.0)
    obs, cobs = torch.randn(B, dims_a[0], device=device), torch.randn(B, dims_c[0], device=device)
    with torch.no_grad():
        mu0, v0 = ac.actor(obs), ac.critic(cobs)
    old_mu, old_sig = mu0 + 0.05 * torch.randn_like(mu0), (ac.std.detach() * (1 + 0.05 * torch.randn(A, device=device))).expand(B, A).contiguous()
    act = old_mu + old_sig * torch.randn_like(old_mu)
    old_lp = torch.distributions.Normal(old_mu, old_sig).log_prob(act).sum(-1, keepdim=True)
    adv, tv = torch.randn(B, 1, device=device), v0 + 0.3 * torch.randn(B, 1, device=device)      # value errors on both sides of the clip range
    ret = tv + 0.5 * torch.randn(B, 1, device=device)
    adv[::7] *= 8.0                                                                               # ratios outside the clip range on both sides
    ac.zero_grad()
    loss, vl, sur, kl = alg._losses(obs, cobs, act, tv, adv, ret, old_lp, old_mu, old_sig)
    loss.backward()
    want = [p.grad.clone() for p in ac.parameters()]
    ent = ac.entropy.mean()
    ac.zero_grad(set_to_none=True)
    fused.set_library(sim_lib); fused.set_nn_library(nn_lib)
    try:
        assert fused.ppo_pair_applicable(ac, obs, cobs)
        stats = fused.ppo_pair_grads(ac, obs, cobs, act, tv, adv, ret, old_lp, old_mu, old_sig, 0.2, 1.0, 0.01, clipv)
    finally:
        fused.set_library(None); fused.set_nn_library(None)
    np.testing.assert_allclose(stats.cpu().numpy(), [float(sur), float(vl), float(kl), float(ent)], rtol=2e-5, atol=2e-6)
    for (n, p_), w in zip(ac.named_parameters(), want):
        assert p_.grad is not None and p_.grad.shape == p_.shape, n
        np.testing.assert_allclose(p_.grad.cpu().numpy(), w.cpu().numpy(), atol=atol, rtol=2e-3, err_msg=n)
    return [stats] + [p_.grad for p_ in ac.parameters()]


@pytest.mark.parametrize("dims_a,dims_c,clipv", [((45, 64, 32, 12), (263, 64, 32, 1), True), ((20, 16, 3), (9, 16, 1), False), ((45, 40, 36, 16), (30, 40, 36, 1), True)])
def test_ppo_heads_path_matches_the_eager_loss_under_autograd(dims_a, dims_c, clipv):
    ppo_grads_vs_autograd(load_nn_emu(), load_oracle(), "cpu", dims_a=dims_a, dims_c=dims_c, clipv=clipv)


def test_ppo_heads_refuses_unsupported_shapes():
    lib = load_nn_emu()
    assert lib.go2nn_ppo_heads_cols(17, 128) < 0 and lib.go2nn_ppo_heads_cols(12, 130) < 0 and lib.go2nn_ppo_heads_rows(0, 12, 128) < 0 and lib.go2nn_ppo_heads(None, None) < 0
    assert lib.go2nn_ppo_heads_cols(12, 128) == 4 + 12 + 13 * 128 + 12 + 2 * 128 + 1
