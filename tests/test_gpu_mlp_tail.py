"""go2nn_head_backward (HIP, go2_rl_gym_amd/csrc/go2nn_train.h) on a real MI355X against float64 torch ops, and the fused MLP tail of
modules/fused.py against plain autograd at the real mini-batch shape.  Run with -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_hip  # noqa: E402
from go2_rl_gym_amd import _nn  # noqa: E402
from test_mlp_tail import check_head_backward, tail_vs_autograd  # noqa: E402


@pytest.mark.parametrize("B,Cn,K", [(1, 1, 4), (37, 12, 128), (24576, 12, 128), (24576, 1, 128), (4099, 3, 260), (1000, 16, 512), (98304, 12, 128), (257, 8, 64), (6144, 12, 128)])
def test_head_backward_on_gpu(B, Cn, K):
    """ragged row counts (not a multiple of the workgroup's rows / of the unroll), every template width, K not a power of two; and the sums are
    bit-reproducible from launch to launch (fixed-order reduction, no atomics)."""
    lib = _nn.load_nn()
    a = check_head_backward(lib, B, Cn, K, device="cuda:0")
    b = check_head_backward(lib, B, Cn, K, device="cuda:0")
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("dims", [(45, 512, 256, 128, 12), (263, 512, 256, 128, 1), (263, 512, 256, 32), (225, 512, 256, 8)])
@pytest.mark.parametrize("split", [True, False])
def test_fused_tail_on_gpu(dims, split):
    """the whole-MLP autograd node (modules/fused.py:_FusedMLP) at the update's row count: the actor, the critic, a CTS encoder (32-wide output on the GEMM kernels), the MoE
    gate (8-wide output: go2nn_head_backward); split-operand kernels (the default) and the fp32-MFMA ones (GO2_GEMM_SPLIT=0, the switch that stays for A/B runs)"""
    if not split and dims[0] % 4:
        # (the plain input gradient of a layer whose width is not a multiple of 4 is the one product the fp32-MFMA formulation leaves to the vendor GEMM)
        pass
    tail_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", B=24576, dims=dims, atol=2e-6, split=split)


from test_mlp_tail import check_linear  # noqa: E402


@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (65, 8, 33), (70, 37, 70), (1000, 45, 512), (24576, 45, 512), (24576, 263, 512), (24576, 512, 256), (24576, 256, 128),
                                   (4099, 260, 132), (6144, 512, 256), (300, 100, 300), (8192, 128, 64)])
def test_linear_layer_on_gpu(M, K, N):
    """every tile shape (128x128, 64x128, 64x64), 16-byte and 4-byte load paths (K = 45 / 263 / 37), ragged edges in M, N and K, split and unsplit
    weight gradients; bit-reproducible from launch to launch"""
    lib = _nn.load_nn()
    a = check_linear(lib, M, K, N, device="cuda:0")
    b = check_linear(lib, M, K, N, device="cuda:0")
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_sum_rows_on_gpu():
    from test_mlp_tail import check_sum_rows
    lib = _nn.load_nn()
    a = check_sum_rows(lib, "cuda:0"); b = check_sum_rows(lib, "cuda:0")
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


from test_mlp_tail import check_linear_group  # noqa: E402


@pytest.mark.parametrize("M,s0,s1", [(70, (45, 96), (263, 96)), (777, (37, 70), (64, 70)), (1000, (130, 33), (8, 33)), (3000, (45, 512), (263, 512)),
                                     (24576, (45, 512), (263, 512)), (24576, (512, 256), (512, 256)), (24576, (256, 128), (256, 128)), (6144, (512, 256), (512, 256)),
                                     (4099, (260, 132), (64, 132)), (300, (100, 300), (100, 300))])
def test_linear_group_on_gpu(M, s0, s1):
    """the grouped layer calls (go2nn_gemm3.h): every tile shape, K tails (45 / 263 / 37 / 130), ragged M and N, jobs whose tile shapes differ (separate
    launches), input gradients that fall back to the single-network kernels (Kin not a multiple of 4); bit-reproducible from launch to launch"""
    lib = _nn.load_nn()
    a = check_linear_group(lib, M, s0, s1, device="cuda:0")
    b = check_linear_group(lib, M, s0, s1, device="cuda:0")
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("M,s0,s1", [(70, (45, 96), (263, 96)), (777, (37, 70), (64, 70)), (1000, (130, 33), (8, 33)), (3000, (45, 512), (263, 512)),
                                     (24576, (45, 512), (48, 512)), (24576, (512, 256), (512, 256)), (24576, (256, 128), (256, 128)), (6144, (512, 256), (512, 256)),
                                     (4099, (260, 132), (64, 132)), (300, (100, 300), (100, 300)), (513, (7, 40), (16, 40))])
def test_linear_group_split_operands_on_gpu(M, s0, s1):
    """the ABI 4 kernels (go2nn_bx3.h: fp32 operands as three bf16 planes, six bf16-MFMA terms, fp32 accumulation) against float64 torch within the SAME
    bounds the fp32-MFMA kernels are held to: whole and ragged contractions (45 / 37 / 7: the peeled last k-tile), both tile heights, ragged M and N;
    bit-reproducible from launch to launch"""
    lib = _nn.load_nn()
    a = check_linear_group(lib, M, s0, s1, device="cuda:0", split=True)
    b = check_linear_group(lib, M, s0, s1, device="cuda:0", split=True)
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


from test_mlp_tail import check_wgrad_below  # noqa: E402


@pytest.mark.parametrize("M,s,kx0,kx1,keep_gz", [(24576, (256, 512), 45, 48, False), (24576, (256, 512), 45, 48, True), (6144, (256, 512), 48, 45, False), (70, (96, 40), 45, 48, True),
                                                 (1000, (37, 70), 64, 7, False), (4099, (8, 33), 33, 32, True), (777, (64, 300), 1, 45, False), (128, (16, 128), 32, 64, False)])
def test_wgrad_below_on_gpu(M, s, kx0, kx1, keep_gz):
    """include/go2nn.h ABI 6: the first layer's weight gradient out of the second layer's input-gradient launch (the gradient at the first layer's pre-activation never
    leaves the chip) at the update's shape (24576 x 256 -> 512, inputs 45 / 48 wide) and at ragged ones (rows that do not fill a tile, Kx = 1 / 7 / 32 / 33 / 64, ragged
    contraction, columns beyond the last tile); against float64 within the bounds of the separate launches; bit-reproducible from launch to launch"""
    lib = _nn.load_nn()
    a = check_wgrad_below(lib, M, s, kx0, kx1, keep_gz, device="cuda:0")
    b = check_wgrad_below(lib, M, s, kx0, kx1, keep_gz, device="cuda:0")
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("M,s0,s1", [(6144, (512, 256), (512, 256)), (6144, (256, 128), (256, 128)), (3072, (45, 512), (263, 512))])
def test_split_operand_error_is_the_fp32_kernels_error(M, s0, s1):
    """What "same float64 error" means, as numbers.  On the same inputs the rms deviation from the float64 result of every PRODUCT (forward, input gradient, weight
    gradient) through the split-operand kernels (3 bf16 planes per fp32 value, six MFMA terms) is at most 1.5 x the fp32-MFMA kernels' (measured 0.8 ... 1.2 x;
    a bf16-arithmetic GEMM would be at ~1000 x).  What IS different: the bf16 matrix pipe does not accumulate round-to-nearest-even — the outputs carry a systematic
    bias of about a third of an fp32 ulp (mean error ~ -2e-8 at |y| ~ 0.8, where the fp32 kernels' mean is 1e-9), invisible per element, but it adds up
    along a column: the bias gradient (a column sum over the mini-batch's rows) is off by ~1.5e-6 of its magnitude instead of ~4e-7.  Asserted here as bounds."""
    lib = _nn.load_nn()
    f32, refs = check_linear_group(lib, M, s0, s1, device="cuda:0", split=False, with_refs=True)
    spl, _ = check_linear_group(lib, M, s0, s1, device="cuda:0", split=True, with_refs=True)
    names = ["forward", "input gradient", "bias gradient", "weight gradient"] * 2
    line = []
    for n, a, b, r in zip(names, f32, spl, refs):
        rms = lambda t: float((t.double() - r).pow(2).mean().sqrt())
        ea, eb, scale = rms(a), rms(b), float(r.abs().mean())
        line.append("%s %.2f" % (n, eb / max(ea, 1e-30)))
        if n == "bias gradient":
            assert eb <= 5e-6 * scale, (n, ea, eb, scale)          # the accumulated bias: a few 1e-6 of the sum's magnitude
        else:
            assert eb <= 1.5 * ea + 1e-12, (n, ea, eb)
            assert abs(float((b.double() - r).mean())) <= 2e-7 * scale, n          # the per-output bias itself: a fraction of an ulp
    print("[split / fp32 rms error vs float64, M=%d %s %s] " % (M, s0, s1) + ", ".join(line))


from test_mlp_tail import ppo_grads_vs_autograd  # noqa: E402


@pytest.mark.parametrize("B,dims_a,dims_c,clipv", [(24576, (45, 512, 256, 128, 12), (263, 512, 256, 128, 1), True), (1000, (45, 512, 256, 128, 12), (263, 512, 256, 128, 1), False),
                                                   (4099, (45, 64, 36, 16), (30, 64, 36, 1), True), (777, (20, 32, 3), (9, 32, 1), True)])
def test_ppo_heads_path_on_gpu(B, dims_a, dims_c, clipv):
    """go2nn_ppo_heads + the grouped hidden layers (the no-autograd PPO mini-batch gradient) on the GPU against the eager loss under autograd, at the update's
    shape and at ragged ones (rows that do not fill the last workgroup, every head width template, K = 36 / 32); bit-reproducible from launch to launch"""
    a = ppo_grads_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", B=B, dims_a=dims_a, dims_c=dims_c, clipv=clipv, atol=3e-6)
    b = ppo_grads_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", B=B, dims_a=dims_a, dims_c=dims_c, clipv=clipv, atol=3e-6)
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
