"""The CTS mini-batch without autograd (modules/fused_cts.py over include/go2nn.h ABI 5) on a real MI355X: the HIP kernels against float64 torch / the reference's
formulation under autograd, at the update's shapes (24576-row mini-batches: 18432 teacher + 6144 student rows; 45 / 263 / 225-wide inputs, 32-wide latent) and at ragged
ones; bit-reproducible from launch to launch.  Run with -m gpu.  CPU twin on the host build: tests/test_cts_own.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_hip  # noqa: E402
from go2_rl_gym_amd import _nn  # noqa: E402
from test_cts_own import cts_policy_grads_vs_autograd, cts_student_grads_vs_autograd, latent_pieces_vs_torch  # noqa: E402

FULL = dict(actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], teacher_encoder_hidden_dims=[512, 256], student_encoder_hidden_dims=[512, 256], latent_dim=32)
SMALL = dict(actor_hidden_dims=[64, 32, 16], critic_hidden_dims=[64, 32, 16], teacher_encoder_hidden_dims=[48, 24], student_encoder_hidden_dims=[40, 24], latent_dim=8)


@pytest.mark.parametrize("n,L,pa,pb", [(1, 4, 4, 9), (37, 8, 53, 271), (18432, 32, 77, 295), (6144, 32, 32, 32), (1500, 128, 128, 130), (513, 16, 61, 279)])
def test_latent_pieces_on_gpu(n, L, pa, pb):
    latent_pieces_vs_torch(_nn.load_nn(), "cuda:0", n, L, pa, pb)


@pytest.mark.parametrize("kind,B,n_t,dims", [("CTS", 24576, 18432, FULL), ("MoECTS", 24576, 18432, FULL), ("CTS", 1000, 750, FULL), ("CTS", 4099, 3001, SMALL), ("CTS", 300, 220, None)])
def test_cts_policy_step_on_gpu(kind, B, n_t, dims):
    priv = 263 if dims is FULL else 60
    a = cts_policy_grads_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", kind, B, n_t, dims, atol=3e-6, priv=priv)
    b = cts_policy_grads_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", kind, B, n_t, dims, atol=3e-6, priv=priv)
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("n,dims", [(6144, FULL), (1000, FULL), (67, SMALL), (150, None)])
def test_cts_student_step_on_gpu(n, dims):
    priv = 263 if dims is FULL else 60
    a = cts_student_grads_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", n, dims, priv=priv)
    b = cts_student_grads_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", n, dims, priv=priv)
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


from test_cts_own import chain_vs_autograd  # noqa: E402


@pytest.mark.parametrize("B,dims", [(12288, (225, 512, 256, 2048)), (6144, (225, 512, 256, 2048)), (1000, (225, 512, 256, 2048)), (200, (45, 64, 32, 96)), (77, (225, 40))])
def test_all_elu_chain_node_on_gpu(B, dims):
    """the MoE student encoder's backbone (225 -> 512 -> 256 -> 8 x 256, an ELU behind every layer) as one autograd node on the split-operand kernels"""
    chain_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", B, dims, atol=3e-6)


from test_cts_own import chain_heads_vs_autograd  # noqa: E402


@pytest.mark.parametrize("B,E,dims,hid,out", [(12288, 8, (225, 512, 256), 256, 32), (1000, 8, (225, 512, 256), 256, 32), (200, 4, (45, 64, 32), 16, 8), (77, 3, (30, 24), 8, 4)])
def test_experts_node_on_gpu(B, E, dims, hid, out):
    """the MoE student encoder's experts (backbone 225 -> 512 -> 256 -> 8 x 256 and the 8 heads 256 -> 32) as one node: the heads' input gradient written by pitched jobs of
    the grouped split-operand kernel into the [B, 2048] gradient of the backbone's top"""
    chain_heads_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", B, E, dims, hid, out, atol=3e-6)


from test_cts_own import moe_head_vs_autograd  # noqa: E402


@pytest.mark.parametrize("n,E,L,coef", [(12288, 8, 32, 0.01), (6144, 8, 32, 0.01), (1, 4, 8, 1.0), (67, 16, 4, 0.5), (300, 3, 128, 0.0), (1000, 8, 32, 1.0)])
def test_moe_loss_head_on_gpu(n, E, L, coef):
    a = moe_head_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", n, E, L, coef)
    b = moe_head_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", n, E, L, coef)
    moe_head_vs_autograd(_nn.load_nn(), load_hip(), "cuda:0", n, E, L, coef, expert_major=True)
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


from test_cts_own import moe_mix_forward_vs_torch  # noqa: E402


@pytest.mark.parametrize("n,E,L", [(256, 8, 32), (1, 4, 8), (67, 16, 4), (2048, 8, 32), (300, 3, 128)])
def test_moe_mix_forward_on_gpu(n, E, L):
    """the rollout's mixture tail (go2nn_moe_mix_forward_kernel) against float64 torch at the student-row counts of 1024 and 8192 envs and at ragged ones"""
    moe_mix_forward_vs_torch(_nn.load_nn(), "cuda:0", n, E, L, N=max(400, 4 * n))
