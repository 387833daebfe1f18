"""The fp32-MFMA policy kernel (include/go2nn.h, go2_rl_gym_amd/csrc/go2nn_impl.cpp) on a real MI355X against plain PyTorch fp32 of the same
modules (actor / critic MLPs, Normal log-prob).  Run with -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from go2_rl_gym_amd import _nn  # noqa: E402
from test_policy_kernel import _ac, reference_act  # noqa: E402


@pytest.mark.parametrize("N", [1, 31, 32, 1000, 4096])
@pytest.mark.parametrize("dims", [(512, 256, 128), (40, 24)])
def test_policy_act_on_gpu(N, dims):
    lib = _nn.load_nn()
    ac = _ac(dims=dims).to("cuda:0")
    pk = _nn.PolicyKernel(lib, ac); pk.pack()
    g = torch.Generator(device="cuda:0").manual_seed(1)
    obs, priv, eps = torch.randn(N, 45, device="cuda:0", generator=g), torch.randn(N, 263, device="cuda:0", generator=g) * 2, torch.randn(N, 12, device="cuda:0", generator=g)
    st = {k: torch.zeros(N, 12, device="cuda:0") for k in ("a", "mu", "sig")}; lp, v = torch.zeros(N, device="cuda:0"), torch.zeros(N, device="cuda:0")
    actions = pk.act(obs, priv, eps, st["a"], st["mu"], st["sig"], lp, v)
    torch.cuda.synchronize()
    a_ref, mu_ref, lp_ref, v_ref = reference_act(ac, obs, priv, eps)
    # fp32 MFMA = a k-ordered fmaf chain per output; hipBLASLt sums in another order: a few ulp of the activations' scale
    np.testing.assert_allclose(st["mu"].cpu().numpy(), mu_ref.cpu().numpy(), atol=5e-6, rtol=5e-6)
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.cpu().numpy(), atol=5e-6, rtol=5e-6)
    np.testing.assert_allclose(actions.cpu().numpy(), a_ref.cpu().numpy(), atol=5e-6, rtol=5e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.cpu().numpy(), atol=5e-5, rtol=5e-6)
    assert torch.equal(actions, st["a"]) and torch.equal(actions, st["mu"] + ac.std.detach() * eps)      # the eager formulation's two rounded operations


def test_asymmetric_weights_catch_a_transposed_tile_on_gpu():
    """A = identity-like input with an asymmetric weight matrix: a kernel whose C-tile rows / columns are swapped cannot pass."""
    lib = _nn.load_nn()
    torch.manual_seed(0)
    seq = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.ELU(), torch.nn.Linear(96, 40)).to("cuda:0")
    with torch.no_grad():
        seq[0].weight.copy_(torch.arange(96 * 64, device="cuda:0").float().view(96, 64) / 1000.0 - 3.0); seq[0].bias.zero_()
    m = _nn.PackedMlp(lib, seq); m.pack()
    x = torch.eye(64, device="cuda:0")[:50].contiguous()
    with torch.no_grad():
        np.testing.assert_allclose(m.forward(x).cpu().numpy(), seq(x).cpu().numpy(), atol=2e-5, rtol=1e-5)


from test_policy_kernel import cts_policy_kernel_vs_modules  # noqa: E402


@pytest.mark.parametrize("kind,N,full", [("CTS", 4096, True), ("MoECTS", 8192, True), ("CTS", 1000, True), ("CTS", 203, False), ("CTS", 5, False)])
def test_cts_policy_kernel_on_gpu(kind, N, full):
    """the two-launch CTS policy step (go2nn_mlp_forward_rows + go2nn_policy_act_latent) on the MI355X against the torch modules: row subsets whose sizes differ between
    the two encoders (3 : 1), ragged last workgroups, the 295- and 77-wide two-segment inputs"""
    cts_policy_kernel_vs_modules(_nn.load_nn(), "cuda:0", kind, N, full, atol=6e-6)          # (hipBLASLt sums in another order: a few ulp of the activations' scale, as above)


@pytest.mark.parametrize("dims,kernel", [((45, 512, 256, 128, 12), "planes"), ((263, 512, 256, 128, 1), "planes"), ((77, 100, 50, 7), "planes"), ((17, 33), "planes"),
                                         ((300, 512, 512, 12), "fp32"), ((512, 512), "fp32")])
def test_split_operand_policy_kernel_is_as_close_to_float64_as_fp32(dims, kernel):
    """Round 5: the rollout's MLPs run on the bf16 matrix pipe with every weight and activation split exactly into three bf16 planes (go2nn_mlp3.h).  Against the same
    network in float64 its error is that of an fp32 evaluation (torch fp32 on hipBLASLt beside it), on ragged widths too; two neighbouring 512-wide activations do
    not fit the LDS as planes — those networks stay on the fp32-MFMA kernel (same entry point, same buffer), checked here to give fp32-grade results as well."""
    lib = _nn.load_nn()
    torch.manual_seed(3)
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(torch.nn.ELU())
    seq = torch.nn.Sequential(*layers).to("cuda:0")
    m = _nn.PackedMlp(lib, seq); m.pack()
    x = torch.randn(997, dims[0], device="cuda:0") * 1.5
    with torch.no_grad():
        y = m.forward(x).double(); y32 = seq(x).double(); y64 = seq.double()(x.double())
    scale = float(y64.abs().max())
    e_kernel, e_torch = float((y - y64).abs().max()) / scale, float((y32 - y64).abs().max()) / scale
    assert e_kernel < 2e-6 and e_kernel < 4.0 * e_torch + 2e-7, (kernel, e_kernel, e_torch)


def test_the_baseline_networks_run_on_the_split_operand_kernel():
    """no silent fall-back: the library reports the kernel that serves each network (go2nn_mlp_arith), and for the tasks' shapes that is the split-operand one"""
    import os
    lib = _nn.load_nn()
    pk = _nn.PolicyKernel(lib, _ac(dims=(512, 256, 128)).to("cuda:0"))
    want = 3 if os.environ.get("GO2_GEMM_SPLIT", "1") == "1" else 1
    assert (pk.actor.arith, pk.critic.arith) == (want, want)
