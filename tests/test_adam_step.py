"""go2sim_adam_clip_step (include/go2sim.h): adaptive-KL learning rate + clip_grad_norm_ + torch.optim.Adam over a tensor list
(rsl_rl/algorithms/ppo.py:140-155,178-181), as restated by the oracle (plain loops, fp64 inside) and by the host build of the product's
source, against torch's own clip_grad_norm_ + Adam on the same gradients.  CPU only (GPU: tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from helpers import load_emu, load_oracle
from go2_rl_gym_amd.rsl_rl.algorithms._graph import FusedClipAdam


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_fused_clip_adam_matches_torch(which):
    lib = load_oracle() if which == "oracle" else load_emu()
    torch.manual_seed(0)
    shapes = [(64, 45), (64,), (32, 64), (32,), (12, 32), (12,), (12,), (4100, 3)]            # one tensor longer than a 4096-element chunk
    pa = [torch.nn.Parameter(torch.randn(s) * 0.1) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    lra, lrb = torch.tensor(1e-3), 1e-3
    oa = torch.optim.Adam(pa, lr=lra, foreach=False)
    ob = torch.optim.Adam(pb, lr=lrb)
    fa = FusedClipAdam(lib, oa, pa, 1.0)
    assert fa.usable
    for it in range(6):
        scale = [3.0, 0.01, 1.0, 30.0, 0.3, 1.0][it]              # clipped and unclipped steps
        kl = [0.05, 0.001, 0.011, 0.0, 0.004, 0.03][it]           # lr down, up, keep, keep (kl == 0), up, down
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * scale * 1e-2
            a.grad, b.grad = g.clone(), g.clone()
        assert fa.step(torch.tensor(kl), 0.01)
        if kl > 0.02:
            lrb = max(1e-5, lrb / 1.5)
        elif 0.0 < kl < 0.005:
            lrb = min(1e-2, lrb * 1.5)
        for gr in ob.param_groups:
            gr["lr"] = lrb
        torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ob.step()
        assert abs(float(lra) - lrb) < 1e-9 + 2e-7 * lrb
        for a, b in zip(pa, pb):
            np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), atol=2e-7, rtol=2e-6)
            np.testing.assert_allclose(oa.state[a]["exp_avg"].numpy(), ob.state[b]["exp_avg"].numpy(), atol=1e-9, rtol=2e-6)
            np.testing.assert_allclose(oa.state[a]["exp_avg_sq"].numpy(), ob.state[b]["exp_avg_sq"].numpy(), atol=1e-12, rtol=2e-6)
            assert float(oa.state[a]["step"]) == it + 1
    assert set(oa.state_dict()["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}      # the torch optimizer's own state: checkpoints unchanged


@pytest.mark.parametrize("which", ["oracle", "lane_emulation"])
def test_per_tensor_step_counters(which):
    """A parameter that gets no gradient in some step (an unused branch under zero_grad(set_to_none=True)) keeps its own step count in
    torch.optim.Adam, and with it its own bias corrections: the fused kernel reads step[i] per tensor."""
    lib = load_oracle() if which == "oracle" else load_emu()
    torch.manual_seed(1)
    pa = [torch.nn.Parameter(torch.randn(40, 7) * 0.1), torch.nn.Parameter(torch.randn(9) * 0.1), torch.nn.Parameter(torch.randn(5, 5) * 0.1)]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = torch.optim.Adam(pa, lr=torch.tensor(1e-3), foreach=False)
    ob = torch.optim.Adam(pb, lr=1e-3)
    fa = FusedClipAdam(lib, oa, pa, 1e9)
    for it in range(5):
        for k, (a, b) in enumerate(zip(pa, pb)):
            if k == 1 and it in (1, 2):              # tensor 1 sits out two steps
                a.grad = b.grad = None
                continue
            g = torch.randn_like(a) * 1e-2
            a.grad, b.grad = g.clone(), g.clone()
        assert fa.step()
        ob.step()
        for a, b in zip(pa, pb):
            np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), atol=2e-7, rtol=2e-6)
    assert [float(oa.state[a]["step"]) for a in pa] == [5.0, 3.0, 5.0]


def test_unsupported_configurations_fall_back():
    lib = load_oracle()
    p = [torch.nn.Parameter(torch.randn(8))]
    assert not FusedClipAdam(lib, torch.optim.Adam(p, lr=1e-3), p, 1.0).usable                            # float lr: nothing to update in place
    assert not FusedClipAdam(lib, torch.optim.Adam(p, lr=torch.tensor(1e-3), weight_decay=0.1), p, 1.0).usable
    assert not FusedClipAdam(lib, torch.optim.Adam(p, lr=torch.tensor(1e-3), amsgrad=True), p, 1.0).usable
    assert not FusedClipAdam(None, torch.optim.Adam(p, lr=torch.tensor(1e-3)), p, 1.0).usable
    many = [torch.nn.Parameter(torch.randn(2)) for _ in range(60)]
    f = FusedClipAdam(lib, torch.optim.Adam(many, lr=torch.tensor(1e-3), foreach=False), many, 1.0)
    for q in many:
        q.grad = torch.randn_like(q)
    assert f.usable and f.step() is False                                                                 # more tensors than GO2_ADAM_MAX_TENSORS


def test_load_optimizer_state_keeps_addresses():
    """A checkpoint loaded into a runner that has already trained (captured HIP graphs hold the addresses of exp_avg / exp_avg_sq / step):
    the values are the checkpoint's, the tensors are the ones that existed (runners/on_policy_runner.py:load)."""
    import torch
    from go2_rl_gym_amd.rsl_rl.algorithms._graph import load_optimizer_state
    torch.manual_seed(0)
    net = torch.nn.Linear(5, 3)
    lr = torch.tensor(1e-3)
    opt = torch.optim.Adam(net.parameters(), lr=lr, capturable=False)
    for _ in range(3):
        net(torch.randn(7, 5)).square().sum().backward(); opt.step(); opt.zero_grad()
    saved = {k: ({kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in opt.state_dict()["state"].items()}
    sd = {"state": saved, "param_groups": opt.state_dict()["param_groups"]}
    for _ in range(2):
        net(torch.randn(7, 5)).square().sum().backward(); opt.step(); opt.zero_grad()
    before = {p: {k: v for k, v in st.items()} for p, st in opt.state.items()}
    load_optimizer_state(opt, sd)
    for i, (p, st) in enumerate(opt.state.items()):
        for k, v in st.items():
            assert v is before[p][k], (i, k)                                  # same tensor object, same address
            assert torch.equal(v, saved[i][k]), (i, k)                        # the checkpoint's values
    fresh = torch.optim.Adam(torch.nn.Linear(5, 3).parameters(), lr=1e-3)     # no state yet: plain load_state_dict behaviour
    load_optimizer_state(fresh, sd)
    assert all(torch.equal(st["exp_avg"], saved[i]["exp_avg"]) for i, st in enumerate(fresh.state.values()))
