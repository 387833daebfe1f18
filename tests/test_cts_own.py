"""The CTS mini-batch without autograd (go2_rl_gym_amd/rsl_rl/modules/fused_cts.py over include/go2nn.h ABI 5) — here on the host build of the same entry points,
against the reference's formulation (algorithms/cts.py:_policy_losses / _student_losses, eager branch = rsl_rl/rsl_rl/algorithms/cts.py:180-275) differentiated by
autograd.  GPU twin (the HIP kernels): tests/test_gpu_cts_own.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import load_nn_emu, load_oracle

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def latent_pieces_vs_torch(lib, device, n, L, pitch_a, pitch_b):
    """go2nn_latent_concat / go2nn_l2norm_backward / go2nn_latent_mse against float64 torch (F.normalize and its autograd)"""
    g = torch.Generator().manual_seed(n * 100 + L)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if device != "cpu" else None
    z = (torch.randn(n, L, generator=g) * torch.logspace(-2, 1, n).unsqueeze(1)).to(device)          # row norms over three decades
    if n > 2:
        z[n // 2] = 0.0                                                                               # the eps branch of F.normalize
    da, db = torch.full((n, pitch_a), 7.0, device=device), torch.full((n, pitch_b), 7.0, device=device)
    inv = torch.empty(n, device=device)
    assert lib.go2nn_latent_concat(P(z), n, L, P(da), pitch_a, P(db), pitch_b, P(inv), stream) == 0, lib.go2nn_last_error()
    zh = torch.nn.functional.normalize(z.double().cpu(), p=2.0, dim=-1)
    np.testing.assert_allclose(da[:, :L].cpu().numpy(), zh.numpy(), atol=2e-7)
    np.testing.assert_array_equal(da[:, :L].cpu().numpy(), db[:, :L].cpu().numpy())
    assert (da[:, L:] == 7.0).all() and (db[:, L:] == 7.0).all()          # the columns behind the latent are not touched
    # backward through the normaliser: g read from the first L columns of a wider matrix
    gm = torch.randn(n, pitch_a, generator=g).to(device)
    rows = lib.go2nn_l2norm_backward_rows(n)
    dz, part = torch.empty(n, L, device=device), torch.empty(rows * L, device=device)
    assert lib.go2nn_l2norm_backward(P(gm), pitch_a, P(da), pitch_a, P(inv), P(dz), P(part), n, L, stream) == 0, lib.go2nn_last_error()
    zr = z.double().cpu().clone().requires_grad_(True)
    (torch.nn.functional.normalize(zr, p=2.0, dim=-1) * gm[:, :L].double().cpu()).sum().backward()
    scale = (1.0 / z.double().cpu().norm(dim=1).clamp_min(1e-12)).unsqueeze(1)          # (the zero row: g / eps in the kernel and in autograd's clamp_min branch alike)
    assert ((dz.double().cpu() - zr.grad).abs() <= 4e-6 * scale + 1e-12).all()
    np.testing.assert_allclose(part.view(rows, L).double().sum(0).cpu().numpy(), dz.double().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-6 * n * float(dz.abs().max()))
    # the student step's loss head
    zs, zt = torch.randn(n, L, generator=g).to(device), torch.randn(n, L, generator=g).to(device)
    dzs, part2 = torch.empty(n, L, device=device), torch.empty(rows * (L + 4), device=device)
    assert lib.go2nn_latent_mse(P(zs), P(zt), P(dzs), P(part2), n, L, 1.0, stream) == 0, lib.go2nn_last_error()
    zsr = zs.double().cpu().clone().requires_grad_(True)
    loss = (torch.nn.functional.normalize(zt.double().cpu(), dim=-1) - torch.nn.functional.normalize(zsr, dim=-1)).pow(2).mean()
    loss.backward()
    tot = part2.view(rows, L + 4).double().sum(0).cpu()
    assert abs(float(tot[0]) - float(loss)) < 2e-6 * max(1.0, float(loss)) and (tot[1:4] == 0).all()
    np.testing.assert_allclose(dzs.double().cpu().numpy(), zsr.grad.numpy(), atol=2e-7 * float(zsr.grad.abs().max()) + 1e-9, rtol=2e-5)
    np.testing.assert_allclose(tot[4:].numpy(), zsr.grad.sum(0).numpy(), atol=2e-6 * float(zsr.grad.abs().max()) * np.sqrt(n))


@pytest.mark.parametrize("n,L,pa,pb", [(1, 4, 4, 9), (37, 8, 53, 271), (700, 32, 77, 295), (1500, 128, 128, 130)])
def test_latent_pieces_match_torch(n, L, pa, pb):
    latent_pieces_vs_torch(load_nn_emu(), "cpu", n, L, pa, pb)


def test_latent_pieces_refuse_bad_shapes():
    lib = load_nn_emu()
    t = torch.zeros(64)
    assert lib.go2nn_latent_concat(P(t), 2, 12, P(t), 12, None, 0, None, None) < 0 and b"latent concat" in lib.go2nn_last_error()          # L / 4 = 3 is not a power of two
    assert lib.go2nn_latent_concat(P(t), 2, 8, P(t), 4, None, 0, None, None) < 0 and lib.go2nn_latent_concat(P(t), 2, 8, None, 0, None, 0, None, None) < 0
    assert lib.go2nn_l2norm_backward(P(t), 8, P(t), 8, None, P(t), P(t), 2, 8, None) < 0 and lib.go2nn_latent_mse(P(t), None, P(t), P(t), 2, 8, 1.0, None) < 0
    assert lib.go2nn_l2norm_backward_rows(0) < 0


def make_cts(device, kind="CTS", dims=dict(actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], teacher_encoder_hidden_dims=[32, 16], student_encoder_hidden_dims=[32, 16], latent_dim=8),
             num_envs=8, obs=45, priv=60, H=5, **alg_kw):
    from go2_rl_gym_amd.rsl_rl.algorithms import CTS, MoECTS
    from go2_rl_gym_amd.rsl_rl.modules import ActorCriticCTS, ActorCriticMoECTS
    torch.manual_seed(5)
    kw = dict(dims)
    if kind == "MoECTS":
        kw.update(student_encoder_hidden_dims=list(dims["student_encoder_hidden_dims"]) + [8], expert_num=4)
    model = (ActorCriticCTS if kind == "CTS" else ActorCriticMoECTS)(obs, priv, 12, num_envs, H, init_noise_std=0.8, **kw).to(device)
    with torch.no_grad():
        model.std.mul_(torch.linspace(0.7, 1.3, 12, device=device))
    args = dict(device=device, lib=None, use_graphs=False, fused_loss=False, fused_rollout=False, entropy_coef=0.01, clip_param=0.2, value_loss_coef=1.0)
    args.update(alg_kw)
    alg = (CTS if kind == "CTS" else MoECTS)(model, num_envs, H, **args)
    return model, alg


def policy_batch(model, device, B, n_t, obs=45, priv=60, H=5):
    g = torch.Generator().manual_seed(B)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    o, p, h = r(B, obs), r(B, priv), r(B, H * obs)
    with torch.no_grad():
        lat = model.latents(p, h, n_t)
        mu0, v0 = model.policy_mean(lat, o), model.value(lat, o, p)[0]
    old_mu = mu0 + 0.05 * r(B, 12)
    old_sig = (model.std.detach() * (1 + 0.05 * r(12))).expand(B, 12).contiguous()
    act = old_mu + old_sig * r(B, 12)
    old_lp = torch.distributions.Normal(old_mu, old_sig).log_prob(act).sum(-1, keepdim=True)
    adv, tv = r(B, 1), v0 + 0.3 * r(B, 1)
    ret = tv + 0.5 * r(B, 1)
    adv[::7] *= 8.0          # ratios outside the clip range on both sides
    return o, p, h, act, tv, adv, ret, old_lp, old_mu, old_sig


def cts_policy_grads_vs_autograd(nn_lib, sim_lib, device, kind="CTS", B=300, n_t=220, dims=None, atol=3e-6, priv=60):
    """fused_cts.cts_policy_grads against algorithms/cts.py:_policy_losses (eager branch) + autograd: statistics and every gradient of optimizer1's parameters"""
    from go2_rl_gym_amd.rsl_rl.modules import fused, fused_cts
    model, alg = make_cts(device, kind, priv=priv, **({"dims": dims} if dims else {}))
    o, p, h, act, tv, adv, ret, old_lp, old_mu, old_sig = policy_batch(model, device, B, n_t, priv=priv)
    model.zero_grad()
    loss, vl, sur, ent, kl = alg._policy_losses(o, p, h, act, tv, adv, ret, old_lp, old_mu, old_sig, n_t)
    loss.backward()
    want = {n: q.grad.clone() for n, q in model.named_parameters() if q.grad is not None}
    assert not any(n.startswith("student") for n in want)
    model.zero_grad(set_to_none=True)
    fused.set_library(sim_lib); fused.set_nn_library(nn_lib)
    try:
        plan = fused_cts.cts_plan(model)
        assert plan is not None and (plan.student is not None) == (kind == "CTS")
        L = plan.L
        with torch.no_grad():
            lat_s = model.student_latent(h[n_t:])[0]
        ain, cin = torch.zeros(B, L + o.shape[1], device=device), torch.zeros(B, L + p.shape[1], device=device)
        ain[:, L:], cin[:, L:] = o, p
        ain[n_t:, :L], cin[n_t:, :L] = lat_s, lat_s
        acc = torch.full((4,), 10.0, device=device)
        stats = fused_cts.cts_policy_grads(plan, model, ain, cin, p[:n_t], (act, tv, adv, ret, old_lp, old_mu, old_sig), n_t, 0.2, 1.0, 0.01, True, acc=acc)
    finally:
        fused.set_library(None); fused.set_nn_library(None)
    np.testing.assert_allclose(stats.cpu().numpy(), [float(sur), float(vl), float(kl), float(ent)], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(acc.cpu().numpy() - 10.0, stats.cpu().numpy(), atol=1e-5)
    for n, q in model.named_parameters():
        if n in want:
            assert q.grad is not None and q.grad.shape == q.shape, n
            np.testing.assert_allclose(q.grad.cpu().numpy(), want[n].cpu().numpy(), atol=atol, rtol=2e-3, err_msg=n)
        else:
            assert q.grad is None, n          # the student encoder is not part of the policy step
    return [stats] + [q.grad for n, q in model.named_parameters() if n in want]


WIDE = dict(actor_hidden_dims=[64, 32, 16], critic_hidden_dims=[64, 32, 16], teacher_encoder_hidden_dims=[48, 24], student_encoder_hidden_dims=[40, 24], latent_dim=32)


@pytest.mark.parametrize("kind,B,n_t,dims", [("CTS", 300, 220, None), ("MoECTS", 260, 195, None), ("CTS", 129, 97, WIDE)])
def test_cts_policy_step_matches_autograd(kind, B, n_t, dims):
    cts_policy_grads_vs_autograd(load_nn_emu(), load_oracle(), "cpu", kind, B, n_t, dims)


def cts_student_grads_vs_autograd(nn_lib, sim_lib, device, n=150, dims=None, atol=2e-7, priv=60):
    from go2_rl_gym_amd.rsl_rl.modules import fused, fused_cts
    model, alg = make_cts(device, "CTS", priv=priv, **({"dims": dims} if dims else {}))
    g = torch.Generator().manual_seed(n)
    h, p = torch.randn(n, 5 * 45, generator=g).to(device), torch.randn(n, priv, generator=g).to(device)
    model.zero_grad()
    loss, logs = alg._student_losses(h, p)
    loss.backward()
    want = {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}
    assert all(k.startswith("student_encoder") for k in want) and len(want) == 6
    model.zero_grad(set_to_none=True)
    fused.set_library(sim_lib); fused.set_nn_library(nn_lib)
    try:
        plan = fused_cts.cts_plan(model)
        acc = torch.full((1,), 3.0, device=device)
        got = fused_cts.cts_student_grads(plan, model, h, p, acc=acc)
    finally:
        fused.set_library(None); fused.set_nn_library(None)
    assert abs(float(got) - float(loss)) < 2e-6 and abs(float(acc) - 3.0 - float(loss)) < 1e-5
    for k, q in model.named_parameters():
        if k in want:
            np.testing.assert_allclose(q.grad.cpu().numpy(), want[k].cpu().numpy(), atol=atol + 2e-6 * float(want[k].abs().max()), rtol=2e-3, err_msg=k)
        else:
            assert q.grad is None, k
    return [got] + [q.grad for k, q in model.named_parameters() if k in want]


@pytest.mark.parametrize("n,dims", [(150, None), (67, WIDE)])
def test_cts_student_step_matches_autograd(n, dims):
    cts_student_grads_vs_autograd(load_nn_emu(), load_oracle(), "cpu", n, dims)


def test_plan_refuses_what_the_kernels_do_not_cover():
    from go2_rl_gym_amd.rsl_rl.modules import fused, fused_cts
    from go2_rl_gym_amd.rsl_rl.modules import ActorCriticACMoECTS, ActorCriticCTS
    fused.set_library(load_oracle()); fused.set_nn_library(load_nn_emu())
    try:
        base = dict(actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], teacher_encoder_hidden_dims=[32, 16], student_encoder_hidden_dims=[32, 16])
        assert fused_cts.cts_plan(ActorCriticCTS(45, 60, 12, 8, 5, latent_dim=8, **base)) is not None
        assert fused_cts.cts_plan(ActorCriticCTS(45, 60, 12, 8, 5, latent_dim=12, **base)) is None          # 12 / 4 lanes per row is not a power of two
        assert fused_cts.cts_plan(ActorCriticCTS(45, 60, 12, 8, 5, latent_dim=8, norm_type="simnorm", **base)) is None
        assert fused_cts.cts_plan(ActorCriticCTS(45, 60, 12, 8, 5, latent_dim=8, **dict(base, critic_hidden_dims=[32, 8]))) is None          # actor / critic widths differ: no grouped launches
        assert fused_cts.cts_plan(ActorCriticCTS(45, 60, 12, 8, 5, latent_dim=8, activation="relu", **base)) is None
        assert fused_cts.cts_plan(ActorCriticACMoECTS(45, 60, 12, 8, 5, latent_dim=8, expert_num=4, **dict(base, actor_hidden_dims=[32, 16, 8], critic_hidden_dims=[32, 16, 8]))) is None
        fused.set_nn_library(None)
        assert fused_cts.cts_plan(ActorCriticCTS(45, 60, 12, 8, 5, latent_dim=8, **base)) is None
    finally:
        fused.set_library(None); fused.set_nn_library(None)


def chain_vs_autograd(nn_lib, sim_lib, device, B=200, dims=(45, 64, 32, 96), atol=2e-6):
    """modules/fused.py:_FusedChain (an MLP with an ELU behind EVERY layer — the experts' backbone of the MoE encoders — as one autograd node on the split-operand
    kernels) against plain autograd: output, every parameter gradient, the input gradient"""
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.utils import MLP
    torch.manual_seed(9)
    net = MLP(list(dims), "elu", last_activation=True).to(device)
    x, tgt = torch.randn(B, dims[0], device=device, requires_grad=True), torch.randn(B, dims[-1], device=device)
    res = []
    for on in (False, True):
        fused.set_library(sim_lib if on else None); fused.set_nn_library(nn_lib if on else None)
        try:
            net.zero_grad(); x.grad = None
            out = net(x)
            assert (type(out.grad_fn).__name__ == "_FusedChainBackward") == on
            ((out - tgt) ** 2).mean().backward()
            res.append((out.detach().clone(), [q.grad.clone() for q in net.parameters()] + [x.grad.clone()]))
        finally:
            fused.set_library(None); fused.set_nn_library(None)
    np.testing.assert_allclose(res[1][0].cpu().numpy(), res[0][0].cpu().numpy(), atol=atol * 4, rtol=2e-5)
    for a, b in zip(res[1][1], res[0][1]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=atol + 2e-5 * float(b.abs().max()), rtol=2e-3)
    return res[1][1]


@pytest.mark.parametrize("B,dims", [(200, (45, 64, 32, 96)), (77, (225, 40)), (130, (60, 48, 24, 8, 64))])
def test_all_elu_chain_node_matches_autograd(B, dims):
    chain_vs_autograd(load_nn_emu(), load_oracle(), "cpu", B, dims)


def chain_heads_vs_autograd(nn_lib, sim_lib, device, B=200, E=4, dims=(45, 64, 32), hid=16, out=8, atol=2e-6):
    """modules/fused.py:_FusedChainHeads (the experts of a MoE encoder — the shared backbone and its E heads — as one autograd node; the heads' input gradient as
    pitched jobs of the grouped input-gradient kernel, Go2nnBwdInJob.ld) against the modules under plain autograd: outputs, every parameter gradient, the input gradient"""
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.utils import MoE
    torch.manual_seed(11)
    moe = MoE(E, dims[0], list(dims[1:]) + [hid], out, "elu").to(device)
    x, tgt = torch.randn(B, dims[0], device=device, requires_grad=True), torch.randn(E, B, out, device=device)
    res = []
    for on in (False, True):
        fused.set_library(sim_lib if on else None); fused.set_nn_library(nn_lib if on else None)
        try:
            moe.zero_grad(); x.grad = None
            logits, outs, bias = moe.parts(x)
            assert (type(outs.grad_fn).__name__ == "_FusedChainHeadsBackward") == on and outs.shape == (E, B, out)
            (((outs - tgt) ** 2).mean() + (logits ** 2).mean()).backward()
            ps = [q for q in moe.experts.parameters() if q is not bias]
            res.append((outs.detach().clone(), [q.grad.clone() for q in ps] + [x.grad.clone()]))
        finally:
            fused.set_library(None); fused.set_nn_library(None)
    np.testing.assert_allclose(res[1][0].cpu().numpy(), res[0][0].cpu().numpy(), atol=atol * 4, rtol=2e-5)
    for a, b in zip(res[1][1], res[0][1]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=atol + 2e-5 * float(b.abs().max()), rtol=2e-3)


@pytest.mark.parametrize("B,E,dims,hid,out", [(200, 4, (45, 64, 32), 16, 8), (77, 3, (30, 24), 8, 4), (130, 8, (60, 48), 128, 32)])
def test_experts_node_matches_autograd(B, E, dims, hid, out):
    chain_heads_vs_autograd(load_nn_emu(), load_oracle(), "cpu", B, E, dims, hid, out)


def moe_head_vs_autograd(nn_lib, sim_lib, device, n=150, E=8, L=32, coef=0.01, expert_major=False):
    """fused_cts.moe_head_grads (go2nn_moe_usage + go2nn_moe_mix_loss) against the reference's formulation under autograd (modules/utils.py:96-152 MoE.forward +
    the normaliser; moe_cts.py:203-214): both losses, d loss / d gate logits, d loss / d expert outputs"""
    from go2_rl_gym_amd.rsl_rl.modules import fused, fused_cts
    g = torch.Generator().manual_seed(n * 31 + E)
    logits = (torch.randn(n, E, generator=g) * 2).to(device).requires_grad_(True)
    outs = torch.randn(n, E, L, generator=g).to(device).requires_grad_(True)
    t_hat = torch.nn.functional.normalize(torch.randn(n, L, generator=g), dim=-1).to(device)
    w = torch.softmax(logits, dim=-1)
    lat = torch.nn.functional.normalize(torch.sum(w.unsqueeze(-1) * outs, dim=1), p=2.0, dim=-1)
    latent_loss = (t_hat - lat).pow(2).mean()
    lb = (w.mean(dim=0) - 1.0 / E).pow(2).mean()
    (latent_loss + coef * lb).backward()
    fused.set_library(sim_lib); fused.set_nn_library(nn_lib)
    try:
        acc = torch.full((2,), 5.0, device=device)
        if expert_major:          # [E, n, L], the batched GEMM's own layout; the heads' bias added (and differentiated) inside the head
            bias = torch.randn(E * L, generator=g).to(device) * 0.3
            stats, dl, do, dbias = fused_cts.moe_head_grads(logits.detach(), (outs.detach() - bias.view(1, E, L)).transpose(0, 1).contiguous(), t_hat, coef, acc=acc, expert_major=True, bias=bias)
            do = do.transpose(0, 1)
            np.testing.assert_allclose(dbias.view(E, L).cpu().numpy(), outs.grad.sum(0).cpu().numpy(), atol=4e-6 * float(outs.grad.abs().max()) * np.sqrt(n) + 1e-10, rtol=2e-4)
        else:
            stats, dl, do = fused_cts.moe_head_grads(logits.detach(), outs.detach(), t_hat, coef, acc=acc)
    finally:
        fused.set_library(None); fused.set_nn_library(None)
    np.testing.assert_allclose(stats.cpu().numpy(), [float(latent_loss), float(lb)], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(acc.cpu().numpy() - 5.0, stats.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), logits.grad.cpu().numpy(), atol=2e-6 * float(logits.grad.abs().max()) + 1e-10, rtol=2e-4)
    np.testing.assert_allclose(do.cpu().numpy(), outs.grad.cpu().numpy(), atol=2e-6 * float(outs.grad.abs().max()) + 1e-10, rtol=2e-4)
    return [stats, dl, do]


@pytest.mark.parametrize("n,E,L,coef", [(150, 8, 32, 0.01), (1, 4, 8, 1.0), (67, 16, 4, 0.5), (300, 3, 128, 0.0)])
def test_moe_loss_head_matches_autograd(n, E, L, coef):
    moe_head_vs_autograd(load_nn_emu(), load_oracle(), "cpu", n, E, L, coef)
    moe_head_vs_autograd(load_nn_emu(), load_oracle(), "cpu", n, E, L, coef, expert_major=True)


def moe_mix_forward_vs_torch(nn_lib, device, n=150, E=8, L=32, N=400):
    """go2nn_moe_mix_forward (include/go2nn.h ABI 6: the rollout's student rows — softmax gate, weighted sum of the expert outputs + bias, normaliser, scatter into the
    env-ordered latent) against the reference's formulation (modules/utils.py:96-152 MoE.forward + the normaliser, float64 torch): both expert layouts, with and
    without a row map, rows it does not own left untouched"""
    g = torch.Generator().manual_seed(n * 13 + E + L)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if device != "cpu" else None
    logits, outs, bias = (torch.randn(n, E, generator=g) * 2).to(device), torch.randn(n, E, L, generator=g).to(device), (torch.randn(E, L, generator=g) * 0.3).to(device)
    if n > 2:
        outs[n // 2] = 0.0          # (with b = None below: a zero mixture, the eps branch of the normaliser)
    rows = torch.randperm(N, generator=g)[:n].to(torch.int32).to(device)
    want = lambda b: torch.nn.functional.normalize(torch.sum(torch.softmax(logits.double().cpu(), -1).unsqueeze(-1) * (outs.double().cpu() + (b.double().cpu() if b is not None else 0.0)), dim=1), p=2.0, dim=-1)
    for em in (0, 1):
        o = outs.transpose(0, 1).contiguous() if em else outs
        for b in (bias, None):
            z = torch.full((N, L + 4), 7.0, device=device)
            assert nn_lib.go2nn_moe_mix_forward(P(logits), P(o), P(b), P(rows), P(z), L + 4, n, E, L, em, stream) == 0, nn_lib.go2nn_last_error()
            np.testing.assert_allclose(z[rows.long(), :L].cpu().numpy(), want(b).numpy(), atol=3e-7)
            mask = torch.ones(N, dtype=torch.bool); mask[rows.long().cpu()] = False
            assert (z[:, L:] == 7.0).all() and (z[mask.to(device)] == 7.0).all()
            d = torch.full((n, L), 7.0, device=device)
            assert nn_lib.go2nn_moe_mix_forward(P(logits), P(o), P(b), None, P(d), L, n, E, L, em, stream) == 0
            np.testing.assert_allclose(d.cpu().numpy(), want(b).numpy(), atol=3e-7)
    t = torch.zeros(64, device=device)
    assert nn_lib.go2nn_moe_mix_forward(P(t), P(t), None, None, P(t), 8, 2, 17, 8, 0, stream) < 0 and nn_lib.go2nn_moe_mix_forward(P(t), P(t), None, None, P(t), 6, 2, 4, 8, 0, stream) < 0


@pytest.mark.parametrize("n,E,L", [(150, 8, 32), (1, 4, 8), (67, 16, 4), (300, 3, 128)])
def test_moe_mix_forward_matches_torch(n, E, L):
    moe_mix_forward_vs_torch(load_nn_emu(), "cpu", n, E, L)
