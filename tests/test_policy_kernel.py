"""include/go2nn.h: the rollout's policy evaluation (PPO.act, rsl_rl/algorithms/ppo.py:90-102) as one kernel — here the host build of the
same source (packing, padding, operand order, the sampling head, the Python side) against plain PyTorch fp32.  GPU twin (the MFMA kernel
itself): tests/test_gpu_policy_kernel.py."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import load_nn_emu
from go2_rl_gym_amd import _nn
from go2_rl_gym_amd.rsl_rl.modules.actor_critic import ActorCritic


def _ac(seed=0, dims=(512, 256, 128), n_obs=45, n_priv=263, A=12):
    torch.manual_seed(seed)
    ac = ActorCritic(n_obs, n_priv, A, actor_hidden_dims=list(dims), critic_hidden_dims=list(dims), activation="elu", init_noise_std=1.0)
    with torch.no_grad():
        ac.std.copy_(torch.rand(A) * 0.8 + 0.3)
    return ac


def reference_act(ac, obs, priv, eps):
    with torch.no_grad():
        mu, v = ac.actor(obs), ac.critic(priv)
        a = mu + ac.std * eps
        lp = torch.distributions.Normal(mu, mu * 0.0 + ac.std).log_prob(a).sum(-1)
    return a, mu, lp, v.view(-1)


@pytest.mark.parametrize("N", [1, 33, 200])
@pytest.mark.parametrize("dims", [(512, 256, 128), (40, 24)])
def test_policy_act_matches_torch(N, dims):
    lib = load_nn_emu()
    ac = _ac(dims=dims)
    assert _nn.PolicyKernel.supports(ac)
    pk = _nn.PolicyKernel(lib, ac); pk.pack()
    g = torch.Generator().manual_seed(1)
    obs, priv, eps = torch.randn(N, 45, generator=g), torch.randn(N, 263, generator=g) * 2, torch.randn(N, 12, generator=g)
    st = {k: torch.zeros(N, 12) for k in ("a", "mu", "sig")}; lp, v = torch.zeros(N), torch.zeros(N)
    actions = pk.act(obs, priv, eps, st["a"], st["mu"], st["sig"], lp, v)
    a_ref, mu_ref, lp_ref, v_ref = reference_act(ac, obs, priv, eps)
    np.testing.assert_allclose(st["mu"].numpy(), mu_ref.numpy(), atol=2e-6, rtol=2e-6)
    np.testing.assert_allclose(v.numpy(), v_ref.numpy(), atol=2e-6, rtol=2e-6)
    np.testing.assert_allclose(actions.numpy(), a_ref.numpy(), atol=2e-6, rtol=2e-6)
    assert torch.equal(actions, st["a"]) and torch.equal(st["sig"], ac.std.detach().expand(N, 12))
    np.testing.assert_allclose(lp.numpy(), lp_ref.numpy(), atol=2e-5, rtol=2e-6)
    # a = mu + std * eps exactly as the eager two-operation formulation on the kernel's own mu
    assert torch.equal(actions, st["mu"] + ac.std.detach() * eps)


def test_forward_and_repack_follow_the_parameters():
    lib = load_nn_emu()
    ac = _ac(seed=3)
    m = _nn.PackedMlp(lib, ac.critic); m.pack()
    x = torch.randn(70, 263)
    with torch.no_grad():
        np.testing.assert_allclose(m.forward(x).numpy(), ac.critic(x).numpy(), atol=2e-6, rtol=2e-6)
        for p in ac.critic.parameters():
            p.add_(torch.randn_like(p) * 0.05)          # an optimizer step, in place: same tensors, new values
        assert m.still_valid()
        assert np.abs(m.forward(x).numpy() - ac.critic(x).numpy()).max() > 1e-3      # stale until re-packed
        m.pack()
        np.testing.assert_allclose(m.forward(x).numpy(), ac.critic(x).numpy(), atol=2e-6, rtol=2e-6)


def test_unsupported_modules_are_rejected():
    lib = load_nn_emu()
    ac = _ac(dims=(600, 64))
    assert not _nn.PolicyKernel.supports(ac)                       # wider than GO2NN_MAX_WIDTH
    ac2 = ActorCritic(45, 263, 12, actor_hidden_dims=[64], critic_hidden_dims=[64], activation="relu")
    assert not _nn.PolicyKernel.supports(ac2)                      # not ELU
    with pytest.raises(ValueError):
        _nn.PackedMlp(lib, ac2.actor)
    d = _nn.Go2nnMlp(); d.num_layers = 9
    import ctypes as C
    assert lib.go2nn_packed_floats(C.byref(d)) < 0 and b"unsupported" in lib.go2nn_last_error()


def test_ppo_act_through_the_policy_kernel(tmp_path, monkeypatch):
    """PPO.act with the fused policy kernel (host build) inside the runner's rollout against the module-by-module path (ppo.py:90-102): the rows
    of the first step agree to fp32 round-off (later steps see different actions at the 1e-7 level, so only behaviour is compared), the
    weights are re-packed at the start of every rollout, checkpoints and state dicts are untouched."""
    from helpers import load_oracle
    from go2_rl_gym_amd.envs import task_registry
    from go2_rl_gym_amd.utils import get_args
    monkeypatch.setenv("GO2_FUSE_STEP", "1")
    res = {}
    for mode in ("modules", "kernel"):
        args = get_args(["--task", "go2_flat", "--num_envs", "16", "--headless", "--sim_device", "cpu", "--rl_device", "cpu", "--seed", "5"])
        env, _ = task_registry.make_env("go2_flat", args, lib=load_oracle())
        runner, _ = task_registry.make_alg_runner(env, "go2_flat", args, log_root=str(tmp_path / mode))
        alg = runner.alg
        alg.fused_rollout = True
        if mode == "kernel":
            alg.nn_lib = load_nn_emu()
        torch.manual_seed(4)
        runner._rollout(None)
        st = alg.storage
        assert (alg._policy_kernel() is not None) == (mode == "kernel")
        res[mode] = {k: getattr(st, k).clone() for k in ("actions", "mu", "sigma", "values", "actions_log_prob", "observations", "rewards")}
        if mode == "kernel":      # an optimizer step later the next rollout sees the new weights (pack at step 0)
            st.clear()
            with torch.no_grad():
                for p in alg.actor_critic.actor.parameters():
                    p.mul_(0.5)
            runner._rollout(None)
            with torch.no_grad():
                np.testing.assert_allclose(st.mu[3].numpy(), alg.actor_critic.actor(st.observations[3]).numpy(), atol=2e-6, rtol=2e-6)
        env.close()
    a, b = res["modules"], res["kernel"]
    assert torch.equal(a["observations"][0], b["observations"][0]) and torch.equal(a["sigma"], b["sigma"])
    for k in ("mu", "values", "actions"):
        np.testing.assert_allclose(a[k][0].numpy(), b[k][0].numpy(), atol=2e-6, rtol=2e-6, err_msg=k)
    np.testing.assert_allclose(a["actions_log_prob"][0].numpy(), b["actions_log_prob"][0].numpy(), atol=2e-5)
    assert torch.isfinite(b["rewards"]).all() and abs(float(a["rewards"].mean() - b["rewards"].mean())) < 0.05


def cts_policy_kernel_vs_modules(lib, device, kind="CTS", N=203, full=False, atol=3e-6):
    """_nn.PolicyKernelCTS (go2nn_mlp_forward_rows + go2nn_policy_act_latent: CTS.act of one rollout step, rsl_rl/algorithms/cts.py:112-149) against the modules:
    the env-ordered latent of both encoders on their env subsets, actor / critic on [latent | obs] / [latent | priv], the sampling head, the bootstrap value"""
    from go2_rl_gym_amd.rsl_rl.modules import ActorCriticCTS, ActorCriticMoECTS, fused, fused_cts
    torch.manual_seed(7)
    dims = (dict(actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], teacher_encoder_hidden_dims=[512, 256], student_encoder_hidden_dims=[512, 256], latent_dim=32) if full else
            dict(actor_hidden_dims=[40, 24], critic_hidden_dims=[40, 24], teacher_encoder_hidden_dims=[48, 20], student_encoder_hidden_dims=[36, 28], latent_dim=8))
    if kind == "MoECTS":
        dims.update(student_encoder_hidden_dims=dims["student_encoder_hidden_dims"] + [16], expert_num=4)
    n_priv = 263 if full else 61
    model = (ActorCriticCTS if kind == "CTS" else ActorCriticMoECTS)(45, n_priv, 12, N, 5, init_noise_std=1.0, **dims).to(device)
    with torch.no_grad():
        model.std.copy_(torch.rand(12) * 0.8 + 0.3)
    saved = (fused._LIB, fused._NN)
    fused._LIB, fused._NN = object(), lib          # (cts_plan only asks whether the pair is there)
    try:
        plan = fused_cts.cts_plan(model)
    finally:
        fused._LIB, fused._NN = saved
    assert plan is not None and (plan.student is not None) == (kind == "CTS")
    ids = torch.arange(N, device=device)
    ti, si = ids[ids % 4 != 0], ids[ids % 4 == 0]
    pk = _nn.PolicyKernelCTS(lib, model, plan, ti, si); pk.pack()
    g = torch.Generator().manual_seed(2)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    obs, priv, hist, eps = r(N, 45), r(N, n_priv) * 2, r(N, 225), r(N, 12)
    with torch.no_grad():
        lat_ref = torch.empty(N, plan.L, device=device)
        lat_ref[ti] = model.teacher_encoder(priv[ti]); lat_ref[si] = model.student_latent(hist[si])[0]
        mu_ref, v_ref = model.policy_mean(lat_ref, obs), model.evaluate_joint(priv, lat_ref, obs).view(-1)
        a_ref = mu_ref + model.std * eps
        lp_ref = torch.distributions.Normal(mu_ref, mu_ref * 0.0 + model.std).log_prob(a_ref).sum(-1)
    latent = torch.full((N, plan.L), 9.0, device=device)
    if pk.enc_s is None:
        latent[si] = lat_ref[si]          # (the MoE student encoder is the caller's)
    pk.latents(priv, hist, latent)
    np.testing.assert_allclose(latent.cpu().numpy(), lat_ref.cpu().numpy(), atol=atol)
    st = {k: torch.zeros(N, 12, device=device) for k in ("a", "mu", "sig")}; lp, v = torch.zeros(N, device=device), torch.zeros(N, device=device)
    actions = pk.act(latent, obs, priv, eps, st["a"], st["mu"], st["sig"], lp, v)
    np.testing.assert_allclose(st["mu"].cpu().numpy(), mu_ref.cpu().numpy(), atol=atol, rtol=2e-6)
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.cpu().numpy(), atol=atol, rtol=2e-6)
    np.testing.assert_allclose(actions.cpu().numpy(), a_ref.cpu().numpy(), atol=atol, rtol=2e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.cpu().numpy(), atol=10 * atol, rtol=2e-6)
    assert torch.equal(actions, st["a"]) and torch.equal(actions, st["mu"] + model.std.detach() * eps)
    np.testing.assert_allclose(pk.value(latent, priv).view(-1).cpu().numpy(), v_ref.cpu().numpy(), atol=atol, rtol=2e-6)


@pytest.mark.parametrize("kind,N,full", [("CTS", 203, False), ("MoECTS", 64, False), ("CTS", 5, False), ("CTS", 130, True)])
def test_cts_policy_kernel_matches_modules(kind, N, full):
    cts_policy_kernel_vs_modules(load_nn_emu(), "cpu", kind, N, full)


def test_forward_rows_refuses_bad_descriptions():
    import ctypes as C
    lib = load_nn_emu()
    ac = _ac(dims=(40, 24))
    m = _nn.PackedMlp(lib, ac.critic); m.pack()
    x, y = torch.zeros(4, 263), torch.zeros(4, 1)
    descs, packed = (C.POINTER(_nn.Go2nnMlp) * 1)(C.pointer(m.desc)), (C.c_void_p * 1)(m.packed.data_ptr())
    io = lambda **kw: (_nn.Go2nnMlpIO * 1)(_nn.Go2nnMlpIO(**dict(dict(x=x.data_ptr(), x2=None, rows=None, y=y.data_ptr(), ldx=263, ldx2=0, kx=263, nrows=4, ldy=1, normalize=0), **kw)))
    assert lib.go2nn_mlp_forward_rows(descs, packed, io(), 1, None) == 0
    assert lib.go2nn_mlp_forward_rows(descs, packed, io(kx=200), 1, None) < 0 and b"segments" in lib.go2nn_last_error()          # a second segment without x2
    assert lib.go2nn_mlp_forward_rows(descs, packed, io(ldx=100), 1, None) < 0 and lib.go2nn_mlp_forward_rows(descs, packed, io(nrows=0), 1, None) < 0
    assert lib.go2nn_mlp_forward_rows(descs, packed, io(), 3, None) < 0


def test_the_query_names_the_kernel_that_serves_a_network():
    """go2nn_mlp_arith (include/go2nn.h): the host test build answers 0; the HIP library — loaded here without a GPU, the query is host arithmetic — answers 3 (split-operand planes)
    for every network of the BASELINE tasks and 1 (fp32 MFMA) where two neighbouring 512-wide activations do not fit the LDS as planes."""
    import ctypes as C
    from go2_rl_gym_amd import _nn, build
    def desc(dims):
        d = _nn.Go2nnMlp(); d.num_layers = len(dims) - 1
        for i, v in enumerate(dims):
            d.dims[i] = v
        return d
    emu = load_nn_emu()
    assert emu.go2nn_mlp_arith(C.byref(desc((45, 512, 256, 128, 12)))) == 0
    hip = _nn.bind(build.build_nn())
    nets = [(45, 512, 256, 128, 12), (263, 512, 256, 128, 1),          # go2 / go2_flat: actor, critic
            (77, 512, 256, 128, 12), (295, 512, 256, 128, 1), (263, 512, 256, 32), (225, 512, 256, 32)]          # go2_cts: [latent | obs] actor, [latent | priv] critic, teacher / student encoder
    if os.environ.get("GO2_GEMM_SPLIT", "1") == "1":
        assert [hip.go2nn_mlp_arith(C.byref(desc(n))) for n in nets] == [3] * len(nets)
        assert hip.go2nn_mlp_arith(C.byref(desc((300, 512, 512, 12)))) == 1
    assert hip.go2nn_mlp_arith(C.byref(desc((45, 600, 12)))) < 0 and b"unsupported" in hip.go2nn_last_error()
