"""This build's PPO / RolloutStorage / ActorCritic against one PPO.update of the reference (rsl_rl/algorithms/ppo.py:90-187)
captured by oracle/gen_golden.py: same weights in, same rollout, same permutation -> same weights out."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, load_oracle
from go2_rl_gym_amd.rsl_rl.algorithms import PPO
from go2_rl_gym_amd.rsl_rl.modules import ActorCritic


@pytest.mark.parametrize("fused_rollout", [False, True])
def test_one_update_matches_reference(monkeypatch, fused_rollout):
    """fused_rollout=True: the sampling head and the transition store go through go2sim_act_head / go2sim_store_transition."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_update.npz")))
    T, N = g["rew"].shape
    ac = ActorCritic(45, 263, 12, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", init_noise_std=1.0)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w0_")}
    assert set(sd) == set(ac.state_dict())                      # same parameter names as the reference (checkpoint compatibility)
    ac.load_state_dict(sd)
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=2, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device="cpu", lib=load_oracle(),
              fused_rollout=fused_rollout)
    assert alg.fused_rollout == fused_rollout
    alg.init_storage(N, T, [45], [263], [12])
    obs, cobs = torch.from_numpy(g["obs"]), torch.from_numpy(g["cobs"])
    noise = torch.from_numpy(g["noise"])
    for t in range(T):
        # PPO.act with the recorded sampling noise: a = mu + std * eps
        monkeypatch.setattr(ActorCritic, "_noise", lambda self, like, _t=t: noise[_t])
        a = alg.act(obs[t], cobs[t])
        np.testing.assert_allclose(a.numpy(), g["actions"][t], atol=1e-6)
        np.testing.assert_allclose(alg.transition.values.numpy(), g["values"][t], atol=1e-6)
        np.testing.assert_allclose(alg.transition.actions_log_prob.numpy(), g["logp"][t], atol=1e-5)
        alg.process_env_step(torch.from_numpy(g["rew"][t]), torch.from_numpy(g["dones"][t]).bool(), {"time_outs": torch.from_numpy(g["time_outs"][t]).bool()})
    np.testing.assert_allclose(alg.storage.rewards.numpy(), g["stored_rewards"], atol=1e-6)       # time-out bootstrap (ppo.py:107-108)
    alg.compute_returns(cobs[T])
    np.testing.assert_allclose(alg.storage.returns.numpy(), g["returns"], atol=2e-6)
    np.testing.assert_allclose(alg.storage.advantages.numpy(), g["advantages"], atol=2e-5)
    perm = torch.from_numpy(g["perm"])
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: perm)
    mvl, msl = alg.update()
    assert abs(mvl - float(g["mean_value_loss"])) < 1e-5 and abs(msl - float(g["mean_surrogate_loss"])) < 1e-5
    assert abs(alg.learning_rate - float(g["final_lr"])) < 1e-12
    for k, v in ac.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["w1_" + k], atol=2e-6, rtol=1e-5, err_msg=k)


def test_fused_loss_kernel_matches_autograd():
    """go2sim_ppo_loss (oracle build; the HIP kernel is checked in tests/test_gpu_parity.py) against torch autograd of the
    eager formulation: loss terms, KL and the gradients w.r.t. every parameter."""
    torch.manual_seed(0)
    lib = load_oracle()
    B = 700
    for clipv in (True, False):
        ac = ActorCritic(45, 263, 12, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16])
        with torch.no_grad():
            ac.std.mul_(0.7)
        obs, cobs = torch.randn(B, 45), torch.randn(B, 263)
        with torch.no_grad():
            mu0 = ac.actor(obs); acts = mu0 + 0.8 * torch.randn(B, 12)
            old_mu = mu0 + 0.1 * torch.randn(B, 12); old_sig = 0.8 * torch.ones(B, 12) + 0.05 * torch.rand(B, 12)
            old_lp = torch.distributions.Normal(old_mu, old_sig).log_prob(acts).sum(-1, keepdim=True) + 0.3 * torch.randn(B, 1)
            tv = ac.critic(cobs) + 0.3 * torch.randn(B, 1); ret = tv + torch.randn(B, 1); adv = torch.randn(B, 1)
        grads = []
        for fused in (False, True):
            alg = PPO(ac, clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.01, use_clipped_value_loss=clipv, schedule="adaptive", device="cpu", lib=lib, fused_loss=fused)
            ac.zero_grad()
            loss, vl, sl, kl = alg._losses(obs, cobs, acts, tv, adv, ret, old_lp, old_mu, old_sig)
            loss.backward()
            grads.append((loss.item(), vl.item(), sl.item(), kl.item(), [p.grad.clone() for p in ac.parameters()]))
        (l0, v0, s0, k0, g0), (l1, v1, s1, k1, g1) = grads
        assert abs(l0 - l1) < 2e-6 and abs(v0 - v1) < 2e-6 and abs(s0 - s1) < 2e-6 and abs(k0 - k1) < 2e-6
        for a, b in zip(g0, g1):
            np.testing.assert_allclose(b.numpy(), a.numpy(), atol=2e-7, rtol=2e-4)


def test_update_with_fused_loss_matches_reference(monkeypatch):
    """The same golden update as above, but through the fused loss kernel: identical final weights within fp32 noise."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_update.npz")))
    T, N = g["rew"].shape
    ac = ActorCritic(45, 263, 12, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", init_noise_std=1.0)
    ac.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w0_")})
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=2, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device="cpu", lib=load_oracle(), fused_loss=True)
    alg.init_storage(N, T, [45], [263], [12])
    noise = torch.from_numpy(g["noise"])
    for t in range(T):
        monkeypatch.setattr(ActorCritic, "_noise", lambda self, like, _t=t: noise[_t])
        alg.act(torch.from_numpy(g["obs"][t]), torch.from_numpy(g["cobs"][t]))
        alg.process_env_step(torch.from_numpy(g["rew"][t]), torch.from_numpy(g["dones"][t]).bool(), {"time_outs": torch.from_numpy(g["time_outs"][t]).bool()})
    alg.compute_returns(torch.from_numpy(g["cobs"][T]))
    perm = torch.from_numpy(g["perm"])
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: perm)
    alg.update()
    assert abs(alg.learning_rate - float(g["final_lr"])) < 1e-12
    for k, v in ac.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["w1_" + k], atol=5e-6, rtol=5e-5, err_msg=k)


def test_fused_linear_elu_backward_matches_autograd():
    """modules/fused.py (go2sim_elu_backward_bias through the oracle build) against plain autograd on the same MLP: outputs and all
    parameter / input gradients; and the state-dict layout is the plain nn.Sequential one."""
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from go2_rl_gym_amd.rsl_rl.modules.actor_critic import _mlp
    torch.manual_seed(0)
    net = _mlp(45, [64, 32, 16], 12, "elu")
    assert list(net.state_dict()) == ["0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias", "6.weight", "6.bias"]
    x = torch.randn(300, 45, requires_grad=True)
    tgt = torch.randn(300, 12)
    res = []
    for lib in (None, load_oracle()):
        fused.set_library(lib)
        try:
            net.zero_grad(); x.grad = None
            out = net(x)
            ((out - tgt) ** 2).mean().backward()
            res.append((out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net.parameters()]))
        finally:
            fused.set_library(None)
    (o0, gx0, gp0), (o1, gx1, gp1) = res
    np.testing.assert_allclose(o1.numpy(), o0.numpy(), atol=1e-6)
    np.testing.assert_allclose(gx1.numpy(), gx0.numpy(), atol=1e-7, rtol=1e-4)
    for a, b in zip(gp0, gp1):
        np.testing.assert_allclose(b.numpy(), a.numpy(), atol=1e-7, rtol=1e-4)
    with torch.no_grad():       # inference takes the plain path
        fused.set_library(load_oracle())
        try:
            np.testing.assert_allclose(net(x).numpy(), o0.numpy(), atol=1e-6)
        finally:
            fused.set_library(None)


# ---- the full-size networks of the go2 tasks (go2_config.py:219-221: 512-256-128) ---------------------------------------------------------
def full_size_weights(keys_shapes, seed):
    """oracle/gen_golden.py:full_size_weights — the initial weights are a pure function of numpy's PCG64 stream (the fixture does not carry them)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k, shp in keys_shapes:
        if k == "std":
            out[k] = np.ones(shp, np.float32)
        else:
            fan_in = shp[1] if len(shp) == 2 else None
            b = 1.0 / np.sqrt(fan_in if fan_in else (512 if shp[0] == 512 else shp[0]))
            out[k] = rng.uniform(-b, b, shp).astype(np.float32)
    return out


def run_full_size_update(monkeypatch, g, device, lib, use_graphs=None, fused_rollout=None, warm=0):
    """The reference's full-size PPO.update golden (tests/golden/ppo_update_full.npz) through this build's PPO on `device`.
    warm > 0 (graph mode): that many updates first so that every mini-batch slot is replayed from its HIP graph, then weights / optimizer state /
    learning rate are restored in place and the golden update runs on the replayed graphs.  -> (alg, actor_critic)"""
    T, N = g["rew"].shape
    d = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=device)
    ac = ActorCritic(45, 263, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    w0 = full_size_weights([(k, tuple(v.shape)) for k, v in ac.state_dict().items()], int(g["seed"]))
    sd0 = {k: d(v) for k, v in w0.items()}
    ac.load_state_dict(sd0)
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=2, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device=device, lib=lib,
              use_graphs=use_graphs, fused_rollout=fused_rollout)
    alg.init_storage(N, T, [45], [263], [12])
    obs, cobs, noise = d(g["obs"]), d(g["cobs"]), d(g["noise"])
    rew, dones, touts = d(g["rew"]), d(g["dones"]).bool(), d(g["time_outs"]).bool()
    perm = d(g["perm"])
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: perm)

    def rollout(check):
        for t in range(T):
            monkeypatch.setattr(ActorCritic, "_noise", lambda self, like, _t=t: noise[_t])
            a = alg.act(obs[t], cobs[t])
            if check:
                np.testing.assert_allclose(a.cpu().numpy(), g["actions"][t], atol=5e-6)
                np.testing.assert_allclose(alg.transition.values.cpu().numpy().reshape(-1), g["values"][t].reshape(-1), atol=5e-6)
                np.testing.assert_allclose(alg.transition.actions_log_prob.cpu().numpy().reshape(-1), g["logp"][t].reshape(-1), atol=5e-5)
            alg.process_env_step(rew[t], dones[t], {"time_outs": touts[t]})
        alg.compute_returns(cobs[T])
    for _ in range(warm):
        rollout(False); alg.update()
    if warm:
        with torch.no_grad():
            ac.load_state_dict(sd0)
            for st in alg.optimizer.state.values():
                for v in st.values():
                    if hasattr(v, "zero_"):
                        v.zero_()
        alg.learning_rate = 1e-3; alg._lr_t.fill_(1e-3)
    rollout(True)
    np.testing.assert_allclose(alg.storage.returns.cpu().numpy(), g["returns"], atol=5e-6)
    np.testing.assert_allclose(alg.storage.advantages.cpu().numpy(), g["advantages"], atol=5e-5)
    mvl, msl = alg.update()
    assert abs(mvl - float(g["mean_value_loss"])) < 2e-5 and abs(msl - float(g["mean_surrogate_loss"])) < 2e-5, (mvl, msl)
    return alg, ac


def check_full_size_weights(ac, g, atol, rtol, frac=1.0, cap=None):
    """element by element over the fixture's sample of the 488 857 final weights; `frac` of them within atol + rtol |w| and ALL within `cap`"""
    n = 0
    for k, v in ac.state_dict().items():
        got = v.detach().cpu().numpy().reshape(-1)[g["idx_" + k]]
        d = np.abs(got - g["w1_" + k]); n += d.size
        ok = d <= atol + rtol * np.abs(g["w1_" + k])
        assert ok.mean() >= frac and d.max() <= (cap if cap is not None else np.inf), (k, float(d.max()), float(ok.mean()))
        assert abs(float(v.double().sum()) - float(g["sum1_" + k])) < 1e-4 * max(1.0, v.numel() ** 0.5), k          # the unsampled rest moved the same way in sum
        assert float(g["step_" + k]) > 1e-3          # the 4 Adam steps did move this tensor by ~lr each: the bound below is a small fraction of that
    return n


def test_full_size_update_matches_reference(monkeypatch):
    """512-256-128 networks, 4 Adam steps: this build's eager PPO on the CPU against the reference's, element by element."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_update_full.npz")))
    alg, ac = run_full_size_update(monkeypatch, g, "cpu", load_oracle())
    assert abs(alg.learning_rate - float(g["final_lr"])) < 1e-12
    assert check_full_size_weights(ac, g, atol=2e-6, rtol=1e-5) > 100000
