"""Row a16 on the GPU path: the update the benchmark times — HIP-graph capture (algorithms/_graph.py), one permutation gather per
update, actor / critic on two streams, modules/fused.py (in-place ELU, fused ELU-backward + bias gradient, row-split weight gradients),
fused multi-tensor Adam, device-resident learning rate, the fused loss head — against the REFERENCE's goldens
(rsl_rl/rsl_rl/algorithms/ppo.py:120-187 -> ppo_update.npz; on_policy_runner_cts.py:123-202 + cts.py:167-286 -> cts_iteration.npz,
moe_cts_iteration.npz): same weights in, same rollout, same sampling noise, same permutation -> same final weights and learning rate.

Both execution modes: eager on the GPU, and graphs — where the goldens are compared on REPLAYED graphs: warm-up updates run first (the
capture needs them), then weights / optimizer state / learning rate / history are restored in place and the golden iteration is replayed.
Run with -m gpu."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from helpers import ROOT, load_hip  # noqa: E402
import test_cts_golden as tc  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
DEV = "cuda:0"


def _reset_adam(opt):
    for st in opt.state.values():
        for v in st.values():
            if hasattr(v, "zero_"):
                v.zero_()


def _check_lr(lr, golden):
    """The adaptive rate moves by exact factors of 1.5 from 1e-3 (ppo.py:139-151): the same number of steps up / down as the reference; in
    graph mode the rate lives in an fp32 device tensor, so it is the fp32 representation of the reference's float."""
    steps = lambda x: np.log(x / 1e-3) / np.log(1.5)
    assert abs(steps(lr) - round(steps(golden))) < 1e-4 and abs(steps(golden) - round(steps(golden))) < 1e-6, (lr, golden)
    assert abs(lr - golden) <= 2e-7 * golden, (lr, golden)


def _check_weights(sd, g, what):
    # analytic (fused loss head) vs autograd gradients and GPU vs CPU GEMM summation order differ in the last bits; Adam's step g / sqrt(v)
    # is scale-free, so an element whose gradient is ~0 can move by a visible fraction of lr (1e-3..3e-3 here): nearly all elements tight,
    # every element << one step x lr.  (Same bound as the CPU fused-path golden, tests/test_cts_golden.py.)
    for k, v in sd.items():
        w = g["w1_" + k]
        d = np.abs(v.detach().cpu().numpy() - w)
        assert (d <= 5e-6 + 5e-5 * np.abs(w)).mean() >= 0.998 and d.max() < 3e-4, (what, k, float(d.max()), float((d <= 5e-6 + 5e-5 * np.abs(w)).mean()))


@pytest.mark.parametrize("mode", ["eager", "graphs"])
def test_ppo_update_golden_on_gpu(monkeypatch, mode):
    import torch
    from go2_rl_gym_amd.rsl_rl.algorithms import PPO
    from go2_rl_gym_amd.rsl_rl.modules import ActorCritic, fused
    hip = load_hip()
    g = dict(np.load(os.path.join(G, "ppo_update.npz")))
    T, N = g["rew"].shape
    d = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=DEV)
    ac = ActorCritic(45, 263, 12, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", init_noise_std=1.0)
    sd0 = {k[3:]: d(v) for k, v in g.items() if k.startswith("w0_")}
    ac.load_state_dict(sd0)
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=2, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device=DEV, lib=hip,
              use_graphs=(mode == "graphs"))
    assert alg.fused_loss and alg.fused_rollout and fused._LIB is hip and alg.use_graphs == (mode == "graphs")
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
                np.testing.assert_allclose(a.cpu().numpy(), g["actions"][t], atol=2e-6)
                np.testing.assert_allclose(alg.transition.values.cpu().numpy().reshape(-1), g["values"][t].reshape(-1), atol=2e-6)
                np.testing.assert_allclose(alg.transition.actions_log_prob.cpu().numpy().reshape(-1), g["logp"][t], atol=1e-5)
            alg.process_env_step(rew[t], dones[t], {"time_outs": touts[t]})
        alg.compute_returns(cobs[T])

    if mode == "graphs":
        for _ in range(2):          # the first epoch of the first update runs eagerly, the second is captured: after two updates everything replays
            rollout(False)
            alg.update()
        torch.cuda.synchronize()
        assert all(s.graph is not None for s in alg._graph), "HIP-graph capture of the PPO mini-batch step degraded to eager"
        ac.load_state_dict(sd0)
        _reset_adam(alg.optimizer)
        alg.learning_rate = 1e-3
        alg._lr_t.fill_(1e-3)
    rollout(True)
    np.testing.assert_allclose(alg.storage.rewards.cpu().numpy(), g["stored_rewards"], atol=1e-6)
    np.testing.assert_allclose(alg.storage.returns.cpu().numpy(), g["returns"], atol=5e-6)
    np.testing.assert_allclose(alg.storage.advantages.cpu().numpy(), g["advantages"], atol=5e-5)
    mvl, msl = alg.update()
    torch.cuda.synchronize()
    assert abs(mvl - float(g["mean_value_loss"])) < 2e-5 and abs(msl - float(g["mean_surrogate_loss"])) < 2e-5
    _check_lr(alg.learning_rate, float(g["final_lr"]))
    _check_weights(ac.state_dict(), g, "PPO " + mode)
    fused.set_library(None)


class DeviceScriptedEnv(tc.ScriptedEnv):
    """The scripted env of tests/test_cts_golden.py with its tensors in device memory and the HIP library behind `lib`, written so that a
    rollout over it can be captured and replayed: the time index is the rollout storage's own step counter (0 at the start of every
    rollout, T after it, replayed or not) and the actions are copied into a device buffer instead of being pulled to the host."""

    def __init__(self, g, lib):
        import torch
        super().__init__(g, lib)
        for k in ("obs_seq", "priv_seq", "rew_seq", "done_seq", "tout_seq", "episode_length_buf"):
            setattr(self, k, getattr(self, k).to(DEV))
        self.device, self.handle = DEV, None          # (no simulator behind it: go2sim_notify_replayed(NULL) is a no-op error code)
        self.act_buf = torch.zeros(self.rew_seq.shape[0], self.num_envs, 12, device=DEV)
        self.alg = None
        self._ep = {"rew_tracking_lin_vel": torch.tensor(0.25, device=DEV), "terrain_level": torch.tensor(1.5, device=DEV)}

    @property
    def t(self):
        return self.alg.storage.step if self.alg is not None else 0

    @t.setter
    def t(self, v):
        pass

    def step(self, actions):
        t = self.t
        self.act_buf[t].copy_(actions)
        return self.obs_seq[t + 1], self.priv_seq[t + 1], self.rew_seq[t], self.done_seq[t], {"time_outs": self.tout_seq[t], "episode": self._ep}


@pytest.mark.parametrize("mode", ["eager", "graphs"])
@pytest.mark.parametrize("kind,fixture", [("CTS", "cts_iteration.npz"), ("MoECTS", "moe_cts_iteration.npz")])
def test_cts_iteration_golden_on_gpu(kind, fixture, mode, monkeypatch):
    """One full OnPolicyRunnerCTS iteration of the reference (rollout with the history ring, GAE, 2 x 2 policy steps, 2 x 2 student steps)
    through the GPU product path."""
    import torch
    from go2_rl_gym_amd.rsl_rl.modules import ActorCriticCTS, fused
    from go2_rl_gym_amd.rsl_rl.runners import OnPolicyRunnerCTS
    hip = load_hip()
    g = dict(np.load(os.path.join(G, fixture)))
    T, N = g["rew"].shape
    env = DeviceScriptedEnv(g, hip)
    runner = OnPolicyRunnerCTS(env, tc._train_cfg(kind, T), log_dir=None, device=DEV, use_graphs=(mode == "graphs"))
    alg, model = runner.alg, runner.alg.model
    env.alg = alg
    assert runner.use_graphs == alg.use_graphs == (mode == "graphs") and alg.fused_loss and alg.fused_rollout
    np.testing.assert_array_equal(alg.teacher_env_idxs.cpu().numpy(), g["teacher_env_idxs"])
    sd0 = {k[3:]: torch.as_tensor(v, device=DEV) for k, v in g.items() if k.startswith("w0_")}
    model.load_state_dict(sd0)
    noise = torch.as_tensor(g["noise"], device=DEV)
    monkeypatch.setattr(ActorCriticCTS, "_noise", lambda self, like: noise[env.t])
    perms = {len(g["perm_teacher"]): torch.as_tensor(g["perm_teacher"], device=DEV), len(g["perm_student"]): torch.as_tensor(g["perm_student"], device=DEV)}
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: perms[n])
    real_update, seen = alg.update, {}

    def update():
        st = alg.storage
        for k in ("returns", "advantages", "values", "rewards", "actions_log_prob", "history", "observations", "mu"):
            seen[k] = getattr(st, k).cpu().numpy().copy()
        seen["history_after_rollout"] = runner.history.cpu().numpy().copy()
        return real_update()

    alg.update = update
    if mode == "graphs":
        for _ in range(3):          # rollout: 2 eager + capture; update slots: captured at their 4th / 2nd call
            runner.history.zero_()
            runner.learn(1, init_at_random_ep_len=False)
        torch.cuda.synchronize()
        assert runner._rollout_graph is not None, "HIP-graph capture of the rollout degraded to eager"
        assert all(s.graph is not None for grp in alg._steps for s in grp), "HIP-graph capture of a CTS update step degraded to eager"
        model.load_state_dict(sd0)
        _reset_adam(alg.optimizer1); _reset_adam(alg.optimizer2)
        alg.learning_rate = 1e-3
        alg._lr_t.fill_(1e-3)
        runner.history.zero_(); model.history.zero_()
        env.act_buf.zero_()
    runner.learn(1, init_at_random_ep_len=False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(env.act_buf.cpu().numpy(), g["actions"], atol=5e-6)
    np.testing.assert_array_equal(seen["history_after_rollout"], g["history_after_rollout"])      # the ring is pure data movement: exact
    np.testing.assert_array_equal(seen["history"], g["storage_history"])
    np.testing.assert_array_equal(seen["observations"], g["storage_observations"])
    for k, tol in (("values", 5e-6), ("mu", 5e-6), ("actions_log_prob", 2e-5), ("rewards", 5e-6), ("returns", 1e-5), ("advantages", 1e-4)):
        np.testing.assert_allclose(seen[k], g["storage_" + k], atol=tol, err_msg=k)
    _check_lr(alg.learning_rate, float(g["final_lr"]))
    _check_weights(model.state_dict(), g, kind + " " + mode)
    fused.set_library(None)


@pytest.mark.parametrize("mode", ["eager", "graphs"])
def test_full_size_ppo_update_golden_on_gpu(monkeypatch, mode):
    """The reference's PPO.update on the FULL-SIZE go2 networks (45-512-256-128-12 / 263-512-256-128-1; tests/golden/ppo_update_full.npz, 4 Adam
    steps) on the GPU product path — fused policy kernel in the rollout, fused MLP backward, row-split weight gradients, fused loss head, fused
    clip + Adam — eager and from REPLAYED HIP graphs: an element-wise bound over a 155 000-element sample of the 488 857 final weights (every
    element of the small tensors, 20 000 of each large one) and each tensor's sum."""
    import torch
    from go2_rl_gym_amd.rsl_rl.modules import fused
    from test_ppo_golden import check_full_size_weights, run_full_size_update
    hip = load_hip()
    g = dict(np.load(os.path.join(G, "ppo_update_full.npz")))
    alg, ac = run_full_size_update(monkeypatch, g, DEV, hip, use_graphs=(mode == "graphs"), warm=3 if mode == "graphs" else 0)
    torch.cuda.synchronize()
    assert alg.fused_loss and alg.fused_rollout and alg._policy_kernel() is not None
    if mode == "graphs":
        assert alg.graphs_captured()
    _check_lr(alg.learning_rate, float(g["final_lr"]))
    # Adam's step lr * g / sqrt(v) is scale-free: an element whose gradient is ~0 turns last-bit differences (GPU vs CPU summation order, the
    # analytic loss head vs autograd) into a visible fraction of lr.  Each tensor moved by 3.3e-3..3.6e-3 in these 4 steps (fixture: step_*):
    # 99.8 % of the elements within 5e-6 + 5e-5 |w| (0.15 % of that move), EVERY element within 3e-4 (8 %)
    n = check_full_size_weights(ac, g, atol=5e-6, rtol=5e-5, frac=0.998, cap=3e-4)
    assert n > 100000
