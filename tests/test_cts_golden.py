"""Scope row f2: this build's OnPolicyRunnerCTS / CTS / MoECTS / RolloutStorageCTS / ActorCritic(MoE)CTS against ONE iteration of
the reference's OnPolicyRunnerCTS.learn (on_policy_runner_cts.py:123-202) captured by oracle/gen_golden.py on a scripted env:
same weights, observations, sampling noise and permutations in -> same actions, history ring, returns, advantages, final
weights, learning rate and checkpoint layout out.  CPU; the library calls (GAE, history ring, fused loss) go to the oracle."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, load_emu, load_oracle
from go2_rl_gym_amd.rsl_rl.modules import ActorCriticCTS
from go2_rl_gym_amd.rsl_rl.runners import OnPolicyRunnerCTS

G = os.path.join(ROOT, "tests", "golden")


class ScriptedEnv:
    """Replays recorded observations / rewards / dones through the VecEnv surface the runner consumes."""

    class _Cfg:
        class env:
            test = True

    def __init__(self, g, lib):
        self.obs_seq, self.priv_seq = torch.from_numpy(g["obs"]), torch.from_numpy(g["priv"])
        self.rew_seq, self.done_seq, self.tout_seq = torch.from_numpy(g["rew"]), torch.from_numpy(g["dones"]).bool(), torch.from_numpy(g["time_outs"]).bool()
        self.num_envs, self.num_obs, self.num_privileged_obs, self.num_actions = self.obs_seq.shape[1], 45, 263, 12
        self.max_episode_length = 1000
        self.episode_length_buf = torch.zeros(self.num_envs, dtype=torch.long)
        self.cfg, self.lib, self.t, self.actions = self._Cfg(), lib, 0, []

    def reset(self):
        return self.obs_seq[0], self.priv_seq[0]

    def get_observations(self):
        return self.obs_seq[self.t]

    def get_privileged_observations(self):
        return self.priv_seq[self.t]

    def step(self, actions):
        self.actions.append(actions.clone().numpy())
        t = self.t
        self.t += 1
        return self.obs_seq[t + 1], self.priv_seq[t + 1], self.rew_seq[t], self.done_seq[t], {
            "time_outs": self.tout_seq[t], "episode": {"rew_tracking_lin_vel": torch.tensor(0.25), "terrain_level": 1.5}}     # as oracle/gen_golden.py's env


class TagRecorder:
    """stands in for the SummaryWriter: the scalar tags a runner writes, in order"""

    def __init__(self):
        self.tags = []

    def add_scalar(self, tag, *a, **k):
        self.tags.append(tag)


def _train_cfg(kind, T):
    policy = dict(init_noise_std=1.0, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], teacher_encoder_hidden_dims=[32, 16],
                  student_encoder_hidden_dims=[32, 16] if kind == "CTS" else [32, 16, 8], activation="elu", latent_dim=8, norm_type="l2norm")
    algorithm = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=2, num_mini_batches=2,
                     learning_rate=1e-3, student_encoder_learning_rate=1e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0,
                     teacher_env_ratio=0.75)
    if kind == "MoECTS":
        policy["expert_num"] = 4
        algorithm["load_balance_coef"] = 0.01
    if kind in ("ACMoECTS", "DualMoECTS"):
        policy.update(expert_num=4, student_encoder_hidden_dims=[32, 16] if kind == "ACMoECTS" else [32, 16, 8], actor_hidden_dims=[32, 16, 8], critic_hidden_dims=[32, 16, 8])
    if kind == "MCPCTS":
        policy.pop("init_noise_std")
        policy.update(actor_hidden_dims=[32, 16], student_expert_num=4, obs_no_goal_mask=[True] * 6 + [False] * 3 + [True] * 36)
    if kind == "MoENGCTS":
        policy.update(student_encoder_hidden_dims=[32, 16], student_expert_num=4, obs_no_goal_mask=[True] * 6 + [False] * 3 + [True] * 36)
        algorithm["load_balance_coef"] = 0.01
    return {"runner": dict(policy_class_name="ActorCritic" + kind, algorithm_class_name=kind, num_steps_per_env=T, max_iterations=1, save_interval=1000,
                           experiment_name="golden", run_name=""), "algorithm": algorithm, "policy": policy, "history_length": 5}


@pytest.mark.parametrize("fused", [False, True, "own"])
@pytest.mark.parametrize("kind,fixture", [("CTS", "cts_iteration.npz"), ("MoECTS", "moe_cts_iteration.npz"), ("MoENGCTS", "moe_ng_cts_iteration.npz"),
                                          ("ACMoECTS", "ac_moe_cts_iteration.npz"), ("DualMoECTS", "dual_moe_cts_iteration.npz"),
                                          ("MCPCTS", "mcp_cts_iteration.npz")])
def test_one_iteration_matches_reference(kind, fixture, fused, monkeypatch, tmp_path):
    """fused: False = the reference's formulation (eager torch); True = the library's loss / rollout heads under autograd; "own" = the graph-mode update run
    uncaptured on the host builds of both libraries — for CTS / MoE-CTS / MoE-NG-CTS the no-autograd mini-batch of modules/fused_cts.py (gather with dst_pitch, student
    latents once per update, explicit launches), for the variants it does not cover the same graph-mode plumbing around the autograd formulation."""
    g = dict(np.load(os.path.join(G, fixture)))
    T, N = g["rew"].shape
    env = ScriptedEnv(g, load_oracle())
    own = fused == "own"
    if own:
        from helpers import load_nn_emu
        from go2_rl_gym_amd.rsl_rl.modules import fused as fmod
        monkeypatch.setattr(fmod, "_LIB", load_oracle()); monkeypatch.setattr(fmod, "_NN", load_nn_emu())
    runner = OnPolicyRunnerCTS(env, _train_cfg(kind, T), log_dir=str(tmp_path), device="cpu", use_graphs="uncaptured" if own else None)
    alg, model = runner.alg, runner.alg.model
    if fused and kind == "MCPCTS":
        pytest.skip("MCP-CTS has a state-dependent std: the fused heads (one std per action dimension) do not apply and the algorithm never takes them")
    alg.fused_loss = alg.fused_rollout = bool(fused)
    if own:
        assert alg.use_graphs and (alg._own_plan() is not None) == (kind in ("CTS", "MoECTS", "MoENGCTS")) and alg._own_student() == (kind == "CTS")
        alg.nn_lib = load_nn_emu()          # the rollout through the host build of the two-launch policy step (go2nn_mlp_forward_rows + go2nn_policy_act_latent)
        assert (alg._policy_kernel() is not None) == (kind in ("CTS", "MoECTS", "MoENGCTS"))
    np.testing.assert_array_equal(alg.teacher_env_idxs.numpy(), g["teacher_env_idxs"])
    np.testing.assert_array_equal(alg.student_env_idxs.numpy(), g["student_env_idxs"])
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w0_")}
    assert set(sd) == set(model.state_dict())            # the reference's parameter names: its checkpoints load unchanged
    model.load_state_dict(sd)
    noise = torch.from_numpy(g["noise"])
    monkeypatch.setattr(ActorCriticCTS, "_noise", lambda self, like: noise[env.t])
    perms = {len(g["perm_teacher"]): torch.from_numpy(g["perm_teacher"]), len(g["perm_student"]): torch.from_numpy(g["perm_student"])}
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: perms[n])
    real_update, seen = alg.update, {}

    def update():
        st = alg.storage
        for k in ("returns", "advantages", "values", "rewards", "actions_log_prob", "history", "observations", "mu"):
            seen[k] = getattr(st, k).numpy().copy()
        seen["history_after_rollout"] = runner.history.numpy().copy()
        return real_update()

    alg.update = update
    runner.writer = TagRecorder()
    runner.learn(1, init_at_random_ep_len=False)
    assert runner.writer.tags == list(g["log_tags"])          # the TensorBoard scalars of on_policy_runner_cts.py:205-256, same names, same order
    np.testing.assert_allclose(np.stack(env.actions), g["actions"], atol=2e-6)
    np.testing.assert_array_equal(seen["history_after_rollout"], g["history_after_rollout"])      # the ring is pure data movement: exact
    np.testing.assert_array_equal(seen["history"], g["storage_history"])
    np.testing.assert_array_equal(seen["observations"], g["storage_observations"])
    for k, tol in (("values", 2e-6), ("mu", 2e-6), ("actions_log_prob", 1e-5), ("rewards", 2e-6), ("returns", 5e-6), ("advantages", 5e-5)):
        np.testing.assert_allclose(seen[k], g["storage_" + k], atol=tol, err_msg=k)
    assert abs(alg.learning_rate - float(g["final_lr"])) < (1e-12 if not own else 1e-6 * float(g["final_lr"]))          # (graph mode keeps the rate in a float32 device tensor)
    for k, v in model.state_dict().items():
        if not fused:
            # eager = the reference's formulation; the expert heads run as a batched GEMM instead of a grouped conv, so a handful of
            # elements with ~0 gradient may land an ulp-scale Adam step apart
            d = np.abs(v.numpy() - g["w1_" + k])
            assert (d <= 2e-6 + 2e-5 * np.abs(g["w1_" + k])).mean() >= 0.999 and d.max() < 5e-5, (k, d.max())
        else:
            # analytic vs autograd gradients differ in the last bits; Adam's step g / sqrt(v) is scale-free, so an element whose
            # gradient is ~0 can move by a visible fraction of lr (3e-3 here): nearly all elements tight, every element << 4 steps x lr
            d = np.abs(v.numpy() - g["w1_" + k])
            assert (d <= 5e-6 + 5e-5 * np.abs(g["w1_" + k])).mean() >= 0.998 and d.max() < 3e-4, (k, d.max())
    # deployment path: act_inference keeps its own history inside the module (actor_critic_cts.py:156-161)
    model.history.zero_()
    obs = torch.from_numpy(g["obs"])
    inf = np.stack([model.act_inference(obs[t]).detach().numpy() for t in range(3)])
    np.testing.assert_allclose(inf, g["act_inference"], atol=2e-5)
    # checkpoint layout (:250-258): same top-level keys, same optimizer param-group structure
    ck = torch.load(os.path.join(runner.log_dir, "model_1.pt"), weights_only=False)
    assert sorted(ck.keys()) == list(g["checkpoint_keys"])
    assert [len(x["params"]) for x in ck["optimizer1_state_dict"]["param_groups"]] == list(g["optimizer1_groups"])
    assert [len(x["params"]) for x in ck["optimizer2_state_dict"]["param_groups"]] == list(g["optimizer2_groups"])


@pytest.mark.parametrize("kind,fixture", [("CTS", "cts_iteration.npz"), ("MoECTS", "moe_cts_iteration.npz")])
def test_update_head_composes_teacher_first_and_advances_the_counter(kind, fixture, monkeypatch, tmp_path):
    """The graph-mode update's HEAD on the host builds with torch.randperm left alone (the goldens above replace randperm, so they never reach
    CTS._update_head): the keyed device-side permutation (go2sim_cts_minibatch_indices) under the algorithm's own key -> every mini-batch of the gathered rollout is
    [teacher rows | student rows] (rollout_storage_cts.py:152-160), every sample is used exactly once, the gathered buffers are the storage's rows in that order, and
    the key's counter advances by one per update (a second head draws another permutation)."""
    from helpers import load_nn_emu
    from go2_rl_gym_amd.rsl_rl.modules import fused as fmod
    g = dict(np.load(os.path.join(G, fixture)))
    T, N = g["rew"].shape
    env = ScriptedEnv(g, load_oracle())
    monkeypatch.setattr(fmod, "_LIB", load_oracle()); monkeypatch.setattr(fmod, "_NN", load_nn_emu())
    torch.manual_seed(5)
    runner = OnPolicyRunnerCTS(env, _train_cfg(kind, T), log_dir=str(tmp_path), device="cpu", use_graphs="uncaptured")
    alg = runner.alg
    alg.fused_loss = alg.fused_rollout = True
    alg.nn_lib = load_nn_emu()
    assert alg.use_graphs and alg._own_plan() is not None
    heads = []
    real_head = alg._update_head

    def head():
        real_head()
        st = alg.storage
        heads.append((alg._order.clone(), {k: alg._perm[k].clone() for k in ("ain", "act", "adv")}, {k: v.clone() for k, v in st.flat().items() if k in ("obs", "act", "adv")}))
    alg._update_head = head
    runner.writer = TagRecorder()
    runner.learn(1, init_at_random_ep_len=False)
    assert len(heads) == 1 and int(alg._shuffle_key[1]) == 1
    order, perm, flat = heads[0]
    nmb, n_t = alg.num_mini_batches, alg._teacher_rows()
    mb = order.numel() // nmb
    assert sorted(order.tolist()) == list(range(T * N))                                  # every sample once (nmb divides both populations in the fixture)
    teacher = set(alg.teacher_env_idxs.tolist())
    for i in range(nmb):
        envs = (order[i * mb:(i + 1) * mb] % N).tolist()                                 # flattened [T, N] storage: index = t N + env
        assert all(e in teacher for e in envs[:n_t]) and not any(e in teacher for e in envs[n_t:]), i
    for k in ("act", "adv"):
        np.testing.assert_array_equal(perm[k].reshape(order.numel(), -1).numpy(), flat[k][order].reshape(order.numel(), -1).numpy())
    L = alg._plan.L
    np.testing.assert_array_equal(perm["ain"][:, L:].numpy(), flat["obs"][order].numpy())          # (the observations land behind the latent's columns of the actor's input matrix)
    alg._update_head()                                                                    # the next update's head: counter 2, another permutation
    assert int(alg._shuffle_key[1]) == 2 and not torch.equal(heads[1][0], order) and sorted(heads[1][0].tolist()) == list(range(T * N))


def test_history_ring_kernel_contract():
    """go2sim_history_push (oracle and the host build of the HIP source) == zero-on-done, shift, append (on_policy_runner_cts.py:155-156)."""
    rng = np.random.default_rng(0)
    N, H, D = 37, 5, 45
    for lib in (load_oracle(), load_emu()):
        hist = rng.normal(size=(N, H, D)).astype(np.float32)
        want = hist.copy()
        for it in range(7):
            obs = rng.normal(size=(N, D)).astype(np.float32)
            dones = (rng.uniform(size=N) < 0.3).astype(np.uint8) if it else None
            if dones is not None:
                want[dones > 0] = 0.0
            want = np.concatenate([want[:, 1:], obs[:, None]], axis=1)
            rc = lib.go2sim_history_push(hist.ctypes.data, obs.ctypes.data, dones.ctypes.data if dones is not None else None, N, H, D, None)
            assert rc == 0
            np.testing.assert_array_equal(hist, want)
        assert lib.go2sim_history_push(None, obs.ctypes.data, None, N, H, D, None) != 0


def test_split_surrogate_kernel_matches_autograd():
    """go2sim_ppo_loss with surrogate_split = teacher rows against torch autograd of cts.py:228-238."""
    from go2_rl_gym_amd.rsl_rl.algorithms import CTS
    torch.manual_seed(1)
    lib = load_oracle()
    B, n_t = 600, 450
    m = ActorCriticCTS(45, 263, 12, 8, 5, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], teacher_encoder_hidden_dims=[32], student_encoder_hidden_dims=[32], latent_dim=8)
    with torch.no_grad():
        m.std.mul_(0.8)
    obs, priv, hist = torch.randn(B, 45), torch.randn(B, 263), torch.randn(B, 225)
    with torch.no_grad():
        lat = m.latents(priv, hist, n_t)
        mu0 = m.actor(torch.cat([lat, obs], 1)); acts = mu0 + 0.8 * torch.randn(B, 12)
        old_mu, old_sig = mu0 + 0.1 * torch.randn(B, 12), 0.8 + 0.05 * torch.rand(B, 12)
        old_lp = torch.distributions.Normal(old_mu, old_sig).log_prob(acts).sum(-1, keepdim=True) + 0.3 * torch.randn(B, 1)
        tv = m.evaluate_joint(priv, lat) + 0.3 * torch.randn(B, 1); ret = tv + torch.randn(B, 1); adv = torch.randn(B, 1)
    res = []
    for fused in (False, True):
        alg = CTS(m, 8, 5, clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.01, schedule="adaptive", device="cpu", lib=lib, fused_loss=fused)
        m.zero_grad()
        loss, vl, sl, ent, kl = alg._policy_losses(obs, priv, hist, acts, tv, adv, ret, old_lp, old_mu, old_sig, n_t)
        loss.backward()
        res.append(([float(x) for x in (loss, vl, sl, ent, kl)], [p.grad.clone() if p.grad is not None else None for p in m.parameters()]))
    for a, b in zip(*[r[0] for r in res]):
        assert abs(a - b) < 3e-6 * max(1.0, abs(a)), (a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert (a is None) == (b is None)
        if a is not None:
            np.testing.assert_allclose(b.numpy(), a.numpy(), atol=3e-7, rtol=3e-4)
    assert all(p.grad is None for p in m.student_encoder.parameters())      # the policy loss never reaches the student encoder


def test_checkpoint_roundtrip(tmp_path):
    """save -> load (with optimizers) restores weights, iteration and learning rate; a checkpoint with FLOAT learning rates
    (the reference's format) loads too."""
    g = dict(np.load(os.path.join(G, "cts_iteration.npz")))
    T = g["rew"].shape[0]
    r1 = OnPolicyRunnerCTS(ScriptedEnv(g, load_oracle()), _train_cfg("CTS", T), log_dir=str(tmp_path / "a"), device="cpu")
    r1.learn(1)
    p = str(tmp_path / "ck.pt")
    r1.save(p)
    r2 = OnPolicyRunnerCTS(ScriptedEnv(g, load_oracle()), _train_cfg("CTS", T), log_dir=None, device="cpu")
    r2.load(p)
    assert r2.current_learning_iteration == 1 and abs(r2.alg.learning_rate - r1.alg.learning_rate) < 1e-12
    for (k, a), b in zip(r1.alg.model.state_dict().items(), r2.alg.model.state_dict().values()):
        assert torch.equal(a, b), k
    st1, st2 = r1.alg.optimizer1.state_dict()["state"], r2.alg.optimizer1.state_dict()["state"]
    assert all(torch.equal(st1[i]["exp_avg"], st2[i]["exp_avg"]) for i in st1)


def test_midrun_checkpoints_carry_their_iteration(tmp_path):
    """on_policy_runner_cts.py:195-199: the CTS runner advances current_learning_iteration BEFORE it saves, so model_{it}.pt holds
    iter = it + 1 and a run resumed from it restores the curricula (train.py: common_step_counter = iter * 24) where they were.  The PPO
    runner of the reference advances only after the loop (on_policy_runner.py:168-171): its mid-run checkpoints carry the start iteration."""
    from go2_rl_gym_amd.rsl_rl.runners import OnPolicyRunner
    g = dict(np.load(os.path.join(G, "cts_iteration.npz")))
    T = g["rew"].shape[0]
    cfg = _train_cfg("CTS", T)
    cfg["runner"]["save_interval"] = 1

    class LoopEnv(ScriptedEnv):
        def step(self, actions):
            out = super().step(actions)
            self.t %= T
            return out
    r = OnPolicyRunnerCTS(LoopEnv(g, load_oracle()), cfg, log_dir=str(tmp_path / "cts"), device="cpu")
    r.writer = TagRecorder(); os.makedirs(str(tmp_path / "cts"))
    r.learn(3)
    for it in range(3):
        assert torch.load(str(tmp_path / "cts" / ("model_%d.pt" % it)), weights_only=False)["iter"] == it + 1
    assert torch.load(str(tmp_path / "cts" / "model_3.pt"), weights_only=False)["iter"] == 3 and r.current_learning_iteration == 3
    r2 = OnPolicyRunnerCTS(LoopEnv(g, load_oracle()), cfg, log_dir=None, device="cpu")
    r2.load(str(tmp_path / "cts" / "model_1.pt"))
    assert r2.current_learning_iteration == 2
    pcfg = {"runner": dict(cfg["runner"], policy_class_name="ActorCritic", algorithm_class_name="PPO"),
            "algorithm": {k: v for k, v in cfg["algorithm"].items() if k not in ("student_encoder_learning_rate", "teacher_env_ratio")},
            "policy": dict(init_noise_std=1.0, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu")}
    p = OnPolicyRunner(LoopEnv(g, load_oracle()), pcfg, log_dir=str(tmp_path / "ppo"), device="cpu")
    p.writer = TagRecorder(); os.makedirs(str(tmp_path / "ppo"))
    p.learn(2)
    assert torch.load(str(tmp_path / "ppo" / "model_1.pt"), weights_only=False)["iter"] == 0
    assert torch.load(str(tmp_path / "ppo" / "model_2.pt"), weights_only=False)["iter"] == 2
