"""This build's PPO / RolloutStorage / ActorCritic against one PPO.update of the reference (rsl_rl/algorithms/ppo.py:90-187)
captured by oracle/gen_golden.py: same weights in, same rollout, same permutation -> same weights out."""
import os

import numpy as np
import torch

from helpers import ROOT, load_oracle
from go2_rl_gym_amd.rsl_rl.algorithms import PPO
from go2_rl_gym_amd.rsl_rl.modules import ActorCritic


def test_one_update_matches_reference(monkeypatch):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ppo_update.npz")))
    T, N = g["rew"].shape
    ac = ActorCritic(45, 263, 12, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", init_noise_std=1.0)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w0_")}
    assert set(sd) == set(ac.state_dict())                      # same parameter names as the reference (checkpoint compatibility)
    ac.load_state_dict(sd)
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=2, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0, entropy_coef=0.01,
              learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device="cpu", lib=load_oracle())
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
