"""[T, N, .] rollout tensors, GAE(lambda), advantage normalisation, shuffled mini-batches
(rsl_rl/rsl_rl/storage/rollout_storage.py:36-183).

compute_returns runs the library's GAE kernel (one lane per env, reverse scan over T) and normalises with the
mean / unbiased std of ALL advantages of ALL shards: the fp64 partial sums {sum a, sum a^2, n} are all-reduced over
torch.distributed (RCCL over xGMI on a multi-GPU node) — the one data-path collective of the rollout.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.observations = None
            self.critic_observations = None
            self.actions = None
            self.rewards = None
            self.dones = None
            self.values = None
            self.actions_log_prob = None
            self.action_mean = None
            self.action_sigma = None
            self.hidden_states = None

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu", lib=None):
        self.device = device
        self.lib = lib
        T, N = num_transitions_per_env, num_envs
        z = lambda *s, **k: torch.zeros(T, N, *s, device=device, **k)
        self.observations = z(*obs_shape)
        self.privileged_observations = z(*privileged_obs_shape) if privileged_obs_shape[0] is not None else None
        self.rewards, self.actions = z(1), z(*actions_shape)
        self.dones = z(1, dtype=torch.uint8)
        self.actions_log_prob, self.values, self.returns, self.advantages = z(1), z(1), z(1), z(1)
        self.mu, self.sigma = z(*actions_shape), z(*actions_shape)
        self._partials = torch.zeros(3, dtype=torch.float64, device=device)
        self.num_transitions_per_env, self.num_envs = T, N
        self.step = 0

    def add_transitions(self, transition):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        s = self.step
        self.observations[s].copy_(transition.observations)
        if self.privileged_observations is not None:
            self.privileged_observations[s].copy_(transition.critic_observations)
        self.actions[s].copy_(transition.actions)
        self.rewards[s].copy_(transition.rewards.view(-1, 1))
        self.dones[s].copy_(transition.dones.view(-1, 1))
        self.values[s].copy_(transition.values)
        self.actions_log_prob[s].copy_(transition.actions_log_prob.view(-1, 1))
        self.mu[s].copy_(transition.action_mean)
        self.sigma[s].copy_(transition.action_sigma)
        self.step += 1

    def clear(self):
        self.step = 0

    def _stream(self):
        if self.lib.go2sim_is_device_library() == 1:
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def compute_returns(self, last_values, gamma, lam):
        """rollout_storage.py:123-137.  [T,N,1] tensors are contiguous, i.e. exactly the kernel's [T][N] layout."""
        if self.lib is None:
            raise RuntimeError("RolloutStorage needs the go2sim library for its GAE kernel (no torch fallback in the product)")
        T, N = self.num_transitions_per_env, self.num_envs
        last = last_values.reshape(-1).contiguous().float()
        self._partials.zero_()
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = self.lib.go2sim_gae(p(self.rewards), p(self.dones), p(self.values), p(last), p(self.returns), p(self.advantages), p(self._partials),
                                 T, N, float(gamma), float(lam), self._stream())
        if rc != 0:
            raise RuntimeError("go2sim_gae failed: %s" % self.lib.go2sim_last_error().decode())
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("GO2_FORCE_COLLECTIVES", "0") == "1"):
            dist.all_reduce(self._partials, op=dist.ReduceOp.SUM)       # {sum adv, sum adv^2, count}: 24 bytes over RCCL
        rc = self.lib.go2sim_normalize_advantages(p(self.advantages), p(self._partials), T * N, self._stream())
        if rc != 0:
            raise RuntimeError("go2sim_normalize_advantages failed: %s" % self.lib.go2sim_last_error().decode())

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        return (idx[1:] - idx[:-1]).float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        batch_size = self.num_envs * self.num_transitions_per_env
        mini_batch_size = batch_size // num_mini_batches
        indices = torch.randperm(num_mini_batches * mini_batch_size, requires_grad=False, device=self.device)
        flat = lambda t: t.flatten(0, 1)
        obs = flat(self.observations)
        cobs = flat(self.privileged_observations) if self.privileged_observations is not None else obs
        acts, vals, rets = flat(self.actions), flat(self.values), flat(self.returns)
        logp, adv, mu, sig = flat(self.actions_log_prob), flat(self.advantages), flat(self.mu), flat(self.sigma)
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                b = indices[i * mini_batch_size:(i + 1) * mini_batch_size]
                yield obs[b], cobs[b], acts[b], vals[b], adv[b], rets[b], logp[b], mu[b], sig[b], (None, None), None
