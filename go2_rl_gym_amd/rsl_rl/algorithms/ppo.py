"""PPO with clipped surrogate / clipped value loss / adaptive-KL learning rate (rsl_rl/rsl_rl/algorithms/ppo.py:38-187).

Two execution modes with identical arithmetic:
  * eager  — the reference's control flow, Python-side learning-rate decision (`.item()` syncs); used on CPU and as the
             numerical reference (tests/test_ppo_golden.py pins it against the reference's own update).
  * graphs — on a GPU: one mini-batch update (gather, forward, losses, backward, gradient clip, Adam step, KL -> learning
             rate) is captured once into a HIP graph and replayed 20x per iteration; the learning rate lives in a device
             tensor, so there is no host sync and no per-op launch overhead inside the update.

Multi-GPU (one process per GPU, torch.distributed backend 'nccl' = RCCL): each rank owns an env shard and a full policy
replica; gradients are averaged across ranks before the clip/step and the mean KL before the learning-rate decision, so
every rank takes the same branch; advantage statistics are all-reduced in RolloutStorage.compute_returns.

The env's observation buffers are persistent device buffers that the step kernel overwrites IN PLACE (the reference
rebinds `obs_buf` to fresh tensors every step), so `act()` copies the observations into the rollout storage immediately
instead of keeping a reference until `process_env_step`.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim

from ..modules.actor_critic import ActorCritic as _AC
from ..storage import RolloutStorage
from ._graph import CapturedStep, FusedClipAdam, GradBucket, ReducedStep, all_captured


_ADAM_IMPL = {"fused": True}          # torch's capturable Adam where the library's clip + Adam kernel does not serve (GO2_FUSED_ADAM=0, a variant it does not cover)


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _collectives_on():
    """True when the cross-rank averaging must run: more than one rank, or GO2_FORCE_COLLECTIVES=1 with an initialised process
    group of any size (lets a 1-GPU box exercise the RCCL-inside-HIP-graph path)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("GO2_FORCE_COLLECTIVES", "0") == "1"


def allreduce_mean_bucket(grads, world, extra=None):
    parts = [g.reshape(-1) for g in grads] + ([extra.detach().reshape(1)] if extra is not None else [])
    flat = torch.cat(parts)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    views, off = [], 0
    for g in grads:
        n = g.numel(); views.append(flat[off:off + n].view_as(g)); off += n
    torch._foreach_copy_(grads, views)           # one multi-tensor launch instead of one copy per parameter
    return flat[off] if extra is not None else None


_RANDPERM = torch.randperm          # (the golden tests replace torch.randperm to replay the reference's permutation: then the update takes that one)


from ..modules.actor_critic_cts import ActorCriticCTS as _ACC          # noqa: E402
_PLAIN = (_AC._noise, _ACC._noise)          # the modules' own _noise functions as they are at import (the tests replace them to inject the reference's draws: then the rollout asks per step)


class _RolloutHeads:
    """Shared pieces of the PPO-family algorithms: two-stream actor/critic evaluation and the per-step rollout heads."""
    _eps_all = None
    _img_cache = None           # split weight images of the networks a rollout evaluates as torch modules on the library's kernels (modules/fused.py:own_forward(images=...)); emptied with _pk_packed
    _pk_packed = False          # the policy kernel's packed weights are those of the current parameters (they change in update() only)
    _pk_recorded = False        # the rollout last RUN THROUGH PYTHON (eager, or while being captured) packed at its first step: what a replay of that capture does too

    def _rollout_noise(self, ac, st, s):
        """The standard-normal draws of step s ([N, A], the shape of the action rows).  One launch for the whole rollout at its first step instead of
        one per step (24 small launches on the chain of dependent kernels); the same law as Normal.sample() per step (actor_critic.py:123-125).
        A module whose _noise is overridden — the tests inject the reference's draws step by step — is asked per step as before."""
        if getattr(type(ac), "_noise", None) not in _PLAIN:
            return ac._noise(st.actions[s])
        if s == 0 or self._eps_all is None or self._eps_all.shape != st.actions.shape or self._eps_all.device != st.actions.device:
            self._eps_all = torch.randn_like(st.actions)
        return self._eps_all[s]

    # The update's permutation in graph mode: a keyed bijection computed on the device (include/go2sim_shuffle.h) under (seed, counter).  The seed follows
    # torch.manual_seed at the time the key is made (the first graph-mode update), NOT torch's generator state afterwards; the counter advances by one per update on the
    # device and is set to the iteration number when a checkpoint is loaded (runner.load -> set_shuffle_counter), so a resumed run does not replay the permutations of
    # iterations 0, 1, ....  DESIGN.md section 8 lists this among the deliberate deviations.
    _shuffle_key = None
    _shuffle_counter0 = 0

    def _make_shuffle_key(self):
        seed = int((torch.initial_seed() * 0x9E3779B1 + 0x7F4A7C15) & 0x7FFFFFFF)          # (follows torch.manual_seed without drawing from the generator)
        self._shuffle_key = torch.tensor([seed, int(self._shuffle_counter0) & 0x7FFFFFFF, 0, 0], dtype=torch.int32, device=self.device)
        return self._shuffle_key

    def set_shuffle_counter(self, it):
        self._shuffle_counter0 = int(it)
        if self._shuffle_key is not None:
            self._shuffle_key[1] = int(it) & 0x7FFFFFFF

    def rollout_replayed(self):
        """The runner replayed the captured rollout (act() did not run in Python): the weights are packed iff the captured rollout recorded the pack launch at its
        first step — known from the flag act() left when it ran under capture (a rollout that took the non-kernel branch must not claim packed weights)."""
        self._pk_packed = bool(self._pk_recorded)

    def _pair(self, main_fn, side_fn, enabled=True):
        """-> (main_fn(), side_fn()).  Rounds 2-4 ran side_fn on a second HIP stream; with the grouped launches (PPO, CTS, MoE-CTS never reach this in their updates) the fork
        only served network shapes outside the BASELINE configurations, and two hipBLASLt stream-K kernels side by side were the hang of round 3 that was never
        explained — the second stream is gone."""
        return main_fn(), side_fn()

    def _actor_critic(self, ac, obs, cobs):
        """-> (ac.actor(obs), ac.evaluate(cobs)) under autograd: one node per network (modules/fused.py:_FusedMLP)"""
        return ac.actor(obs), ac.evaluate(cobs)

    # The two per-step element-wise heads of the rollout as library kernels (go2sim_act_head, go2sim_store_transition): sampling +
    # log-prob + the storage rows in one launch, reward bootstrap + done copy in another, instead of ~23 small launches.

    def _ptr(self, t):
        import ctypes as C
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def _stream(self, t):
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None

    def _act_head(self, mu, std, eps, value, s):
        st, t = self.storage, self.transition
        mu, eps, value = mu.detach().contiguous(), eps.contiguous(), value.detach().contiguous()
        actions = torch.empty_like(mu)
        p = self._ptr
        rc = self.lib.go2sim_act_head(p(mu), p(std.detach()), p(eps), p(value), p(actions), p(st.actions[s]), p(st.mu[s]), p(st.sigma[s]),
                                      p(st.actions_log_prob[s]), p(st.values[s]), mu.shape[0], mu.shape[1], self._stream(mu))
        if rc != 0:
            raise RuntimeError("go2sim_act_head failed: %s" % self.lib.go2sim_last_error().decode())
        t.actions, t.values, t.actions_log_prob = actions, st.values[s], st.actions_log_prob[s].view(-1)
        t.action_mean, t.action_sigma = st.mu[s], st.sigma[s]
        return actions

    def rollout_targets(self):
        """Destinations for LeggedRobot.step(rollout=...) of the step being collected: the env kernel writes the next observations into the
        next storage rows and this step's reward / done rows itself (None: not applicable)."""
        st = self.storage
        if not self.fused_rollout or st.privileged_observations is None or st.dones.dtype != torch.uint8:
            return None
        s = st.step
        nxt = s + 1 < st.num_transitions_per_env
        return {"obs_out": st.observations[s + 1] if nxt else None, "priv_out": st.privileged_observations[s + 1] if nxt else None,
                "values": st.values[s].view(-1), "rewards_out": st.rewards[s].view(-1), "dones_out": st.dones[s].view(-1), "gamma": self.gamma}

    def _store_transition(self, rewards, dones, infos, s):
        if isinstance(infos, dict) and infos.get("transition_stored"):      # (the env step stored it: go2sim_step_rollout)
            return
        st = self.storage
        touts = infos.get("time_outs") if isinstance(infos, dict) else None
        rewards = rewards.contiguous().float()
        as_u8 = lambda x: None if x is None else (x if x.dtype == torch.uint8 else x.contiguous().view(torch.uint8) if x.dtype == torch.bool else x.to(torch.uint8))
        d, to = as_u8(dones), as_u8(touts.to(self.device) if touts is not None else None)
        p = self._ptr
        rc = self.lib.go2sim_store_transition(p(rewards), p(d), p(to), p(st.values[s]), p(st.rewards[s]), p(st.dones[s]), float(self.gamma), rewards.shape[0], self._stream(rewards))
        if rc != 0:
            raise RuntimeError("go2sim_store_transition failed: %s" % self.lib.go2sim_last_error().decode())


class _FusedPPOLoss(torch.autograd.Function):
    """loss = surrogate + c_v * value_loss - c_e * entropy through ONE library kernel (go2sim_ppo_loss) that also returns the
    analytic gradients w.r.t. mu, std and value; autograd then continues into the actor / critic MLPs.  Replaces ~150
    element-wise launches of the eager formulation per mini-batch with 2."""

    @staticmethod
    def kernel(alg, mu, std, value, act_b, tv_b, adv_b, ret_b, old_lp_b, old_mu_b, old_sig_b):
        """-> stats [surrogate, value loss, kl, entropy, loss], d loss / d mu, / d std, / d value (shaped like `value`)"""
        import ctypes as C
        B, A = mu.shape
        c = lambda t: t.detach().contiguous().float()
        mu_c, std_c, val_c = c(mu), c(std), c(value).view(-1)
        args = [mu_c, std_c, val_c, c(act_b), c(old_mu_b), c(old_sig_b), c(old_lp_b).view(-1), c(adv_b).view(-1), c(tv_b).view(-1), c(ret_b).view(-1)]
        gmu, gstd, gval = torch.empty_like(mu_c), torch.empty_like(std_c), torch.empty_like(val_c)
        stats = torch.empty(5, device=mu.device)
        ws = torch.empty(24 * ((B + 63) // 64), device=mu.device)       # (go2sim.h: 24 floats per block of 64 rows)
        lib = alg.lib
        stream = C.c_void_p(torch.cuda.current_stream(mu.device).cuda_stream) if mu.is_cuda else None
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = lib.go2sim_ppo_loss(*[p(t) for t in args], p(gmu), p(gstd), p(gval), p(stats), p(ws), B, A, float(alg.clip_param), float(alg.value_loss_coef),
                                 float(alg.entropy_coef), int(alg.use_clipped_value_loss), int(getattr(alg, "surrogate_split", 0)), stream)
        if rc != 0:
            raise RuntimeError("go2sim_ppo_loss failed: %s" % lib.go2sim_last_error().decode())
        return stats, gmu, gstd, gval.view_as(value)

    @staticmethod
    def forward(ctx, mu, std, value, alg, act_b, tv_b, adv_b, ret_b, old_lp_b, old_mu_b, old_sig_b):
        stats, gmu, gstd, gval = _FusedPPOLoss.kernel(alg, mu, std, value, act_b, tv_b, adv_b, ret_b, old_lp_b, old_mu_b, old_sig_b)
        ctx.save_for_backward(gmu, gstd, gval)
        ctx.mark_non_differentiable(stats)
        return stats[4].clone(), stats

    @staticmethod
    def backward(ctx, g_loss, g_stats):
        gmu, gstd, gval = ctx.saved_tensors
        return gmu * g_loss, gstd * g_loss, gval * g_loss, None, None, None, None, None, None, None, None


class PPO(_RolloutHeads):
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95, value_loss_coef=1.0,
                 entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01,
                 device="cpu", lib=None, use_graphs=None, fused_loss=None, fused_rollout=None):
        self.device = device
        self.lib = lib
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic
        self.actor_critic.to(self.device)
        self.storage = None
        on_gpu = str(device).startswith("cuda")
        self.use_graphs = on_gpu if use_graphs is None else bool(use_graphs and on_gpu)
        self._capture = self.use_graphs and use_graphs != "uncaptured"
        if use_graphs == "uncaptured":     # the graph-mode update (device-side LR decision, permuted chunks, split all-reduce) run eagerly on any
            self.use_graphs = True         # device: how the CPU tests cover it
        if self.use_graphs:
            self._lr_t = torch.tensor(float(learning_rate), device=device)
            self.optimizer = (optim.Adam(self.actor_critic.parameters(), lr=self._lr_t, capturable=True, **_ADAM_IMPL) if self._capture else
                              optim.Adam(self.actor_critic.parameters(), lr=self._lr_t, foreach=False))
        else:
            self._lr_t = None
            self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=learning_rate)
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self._graph = None
        self._fused_adam = None
        # PPO.act as ONE fp32-MFMA launch (include/go2nn.h): both MLPs + the sampling head; on the GPU by default when the modules are plain
        # Linear / ELU stacks (GO2_FUSED_POLICY=0 restores the hipBLASLt chains + go2sim_act_head).  nn_lib: tests hand in the host build.
        self.nn_lib, self._pk = None, None
        self._pk_on = on_gpu and lib is not None and os.environ.get("GO2_FUSED_POLICY", "1") == "1"
        # the fused loss kernel is the default on the GPU; on the CPU it is opt-in (tests compare it with the eager formulation)
        self.fused_loss = (on_gpu and lib is not None) if fused_loss is None else bool(fused_loss and lib is not None)
        self.fused_rollout = (on_gpu and lib is not None) if fused_rollout is None else bool(fused_rollout and lib is not None)
        if on_gpu and lib is not None and os.environ.get("GO2_FUSED_MLP", "1") == "1":
            from ..modules import fused
            fused.set_library(lib)         # Linear->ELU pairs: activation + bias gradients in one HBM pass (go2sim_elu_backward_bias), ELU in place
                                           # (+6 % whole-job, measured; GO2_FUSED_MLP=0 restores plain autograd)
        if _world() > 1:   # identical initial replicas
            for p in self.actor_critic.parameters():
                dist.broadcast(p.data, src=0)

    def rebind_lr(self):
        """After optimizer.load_state_dict (which may bring a float lr, e.g. from a reference checkpoint): put the device-resident
        learning-rate tensor the captured graph reads and writes back into the param groups."""
        lr = self.optimizer.param_groups[0]["lr"]
        self.learning_rate = float(lr)
        if self._lr_t is not None:
            self._lr_t.fill_(self.learning_rate)
            for g in self.optimizer.param_groups:
                g["lr"] = self._lr_t

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device, lib=self.lib)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    # ------------------------------------------------------------------ rollout half (ppo.py:90-118)
    def act(self, obs, critic_obs):
        st, t, ac = self.storage, self.transition, self.actor_critic
        s = st.step
        if s >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        if self.fused_rollout:
            if obs.data_ptr() != st.observations[s].data_ptr():            # (the env wrote this row itself: LeggedRobot.step(rollout=...))
                st.observations[s].copy_(obs)
            if st.privileged_observations is not None and critic_obs.data_ptr() != st.privileged_observations[s].data_ptr():
                st.privileged_observations[s].copy_(critic_obs)
            t.observations, t.critic_observations = obs, critic_obs
            pk = self._policy_kernel()
            if pk is not None and obs.is_contiguous() and critic_obs.is_contiguous() and obs.dtype == torch.float32 and critic_obs.dtype == torch.float32:
                if s == 0 or not self._pk_packed:
                    pk.pack()                  # the parameters only change in update(): once per rollout (inside the captured rollout graph too); a rollout whose
                    self._pk_packed = True     # first steps took the eager branch packs at its first kernel step
                    self._pk_recorded = self._pk_recorded or s == 0
                actions = pk.act(obs, critic_obs, self._rollout_noise(ac, st, s), st.actions[s], st.mu[s], st.sigma[s], st.actions_log_prob[s].view(-1), st.values[s].view(-1))
                t.actions, t.values, t.actions_log_prob = actions, st.values[s], st.actions_log_prob[s].view(-1)
                t.action_mean, t.action_sigma = st.mu[s], st.sigma[s]
                return actions
            if s == 0:
                self._pk_recorded = False      # (this rollout does not start on the kernel: a replay of it must not claim packed weights)
            mu, value = self._pair(lambda: ac.actor(obs), lambda: ac.evaluate(critic_obs), enabled=self._capture)
            return self._act_head(mu, ac.std, self._rollout_noise(ac, st, s), value, s)
        t.actions = ac.act(obs).detach()
        t.values = ac.evaluate(critic_obs).detach()
        t.actions_log_prob = ac.get_actions_log_prob(t.actions).detach()
        t.action_mean, t.action_sigma = ac.action_mean.detach(), ac.action_std.detach()
        t.observations, t.critic_observations = obs, critic_obs
        # record everything that env.step() is about to overwrite or that belongs to this step
        st.observations[s].copy_(obs)
        if st.privileged_observations is not None:
            st.privileged_observations[s].copy_(critic_obs)
        st.actions[s].copy_(t.actions)
        st.values[s].copy_(t.values)
        st.actions_log_prob[s].copy_(t.actions_log_prob.view(-1, 1))
        st.mu[s].copy_(t.action_mean)
        st.sigma[s].copy_(t.action_sigma)
        return t.actions

    def _policy_kernel(self):
        """-> the fused policy kernel for this actor-critic, or None (not asked for / modules it does not cover).  On a GPU the library must
        be there: a missing libgo2nn_hip.so raises (no silent fallback to the slower path)."""
        if self._pk is not None:
            return self._pk if self._pk is not False else None
        lib = self.nn_lib
        if lib is None and self._pk_on:
            from ... import _nn
            lib = _nn.load_nn()
        ok = False
        if lib is not None:
            from ... import _nn
            ok = _nn.PolicyKernel.supports(self.actor_critic) and self.storage is not None and self.storage.privileged_observations is not None
        self._pk = _nn.PolicyKernel(lib, self.actor_critic) if ok else False
        return self._pk if self._pk is not False else None

    def process_env_step(self, rewards, dones, infos):
        st, t = self.storage, self.transition
        s = st.step
        if self.fused_rollout:
            self._store_transition(rewards, dones, infos, s)
            st.step += 1
            t.clear()
            self.actor_critic.reset(dones)
            return
        r = rewards.clone()
        if "time_outs" in infos:   # bootstrap on time-outs (ppo.py:107-108)
            r += self.gamma * torch.squeeze(st.values[s] * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        st.rewards[s].copy_(r.view(-1, 1))
        st.dones[s].copy_(dones.view(-1, 1))
        st.step += 1
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        pk = self._pk if self._pk not in (None, False) else None
        if pk is not None and self._pk_packed and last_critic_obs.is_contiguous() and last_critic_obs.dtype == torch.float32:
            last_values = pk.critic.forward(last_critic_obs)          # the bootstrap value as ONE launch on the weights packed for this rollout (they have not changed since)
        else:
            last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    # ------------------------------------------------------------------ update half (ppo.py:120-187)
    def _losses(self, obs_b, cobs_b, act_b, tv_b, adv_b, ret_b, old_lp_b, old_mu_b, old_sig_b):
        ac = self.actor_critic
        if self.fused_loss:
            mu_b, val_b = self._actor_critic(ac, obs_b, cobs_b)
            loss, stats = _FusedPPOLoss.apply(mu_b, ac.std, val_b, self, act_b, tv_b, adv_b, ret_b, old_lp_b, old_mu_b, old_sig_b)
            return loss, stats[1], stats[0], stats[2]
        ac.update_distribution(obs_b)     # the reference calls act() here and discards the sample (ppo.py:131)
        lp_b = ac.get_actions_log_prob(act_b)
        val_b = ac.evaluate(cobs_b)
        mu_b, sig_b, ent_b = ac.action_mean, ac.action_std, ac.entropy
        with torch.no_grad():
            kl = torch.sum(torch.log(sig_b / old_sig_b + 1.0e-5) + (torch.square(old_sig_b) + torch.square(old_mu_b - mu_b)) / (2.0 * torch.square(sig_b)) - 0.5, axis=-1)
            kl_mean = torch.mean(kl)
        ratio = torch.exp(lp_b - torch.squeeze(old_lp_b))
        sur = -torch.squeeze(adv_b) * ratio
        sur_clip = -torch.squeeze(adv_b) * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)
        surrogate_loss = torch.max(sur, sur_clip).mean()
        if self.use_clipped_value_loss:
            v_clip = tv_b + (val_b - tv_b).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((val_b - ret_b).pow(2), (v_clip - ret_b).pow(2)).mean()
        else:
            value_loss = (ret_b - val_b).pow(2).mean()
        loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * ent_b.mean()
        return loss, value_loss, surrogate_loss, kl_mean

    def _allreduce_grads(self, world, kl_mean=None):
        """Average the gradients — and the mean KL riding in the same bucket — over the shards with ONE all-reduce (1.96 MB of
        fp32, latency-bound on xGMI): the learning-rate decision only matters at optimizer.step(), so it can wait for the
        backward pass and share its collective.  -> the shard-averaged KL (or None)."""
        return allreduce_mean_bucket([p.grad for p in self.actor_critic.parameters() if p.grad is not None], world, kl_mean)

    def _update_eager(self):
        mean_value_loss, mean_surrogate_loss = 0.0, 0.0
        world = _world()
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        for batch in self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
            loss, value_loss, surrogate_loss, kl_mean = self._losses(*batch[:9])
            self.optimizer.zero_grad()
            loss.backward()
            if _collectives_on():
                kl_mean = self._allreduce_grads(world, kl_mean if adaptive else None)
            if adaptive:        # the reference decides before backward (ppo.py:140-155); the rate is only read by optimizer.step()
                if kl_mean > self.desired_kl * 2.0:
                    self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                elif kl_mean < self.desired_kl / 2.0 and kl_mean > 0.0:
                    self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                for g in self.optimizer.param_groups:
                    if torch.is_tensor(g["lr"]):
                        g["lr"].fill_(self.learning_rate)
                    else:
                        g["lr"] = self.learning_rate
            nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
            self.optimizer.step()
            mean_value_loss += value_loss.item()
            mean_surrogate_loss += surrogate_loss.item()
        n = self.num_learning_epochs * self.num_mini_batches
        return mean_value_loss / n, mean_surrogate_loss / n

    # ---- graph mode -------------------------------------------------------------------------------------------
    _KEYS = ("obs", "cobs", "act", "val", "adv", "ret", "logp", "mu", "sig")

    def _graph_front(self, i, split=False):
        """Forward, losses, backward of mini-batch i with every decision on the device (same arithmetic as _update_eager).  Mini-batch i
        is the i-th contiguous chunk of the rollout permuted ONCE per update (the reference reuses one permutation for all epochs,
        rollout_storage.py:150): 4 chunk gathers per iteration instead of 20 mini-batch gathers.  split: the gradients and the mean
        KL are packed into the all-reduce bucket (more than one rank)."""
        mb = self._mb
        batch = [self._perm[k][i * mb:(i + 1) * mb] for k in self._KEYS]
        if self.fused_loss and type(self.actor_critic) is _AC and self._heads_path(batch):
            # a plain ActorCritic: forward, loss and backward of the mini-batch as explicit launches, no autograd graph (modules/fused.py:ppo_pair_grads):
            # grouped hidden layers, go2nn_ppo_heads (heads forward + loss + heads backward in one pass), ONE go2nn_sum_rows; .grad of every parameter is set
            from ..modules import fused
            ac = self.actor_critic
            self.optimizer.zero_grad(set_to_none=True)
            stats = fused.ppo_pair_grads(ac, batch[0], batch[1], batch[2], batch[3], batch[4], batch[5], batch[6], batch[7], batch[8], self.clip_param, self.value_loss_coef,
                                         self.entropy_coef, self.use_clipped_value_loss, acc=self._acc)          # (the running loss sums: added by the pass's go2nn_sum_rows launch)
            kl_mean = stats[2]
        elif self.fused_loss:
            # the loss kernel already holds d loss / d (mu, std, value): seed the backward pass of the two networks with them directly
            # (loss.backward() through the autograd.Function costs a clone and three multiplications by the unit upstream gradient)
            ac = self.actor_critic
            mu_b, val_b = self._actor_critic(ac, batch[0], batch[1])
            stats, gmu, gstd, gval = _FusedPPOLoss.kernel(self, mu_b, ac.std, val_b, *batch[2:])
            self.optimizer.zero_grad(set_to_none=True)
            self._acc.add_(stats[:2])             # [surrogate, value loss]; read back swapped in _update_graphs (issued before the backward pass: off its tail)
            if ac.std.requires_grad and ac.std.grad_fn is None:
                ac.std.grad = gstd.view_as(ac.std)             # a leaf: the kernel's gradient IS its .grad (autograd would copy it there with one more launch)
                torch.autograd.backward([mu_b, val_b], [gmu, gval])
            else:
                torch.autograd.backward([mu_b, ac.std, val_b], [gmu, gstd.view_as(ac.std), gval])
            kl_mean = stats[2]
        else:
            loss, value_loss, surrogate_loss, kl_mean = self._losses(*batch)
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            self._acc.add_(torch.stack([surrogate_loss.detach(), value_loss.detach()]))
        if split:
            if self._bucket is None:
                self._bucket = GradBucket(list(self.actor_critic.parameters()), 1 if self._adaptive() else 0)
            self._bucket.pack(kl_mean)
        else:
            self._kl = kl_mean

    def _adaptive(self):
        return self.desired_kl is not None and self.schedule == "adaptive"

    def _heads_path(self, batch):
        if getattr(self, "surrogate_split", 0):
            return False
        from ..modules import fused
        return fused.ppo_pair_applicable(self.actor_critic, batch[0], batch[1])

    def _graph_back(self, split=False):
        """LR decision, gradient clipping, Adam.  split: on the all-reduced bucket (shard-mean gradients and KL: every rank takes the
        same LR branch); otherwise on this rank's gradients."""
        if split:
            kl_mean = self._bucket.unpack(_world())
        else:
            kl_mean = self._kl
        if self._fused_adam is None:
            self._fused_adam = FusedClipAdam(self.lib, self.optimizer, self.actor_critic.parameters(), self.max_grad_norm)
        if self._fused_adam.usable and self._fused_adam.step(kl_mean if self._adaptive() else None, self.desired_kl if self._adaptive() else 0.0):
            return
        if self._adaptive():
            lr = self._lr_t
            up = torch.clamp(lr * 1.5, max=1e-2)
            down = torch.clamp(lr / 1.5, min=1e-5)
            kl_mean = kl_mean.reshape(())
            new_lr = torch.where(kl_mean > self.desired_kl * 2.0, down, torch.where((kl_mean < self.desired_kl / 2.0) & (kl_mean > 0.0), up, lr))
            lr.copy_(new_lr)
        nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm, foreach=True)
        self.optimizer.step()

    def _graph_step(self, i):
        self._graph_front(i)
        self._graph_back()

    def _update_graphs(self):
        st = self.storage
        nmb, mb = self.num_mini_batches, (st.num_envs * st.num_transitions_per_env) // self.num_mini_batches
        if self._graph is None:
            flat = lambda t: t.flatten(0, 1)
            self._flat = {"obs": flat(st.observations), "cobs": flat(st.privileged_observations) if st.privileged_observations is not None else flat(st.observations),
                          "act": flat(st.actions), "val": flat(st.values), "ret": flat(st.returns), "logp": flat(st.actions_log_prob), "adv": flat(st.advantages),
                          "mu": flat(st.mu), "sig": flat(st.sigma)}
            self._mb = mb
            self._perm = {k: torch.empty((nmb * mb,) + tuple(v.shape[1:]), device=self.device, dtype=v.dtype) for k, v in self._flat.items()}
            self._acc = torch.zeros(2, device=self.device)
            # one captured step per mini-batch slot (each reads its own chunk of the permuted rollout).  Slot 0 runs 3 eager steps on
            # a side stream first (allocator / lazy initialisation settle; they are real PPO steps of the first update), the others
            # one; then each is captured once and replayed.  A failed capture degrades that slot to eager execution.
            # More than one rank: two captured halves per slot with the gradient all-reduce eager between them (_graph.py).
            self._bucket = None
            if _collectives_on():
                self._graph = [ReducedStep((lambda i=i: self._graph_front(i, True)), (lambda: self._graph_back(True)), (lambda: self._bucket),
                                           enabled=self._capture, warmup=3 if i == 0 else 1, name="PPO mini-batch step %d" % i) for i in range(nmb)]
            else:
                # one rank: the whole UPDATE (every epoch's nmb mini-batch steps, each on its own chunk of the permuted rollout) is one graph — a graph launch costs ≈ 9 µs of idle
                # chip between two mini-batches (profiles/r4_timeline…: Adam step -> next split): round 4 went from 20 launches per update to 5 (one per epoch: 15.85 -> 15.76 ms per
                # iteration), round 5 to one.  The first update runs eagerly.
                ne = self.num_learning_epochs
                self._graph = [CapturedStep((lambda: [self._graph_step(i) for _ in range(ne) for i in range(nmb)] and None), enabled=self._capture, warmup=1,
                                            name="PPO update (%d epochs x %d mini-batch steps)" % (ne, nmb))]
            self._graph_reps = self.num_learning_epochs if _collectives_on() else 1
            # ONE permutation for the whole update, reused by every epoch, as in the reference (rollout_storage.py:150), and the nine storage tensors gathered
            # into mini-batch order: ONE launch (go2sim_shuffle_gather: a keyed sort-free shuffle computed per output row + all gathers + the loss accumulators'
            # reset) instead of torch.randperm's 12-kernel radix sort, 9 index_select launches and the fills between them (~0.36 ms of launch chain per update)
            import ctypes as C
            from ..._abi import Go2GatherJob
            keys = [k for k in self._KEYS]
            use_lib = self.lib is not None and hasattr(self.lib, "go2sim_shuffle_gather") and all(self._flat[k].dtype == torch.float32 and self._flat[k].is_contiguous() for k in keys)
            if use_lib:
                row = lambda t: int(t[0].numel()) if t.dim() > 1 else 1
                self._gather_jobs = (Go2GatherJob * len(keys))(*[Go2GatherJob(self._flat[k].data_ptr(), self._perm[k].data_ptr(), row(self._flat[k]), 0) for k in keys])
                self._make_shuffle_key()
            rows = nmb * mb

            def permute():
                if not use_lib:
                    self._acc.zero_()
                    indices = torch.randperm(rows, requires_grad=False, device=self.device)
                    for k in self._KEYS:
                        torch.index_select(self._flat[k], 0, indices, out=self._perm[k])
                    return
                stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if str(self.device).startswith("cuda") else None
                idx = None
                if torch.randperm is not _RANDPERM:
                    idx = torch.randperm(rows, requires_grad=False, device=self.device).to(torch.int64).contiguous()
                rc = self.lib.go2sim_shuffle_gather(self._gather_jobs, len(keys), rows, C.c_void_p(idx.data_ptr()) if idx is not None else None,
                                                    C.c_void_p(self._shuffle_key.data_ptr()), C.c_void_p(self._acc.data_ptr()), int(self._acc.numel()), stream)
                if rc != 0:
                    raise RuntimeError("go2sim_shuffle_gather failed: %s" % self.lib.go2sim_last_error().decode())
            self._permute = CapturedStep(permute, enabled=self._capture, warmup=2, name="PPO rollout permutation", optional=True)
        self._permute()
        for _ in range(self._graph_reps):
            for g in self._graph:
                g()
        n = self.num_learning_epochs * nmb
        out = torch.cat([self._acc / n, self._lr_t.reshape(1)]).tolist()          # ONE device -> host read per update
        self.learning_rate = float(out[2])
        return out[1], out[0]

    def graphs_captured(self):
        """True iff every mini-batch step of the update is being replayed from a HIP graph (bench.py reports it and refuses to quote a
        number from a silently degraded run)."""
        # (the permutation step is a CapturedStep too — `optional`, a failed capture keeps it eager even under GO2_STRICT_GRAPHS — and counts here: a run
        #  whose update replays but whose permutation fell back to ten eager launches does not report "update": true)
        head = getattr(self, "_permute", None)
        return bool(self.use_graphs and self._capture and all_captured(self._graph) and (head is None or head.calls == 0 or head.graph is not None))

    def update(self):
        self._pk_packed = False          # the optimizer steps below change the parameters: the next rollout re-packs
        if self.use_graphs:
            out = self._update_graphs()
        else:
            out = self._update_eager()
        self.storage.clear()
        return out
