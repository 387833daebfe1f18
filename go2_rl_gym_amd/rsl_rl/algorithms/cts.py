"""Concurrent Teacher-Student PPO (rsl_rl/rsl_rl/algorithms/cts.py:39-286; CTS, arXiv 2405.10830).

75 % of the envs act through the teacher encoder (privileged obs -> latent), 25 % through the student encoder (5-frame
history -> latent); both share the actor and the critic.  optimizer1 (teacher encoder, critic, actor, std) takes the PPO
loss with   surrogate = mean(teacher rows) + mean(student rows)   (:228-231); optimizer2 trains the student encoder to
reproduce the teacher's latent on the student rows (:259-275), after the policy epochs, on the same mini-batches.

Re-designed for the GPU: the rollout stays in env order (storage/rollout_storage_cts.py); each group's latent is computed
once and ONE actor pass and ONE critic pass serve all rows, in the rollout and in the update; the loss head is the fused
library kernel (go2sim_ppo_loss, surrogate_split = teacher rows); the policy step and the student step are each captured
into a HIP graph and replayed 20x per iteration with the learning rate in a device tensor.  The eager mode keeps the
reference's control flow and is what tests/test_cts_golden.py pins against the reference's own update."""
import itertools
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim

from ..storage import RolloutStorageCTS
from ._graph import CapturedStep, FusedClipAdam, GradBucket, ReducedStep, all_captured
from .ppo import _ADAM_IMPL, _RANDPERM, _collectives_on, _FusedPPOLoss, _RolloutHeads, _world, allreduce_mean_bucket


def _allreduce_mean_grads(params, world, extra=None):
    return allreduce_mean_bucket([p.grad for p in params if p.grad is not None], world, extra)


class CTS(_RolloutHeads):
    def __init__(self, model, num_envs, history_length, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, student_encoder_learning_rate=1e-3, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01, teacher_env_ratio=0.75, device="cpu", lib=None,
                 use_graphs=None, fused_loss=None, fused_rollout=None):
        self.device, self.lib = device, lib
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.history_length = history_length
        self.model = model
        self.model.to(self.device)
        self.storage = None
        on_gpu = str(device).startswith("cuda")
        self.use_graphs = on_gpu if use_graphs is None else bool(use_graphs and on_gpu)
        self._capture = self.use_graphs and use_graphs != "uncaptured"
        if use_graphs == "uncaptured":     # the graph-mode update run eagerly on any device (CPU tests), see PPO
            self.use_graphs = True
        self.fused_loss = (on_gpu and lib is not None) if fused_loss is None else bool(fused_loss and lib is not None)
        self.fused_rollout = (on_gpu and lib is not None) if fused_rollout is None else bool(fused_rollout and lib is not None)
        if getattr(model, "state_dependent_std", False):       # MCP actor: the fused heads assume one std per action dimension
            self.fused_loss = self.fused_rollout = False
        if on_gpu and lib is not None and os.environ.get("GO2_FUSED_MLP", "1") == "1":
            from ..modules import fused
            fused.set_library(lib)
        groups1 = [{"params": g} for g in self.model.policy_parameter_groups()]                           # same 4 groups as the reference (:72-77)
        self._params1 = list(itertools.chain.from_iterable(g["params"] for g in groups1))
        self._params2 = list(self.model.student_parameters())
        if self.use_graphs:
            self._lr_t = torch.tensor(float(learning_rate), device=device)
            kw = dict(capturable=True, **_ADAM_IMPL) if self._capture else dict(foreach=False)
            self.optimizer1 = optim.Adam(groups1, lr=self._lr_t, **kw)
            self.optimizer2 = optim.Adam(self._params2, lr=torch.tensor(float(student_encoder_learning_rate), device=device), **kw)
        else:
            self._lr_t = None
            self.optimizer1 = optim.Adam(groups1, lr=learning_rate)
            self.optimizer2 = optim.Adam(self._params2, lr=student_encoder_learning_rate)
        self.transition = RolloutStorageCTS.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        # teacher / student split (:90-101): with ratio 0.75 every 4th env is a student
        self.teacher_num_envs = max(int(num_envs * teacher_env_ratio), 1)
        self.student_num_envs = num_envs - self.teacher_num_envs
        every = int(1 / (1 - teacher_env_ratio))
        ids = torch.arange(num_envs, device=self.device)
        self.teacher_env_idxs, self.student_env_idxs = ids[ids % every != 0], ids[ids % every == 0]
        assert len(self.teacher_env_idxs) == self.teacher_num_envs, f"{len(self.teacher_env_idxs)=} != {self.teacher_num_envs=}"
        assert len(self.student_env_idxs) == self.student_num_envs, f"{len(self.student_env_idxs)=} != {self.student_num_envs=}"
        self.surrogate_split = 0            # set per update: teacher rows of a mini-batch (read by _FusedPPOLoss)
        self._steps = None
        self._step_reps = None
        self._plan = False                  # the no-autograd mini-batch (modules/fused_cts.py): decided at the first graph-mode update (None: not applicable)
        self._fused_adam1 = self._fused_adam2 = None
        if _world() > 1:
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)

    # the runner reaches the networks through either name
    @property
    def actor_critic(self):
        return self.model

    def rebind_lr(self):
        """See PPO.rebind_lr; optimizer2's rate is fixed but must be a device tensor again for the captured step."""
        self.learning_rate = float(self.optimizer1.param_groups[0]["lr"])
        if self._lr_t is not None:
            self._lr_t.fill_(self.learning_rate)
            for g in self.optimizer1.param_groups:
                g["lr"] = self._lr_t
            for g in self.optimizer2.param_groups:
                if not torch.is_tensor(g["lr"]):
                    g["lr"] = torch.tensor(float(g["lr"]), device=self.device)

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorageCTS(num_envs, self.teacher_env_idxs, self.student_env_idxs, self.history_length, num_transitions_per_env,
                                         actor_obs_shape, critic_obs_shape, action_shape, self.device, lib=self.lib)

    def test_mode(self):
        self.model.eval()

    def train_mode(self):
        self.model.train()

    # ------------------------------------------------------------------ rollout half (:112-166), env order
    def _latent_env_order(self, privileged_obs, history):
        m, ti, si = self.model, self.teacher_env_idxs, self.student_env_idxs
        # the two encoders are independent: on two HIP streams in the captured rollout (they are launch-bound chains of small kernels)
        lt, ls = self._pair(lambda: m.teacher_encoder(privileged_obs[ti]), lambda: m.student_latent(history[si])[0], enabled=self._capture)
        latent = torch.empty(privileged_obs.shape[0], lt.shape[1], device=lt.device, dtype=lt.dtype)
        latent.index_copy_(0, ti, lt.detach())
        latent.index_copy_(0, si, ls.detach())
        return latent

    def act(self, obs, privileged_obs, history):
        st, t, m = self.storage, self.transition, self.model
        s = st.step
        if s >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        # record what env.step() is about to overwrite (the env's buffers and the runner's history ring are updated in place)
        if obs.data_ptr() != st.observations[s].data_ptr():                     # (the env wrote this row itself: LeggedRobot.step(rollout=...))
            st.observations[s].copy_(obs)
        if privileged_obs.data_ptr() != st.privileged_observations[s].data_ptr():
            st.privileged_observations[s].copy_(privileged_obs)
        st.history[s].copy_(history)
        pk = self._policy_kernel() if self.fused_rollout else None
        if pk is not None and all(x.is_contiguous() and x.dtype == torch.float32 for x in (obs, privileged_obs, history)):
            # two launches (include/go2nn.h ABI 5): both encoders on their env subsets -> the env-ordered latent; actor + critic + sampling head on [latent | obs] / [latent | priv]
            if s == 0 or not self._pk_packed:
                self._img_cache = {}
                pk.pack()                  # the parameters only change in update(): once per rollout (inside the captured rollout graph too)
                self._pk_packed = True
                self._pk_recorded = self._pk_recorded or s == 0
            latent = self._rollout_latent(pk, privileged_obs, history)
            actions = pk.act(latent, obs, privileged_obs, self._rollout_noise(m, st, s), st.actions[s], st.mu[s], st.sigma[s], st.actions_log_prob[s].view(-1), st.values[s].view(-1))
            t.actions, t.values, t.actions_log_prob = actions, st.values[s], st.actions_log_prob[s].view(-1)
            t.action_mean, t.action_sigma = st.mu[s], st.sigma[s]
            return actions
        if s == 0:
            self._pk_recorded = False
        latent = self._latent_env_order(privileged_obs, history)
        if self.fused_rollout:
            mu, value = self._pair(lambda: m.policy_mean(latent, obs), lambda: m.evaluate_joint(privileged_obs, latent, obs), enabled=self._capture and not m.heads_share_parameters)
            return self._act_head(mu, m.std, m._noise(mu), value, s)      # (MCP-CTS: state-dependent std, no storage-shaped stand-in for the noise)
        t.actions = m.act_joint(obs, latent).detach()
        t.values = m.evaluate_joint(privileged_obs, latent, obs).detach()
        t.actions_log_prob = m.get_actions_log_prob(t.actions).detach()
        t.action_mean, t.action_sigma = m.action_mean.detach(), m.action_std.detach()
        st.actions[s].copy_(t.actions)
        st.values[s].copy_(t.values)
        st.actions_log_prob[s].copy_(t.actions_log_prob.view(-1, 1))
        st.mu[s].copy_(t.action_mean)
        st.sigma[s].copy_(t.action_sigma)
        return t.actions

    def process_env_step(self, rewards, dones, infos):
        st = self.storage
        s = st.step
        if self.fused_rollout:
            self._store_transition(rewards, dones, infos, s)
        else:
            r = rewards.clone()
            if "time_outs" in infos:   # bootstrap on time-outs (:156-158)
                r += self.gamma * torch.squeeze(st.values[s] * infos["time_outs"].unsqueeze(1).to(self.device), 1)
            st.rewards[s].copy_(r.view(-1, 1))
            st.dones[s].copy_(dones.view(-1, 1))
        st.step += 1
        self.transition.clear()
        # model.reset(dones) (:163): the deployment-side history inside the module (written by act_inference only) — all zeros throughout training, where zeroing
        # rows of it is a no-op: two launches per env step saved
        if getattr(self.model, "_history_dirty", True):
            self.model.history.masked_fill_(dones.view(-1, 1, 1) > 0, 0.0)          # without a boolean-index sync

    def compute_returns(self, last_privileged_obs, last_history, last_obs=None):
        pk = self._pk if self._pk not in (None, False) else None
        if pk is not None and self._pk_packed and all(x.is_contiguous() and x.dtype == torch.float32 for x in (last_privileged_obs, last_history)):
            last_values = pk.value(self._rollout_latent(pk, last_privileged_obs, last_history), last_privileged_obs)      # on the weights packed for this rollout (unchanged since)
        else:
            latent = self._latent_env_order(last_privileged_obs, last_history)
            last_values = self.model.evaluate_joint(last_privileged_obs, latent, last_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    _pk = None

    def _policy_kernel(self):
        """-> _nn.PolicyKernelCTS for this model, or None (GO2_FUSED_POLICY=0, or heads / encoders the kernel does not cover: modules/fused_cts.py:cts_plan).  On a
        GPU the library must be there (fused.set_library raises otherwise); nn_lib: tests hand in the host build."""
        if self._pk is None:
            self._pk = False
            nn_lib = getattr(self, "nn_lib", None)
            on = nn_lib is not None or (str(self.device).startswith("cuda") and self.lib is not None and os.environ.get("GO2_FUSED_POLICY", "1") == "1")
            if on and self.storage is not None and self.storage.privileged_observations is not None:
                from ..modules import fused, fused_cts
                from ... import _nn
                lib = nn_lib if nn_lib is not None else (fused._NN if fused._NN is not None else _nn.load_nn())
                saved = fused._NN
                fused._NN = lib
                try:
                    plan = fused_cts.cts_plan(self.model)
                finally:
                    fused._NN = saved
                try:
                    if plan is not None:
                        self._pk = _nn.PolicyKernelCTS(lib, self.model, plan, self.teacher_env_idxs, self.student_env_idxs)
                        self._latent_buf = torch.zeros(self.storage.num_envs, plan.L, device=self.device)
                except ValueError:          # (a width the kernel's LDS tiles do not hold)
                    self._pk = False
        return self._pk if self._pk is not False else None

    def _rollout_latent(self, pk, privileged_obs, history):
        """-> the env-ordered latent [N, L]: one launch for both encoders; a student encoder that is not a plain MLP (MoE) runs as torch modules, without gradient, on the same stream"""
        latent = self._latent_buf
        if pk.enc_s is None:
            from ..modules.fused import own_forward
            if self._img_cache is None:
                self._img_cache = {}
            with torch.no_grad(), own_forward(images=self._img_cache):          # (the split weight images: once per rollout — act() empties the cache at step 0)
                hs = history[self.student_env_idxs]
                parts = getattr(self.model, "student_moe_parts", None)
                if parts is not None and pk.moe_mix_ok(self.model):
                    # the soft mixture's tail (softmax gate, weighted sum + bias, normaliser, scatter to the env-ordered latent) as ONE launch (include/go2nn.h ABI 6)
                    logits, outs, bias = parts(hs)
                    pk.moe_mix(logits, outs, bias, latent)
                else:
                    latent.index_copy_(0, self.student_env_idxs, self.model.student_latent(hs)[0])
        pk.latents(privileged_obs, history, latent)
        return latent

    # ------------------------------------------------------------------ update half (:167-286)
    def _policy_losses(self, obs_b, priv_b, hist_b, act_b, tv_b, adv_b, ret_b, old_lp_b, old_mu_b, old_sig_b, n_t):
        """-> loss, value_loss, surrogate_loss, entropy_mean, kl_mean   (rows [0,n_t) teacher, the rest student)"""
        m = self.model
        latent = m.latents(priv_b, hist_b, n_t, pair=(lambda f, g: self._pair(f, g, enabled=self._capture)))
        if self.fused_loss:
            mu_b, (val_b, aux) = self._pair(lambda: m.policy_mean(latent, obs_b), lambda: m.value(latent, obs_b, priv_b), enabled=self._capture and not m.heads_share_parameters)
            self.surrogate_split = n_t
            loss, stats = _FusedPPOLoss.apply(mu_b, m.std, val_b, self, act_b, tv_b, adv_b, ret_b, old_lp_b, old_mu_b, old_sig_b)
            return self._policy_extra(loss, aux), stats[1], stats[0], stats[3], stats[2]
        m.update_distribution(torch.cat([latent, obs_b], dim=1))
        lp_b = m.get_actions_log_prob(act_b)
        val_b, aux = m.value(latent, obs_b, priv_b)
        mu_b, sig_b, ent_b = m.action_mean, m.action_std, m.entropy
        with torch.no_grad():
            kl = torch.sum(torch.log(sig_b / old_sig_b + 1.0e-5) + (torch.square(old_sig_b) + torch.square(old_mu_b - mu_b)) / (2.0 * torch.square(sig_b)) - 0.5, axis=-1)
            kl_mean = torch.mean(kl)
        ratio = torch.exp(lp_b - torch.squeeze(old_lp_b))
        sur = -torch.squeeze(adv_b) * ratio
        sur_clip = -torch.squeeze(adv_b) * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)
        sl = torch.max(sur, sur_clip)
        surrogate_loss = sl[:n_t].mean() + sl[n_t:].mean()
        if self.use_clipped_value_loss:
            v_clip = tv_b + (val_b - tv_b).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((val_b - ret_b).pow(2), (v_clip - ret_b).pow(2)).mean()
        else:
            value_loss = (ret_b - val_b).pow(2).mean()
        ent = ent_b.mean()
        loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * ent
        return self._policy_extra(loss, aux), value_loss, surrogate_loss, ent, kl_mean

    _NUM_POLICY_LOGS = 0
    _policy_logs = ()           # 0-d tensors logged by the last _policy_extra call (the actor load-balance term of the AC-MoE variants)

    def _policy_extra(self, loss, aux):
        """Hook: extra terms of the policy loss from the value head's auxiliary output (the actor gate weights of the AC-MoE variants)."""
        return loss

    def _student_losses(self, hist_s, priv_s, teacher_latent=None):
        """-> total loss, (latent_loss, ...) for the log    (student rows only).  teacher_latent: the teacher encoder's latent of these rows when the caller holds it
        (graph mode computes it once per update: optimizer2 never touches the teacher encoder, so it is a constant of the student epochs)"""
        student_latent, _ = self.model.student_latent(hist_s)
        teacher_latent = self._teacher_latent(priv_s, teacher_latent)
        latent_loss = (teacher_latent - student_latent).pow(2).mean()
        return latent_loss, (latent_loss,)

    def _teacher_latent(self, priv_s, given=None):
        if given is not None:
            return given
        with torch.no_grad():
            return self.model.teacher_encoder(priv_s)

    _NUM_STUDENT_LOGS = 1

    def _gather(self, fl, b):
        return tuple(fl[k][b] for k in ("obs", "cobs", "hist", "act", "val", "adv", "ret", "logp", "mu", "sig"))

    def _teacher_rows(self):
        return self.teacher_num_envs * self.storage.num_transitions_per_env // self.num_mini_batches

    def _update_eager(self):
        fl, n_t = self.storage.flat(), self._teacher_rows()
        idx = self.storage.mini_batch_indices(self.num_mini_batches)
        world, sync = _world(), _collectives_on()
        P = self._NUM_POLICY_LOGS
        acc = [0.0] * (3 + P + self._NUM_STUDENT_LOGS)
        for _ in range(self.num_learning_epochs):
            for b in idx:
                loss, value_loss, surrogate_loss, ent, kl_mean = self._policy_losses(*self._gather(fl, b), n_t)
                adaptive = self.desired_kl is not None and self.schedule == "adaptive"
                self.optimizer1.zero_grad()
                loss.backward()
                if sync:
                    kl_mean = _allreduce_mean_grads(self._params1, world, kl_mean if adaptive else None)
                if adaptive:     # decided after backward (the rate is only read by optimizer1.step()) so the KL shares the gradient all-reduce
                    if kl_mean > self.desired_kl * 2.0:
                        self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                    elif kl_mean < self.desired_kl / 2.0 and kl_mean > 0.0:
                        self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                    for g in self.optimizer1.param_groups:
                        if torch.is_tensor(g["lr"]):
                            g["lr"].fill_(self.learning_rate)
                        else:
                            g["lr"] = self.learning_rate
                nn.utils.clip_grad_norm_(self._params1, self.max_grad_norm)
                self.optimizer1.step()
                acc[0] += value_loss.item(); acc[1] += surrogate_loss.item(); acc[2] += ent.item()
                for i, v in enumerate(self._policy_logs):
                    acc[3 + i] += v.item()
        for _ in range(self.num_learning_epochs):
            for b in idx:
                bs = b[n_t:]
                loss, logs = self._student_losses(fl["hist"][bs], fl["cobs"][bs])
                self.optimizer2.zero_grad()
                loss.backward()
                if sync:
                    _allreduce_mean_grads(self._params2, world)
                nn.utils.clip_grad_norm_(self._params2, self.max_grad_norm)
                self.optimizer2.step()
                for i, v in enumerate(logs):
                    acc[3 + P + i] += v.item()
        n = self.num_learning_epochs * self.num_mini_batches
        return self._ordered(tuple(a / n for a in acc))

    def _ordered(self, acc):
        """accumulators [value, surrogate, entropy | policy logs | student logs] -> the reference's return order: ..., student logs, policy logs"""
        P = self._NUM_POLICY_LOGS
        return acc[:3] + acc[3 + P:] + acc[3:3 + P]

    # ---- graph mode: every decision on the device; one captured policy step and one captured student step per mini-batch slot ----
    _KEYS = ("obs", "cobs", "hist", "act", "val", "adv", "ret", "logp", "mu", "sig")

    def _adaptive(self):
        return self.desired_kl is not None and self.schedule == "adaptive"

    def _own_plan(self):
        """The CTS mini-batch as explicit launches without autograd (modules/fused_cts.py), when it covers this model and algorithm — the plain CTS heads and loss
        (MoE-CTS included: only its student step differs) on the library pair; None keeps the autograd formulation (GO2_FUSED_MLP=0 forces that for A/B runs: the
        library pair is then never handed to modules/fused.py)."""
        if self._plan is False:
            self._plan = None
            if self.fused_loss and self.lib is not None and type(self)._policy_extra is CTS._policy_extra:
                from ..modules import fused_cts
                self._plan = fused_cts.cts_plan(self.model)
        return self._plan

    def _own_student(self):
        plan = self._own_plan()
        return plan is not None and plan.student is not None and type(self)._student_losses is CTS._student_losses

    def _policy_front(self, i, split=False):
        mb = self._mb
        if self._own_plan() is not None:
            from ..modules import fused_cts
            P, n_t, s = self._perm, self._teacher_rows(), slice(i * mb, (i + 1) * mb)
            self.optimizer1.zero_grad(set_to_none=True)
            stats = fused_cts.cts_policy_grads(self._plan, self.model, P["ain"][s], P["cin"][s], P["cobs"][i * mb:i * mb + n_t],
                                               tuple(P[k][s] for k in ("act", "val", "adv", "ret", "logp", "mu", "sig")), n_t, self.clip_param, self.value_loss_coef,
                                               self.entropy_coef, self.use_clipped_value_loss, acc=self._acc_own)          # (the running sums: added by the pass's go2nn_sum_rows launch)
            kl_mean = stats[2]
        else:
            loss, value_loss, surrogate_loss, ent, kl_mean = self._policy_losses(*(self._perm[k][i * mb:(i + 1) * mb] for k in self._KEYS), self._teacher_rows())
            self.optimizer1.zero_grad(set_to_none=True)
            loss.backward()
            self._acc[:3 + self._NUM_POLICY_LOGS].add_(torch.stack([value_loss.detach(), surrogate_loss.detach(), ent.detach()] + [v.detach() for v in self._policy_logs]))
        if split:      # more than one rank: gradients + mean KL into the all-reduce bucket (_graph.py)
            if self._bucket1 is None:
                self._bucket1 = GradBucket(self._params1, 1 if self._adaptive() else 0)
            self._bucket1.pack(kl_mean)
        else:
            self._kl = kl_mean

    def _policy_back(self, split=False):
        if split:
            kl_mean = self._bucket1.unpack(_world())
        else:
            kl_mean = self._kl
        if self._fused_adam1 is None:
            self._fused_adam1 = FusedClipAdam(self.lib, self.optimizer1, self._params1, self.max_grad_norm)
        if self._fused_adam1.usable and self._fused_adam1.step(kl_mean if self._adaptive() else None, self.desired_kl if self._adaptive() else 0.0):
            return
        if self._adaptive():
            lr = self._lr_t
            kl_mean = kl_mean.reshape(())
            up, down = torch.clamp(lr * 1.5, max=1e-2), torch.clamp(lr / 1.5, min=1e-5)
            lr.copy_(torch.where(kl_mean > self.desired_kl * 2.0, down, torch.where((kl_mean < self.desired_kl / 2.0) & (kl_mean > 0.0), up, lr)))
        nn.utils.clip_grad_norm_(self._params1, self.max_grad_norm, foreach=True)
        self.optimizer1.step()

    def _policy_step(self, i):
        self._policy_front(i)
        self._policy_back()

    def _student_front(self, i, split=False):
        mb, n_t = self._mb, self._teacher_rows()
        hist_s, priv_s = self._perm["hist"][i * mb + n_t:(i + 1) * mb], self._perm["cobs"][i * mb + n_t:(i + 1) * mb]
        self.optimizer2.zero_grad(set_to_none=True)
        if self._own_student():
            from ..modules import fused_cts
            fused_cts.cts_student_grads(self._plan, self.model, hist_s, priv_s, acc=self._acc_own[4:])
        else:
            self._student_backward(hist_s, priv_s, self._tlat[i] if self._tlat is not None else None)
        if split:
            if self._bucket2 is None:
                self._bucket2 = GradBucket(self._params2)
            self._bucket2.pack()

    def _student_back(self, split=False):
        if split:
            self._bucket2.unpack(_world())
        if self._fused_adam2 is None:
            self._fused_adam2 = FusedClipAdam(self.lib, self.optimizer2, self._params2, self.max_grad_norm)
        if self._fused_adam2.usable and self._fused_adam2.step():
            return
        nn.utils.clip_grad_norm_(self._params2, self.max_grad_norm, foreach=True)
        self.optimizer2.step()

    def _student_backward(self, hist_s, priv_s, teacher_latent):
        """forward + backward of the student loss on one mini-batch's student rows (graph mode); the logs are added to the accumulators"""
        loss, logs = self._student_losses(hist_s, priv_s, teacher_latent) if teacher_latent is not None else self._student_losses(hist_s, priv_s)
        loss.backward()
        self._acc[3 + self._NUM_POLICY_LOGS:].add_(torch.stack([v.detach() for v in logs]))

    def _student_step(self, i):
        self._student_front(i)
        self._student_back()

    def _gather_update(self, order):
        """The rollout into mini-batch order, once per update.  Own path: ONE go2sim_shuffle_gather launch over the given index list — obs / privileged obs land
        straight in the column blocks behind the latent of the actor's / critic's input matrices (dst_pitch) — which also clears the loss accumulators."""
        if self._gather_jobs is None:
            self._accbuf.zero_()
            for k in self._KEYS:
                torch.index_select(self._flat[k], 0, order, out=self._perm[k])
            return
        import ctypes as C
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if str(self.device).startswith("cuda") else None
        self._order_live = order.to(torch.int64).contiguous()          # (alive until the launch has run)
        rc = self.lib.go2sim_shuffle_gather(self._gather_jobs, len(self._gather_jobs), int(order.numel()), C.c_void_p(self._order_live.data_ptr()), None,
                                            C.c_void_p(self._accbuf.data_ptr()), int(self._accbuf.numel()), stream)
        if rc != 0:
            raise RuntimeError("go2sim_shuffle_gather failed: %s" % self.lib.go2sim_last_error().decode())

    def _update_head(self):
        """rollout_storage_cts.py:152-160 on the device: the two keyed permutations -> the update's index list (go2sim_cts_minibatch_indices), the gather, the student latents"""
        import ctypes as C
        st, nmb = self.storage, self.num_mini_batches
        T = st.num_transitions_per_env
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if str(self.device).startswith("cuda") else None
        rc = self.lib.go2sim_cts_minibatch_indices(C.c_void_p(self._order.data_ptr()), nmb, st.teacher_num_envs * T, st.student_num_envs * T, C.c_void_p(st.ref2mine.data_ptr()),
                                                   C.c_void_p(self._shuffle_key.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError("go2sim_cts_minibatch_indices failed: %s" % self.lib.go2sim_last_error().decode())
        self._gather_update(self._order)
        self._student_latents()

    def _student_takes_teacher_latent(self):
        import inspect
        return "teacher_latent" in inspect.signature(self._student_losses).parameters

    def _teacher_latents(self):
        from ..modules import fused_cts
        nmb, mb, n_t = self.num_mini_batches, self._mb, self._teacher_rows()
        priv_s = self._perm["cobs"].view(nmb, mb, -1)[:, n_t:].reshape(nmb * (mb - n_t), -1)          # all mini-batches' student rows: one forward
        fused_cts.encoder_latents(self._plan, self._plan.teacher, priv_s, self._tlat.view(nmb * (mb - n_t), -1), None)

    def _student_latents(self):
        """The student rows' latents of the whole update into the first L columns of both input matrices: optimizer1 never touches the student encoder (cts.py:72-77), so
        they are constants of the policy epochs — computed once per update instead of in each of the 20 policy steps."""
        from ..modules import fused_cts
        nmb, mb, n_t, plan = self.num_mini_batches, self._mb, self._teacher_rows(), self._plan
        L, P = plan.L, self._perm
        hist = P["hist"].view(nmb, mb, -1)[:, n_t:].reshape(nmb * (mb - n_t), -1)
        from ..modules.fused import own_forward
        with torch.no_grad(), own_forward():          # (the MoE encoders' Linear / ELU stacks on the library's kernels also without a gradient)
            n_s = mb - n_t
            if plan.student is not None:              # a plain MLP: ONE forward over all mini-batches' student rows on the own kernels; the normaliser writes into both matrices
                z = fused_cts.encoder_forward(plan.student, hist).view(nmb, n_s, L)
                k = fused_cts._Launch(hist.device)
            else:                                     # (the MoE encoders: torch modules, no gradient)
                lat = self.model.student_latent(hist)[0].view(nmb, n_s, L)
            for i in range(nmb):
                da, dc = P["ain"][i * mb + n_t:(i + 1) * mb], P["cin"][i * mb + n_t:(i + 1) * mb]
                if plan.student is not None:
                    fused_cts.latent_concat(k, z[i], da, dc)
                else:
                    da[:, :L] = lat[i]; dc[:, :L] = lat[i]

    def _update_graphs(self):
        st, nmb = self.storage, self.num_mini_batches
        if self._steps is None:
            plan = self._own_plan()
            self._flat = st.flat()
            self._mb = (st.teacher_num_envs * st.num_transitions_per_env) // nmb + (st.student_num_envs * st.num_transitions_per_env) // nmb
            rows = nmb * self._mb
            new = lambda *shape: torch.empty(*shape, device=self.device, dtype=torch.float32)
            self._perm = {k: new(rows, *self._flat[k].shape[1:]) for k in self._KEYS if not (plan is not None and k == "obs")}
            # accumulators: [own policy step: surrogate, value, KL, entropy | own student step: latent loss, 0, 0, 0 | the autograd formulation's: value, surrogate, entropy, logs]
            self._accbuf = torch.zeros(8 + 3 + self._NUM_POLICY_LOGS + self._NUM_STUDENT_LOGS, device=self.device)
            self._acc_own, self._acc = self._accbuf[:8], self._accbuf[8:]
            self._bucket1 = self._bucket2 = None
            self._gather_jobs = None
            self._tlat, self._tlat_step = None, None
            if plan is not None:
                import ctypes as C
                from ..._abi import Go2GatherJob
                L, no, npriv = plan.L, self._flat["obs"].shape[1], self._flat["cobs"].shape[1]
                self._perm["ain"], self._perm["cin"] = new(rows, L + no), new(rows, L + npriv)
                F, Pm = self._flat, self._perm
                jobs = [Go2GatherJob(F["obs"].data_ptr(), Pm["ain"].data_ptr() + 4 * L, no, L + no), Go2GatherJob(F["cobs"].data_ptr(), Pm["cin"].data_ptr() + 4 * L, npriv, L + npriv)]
                jobs += [Go2GatherJob(F[k].data_ptr(), Pm[k].data_ptr(), int(F[k][0].numel()), 0) for k in self._KEYS if k != "obs"]
                assert all(F[k].dtype == torch.float32 and F[k].is_contiguous() for k in self._KEYS)
                self._gather_jobs = (Go2GatherJob * len(jobs))(*jobs)
                self._make_shuffle_key()
                self._order = torch.empty(rows, dtype=torch.int64, device=self.device)
                if not self._own_student() and self._student_takes_teacher_latent():
                    self._tlat = new(nmb, self._mb - self._teacher_rows(), L)
                    self._tlat_step = CapturedStep(self._teacher_latents, enabled=self._capture, warmup=2, name="CTS teacher latents of the student rows", optional=True)
                self._head_step = CapturedStep(self._update_head, enabled=self._capture, warmup=2, name="CTS update head (permutation, gather, student latents)", optional=True)
            self._step_reps = None          # (set by the branch whose graphs span every epoch)
            if _collectives_on():     # two captured halves per slot, the gradient all-reduce eager between them
                mk = lambda front, back, bucket, name: [ReducedStep((lambda i=i: front(i, True)), (lambda: back(True)), bucket, enabled=self._capture, warmup=3 if i == 0 else 1,
                                                                    name="CTS %s step %d" % (name, i)) for i in range(nmb)]
                self._steps = (mk(self._policy_front, self._policy_back, (lambda: self._bucket1), "policy"),
                               mk(self._student_front, self._student_back, (lambda: self._bucket2), "student"))
            elif plan is not None:
                # one rank, own path: a whole PHASE (every epoch's nmb steps, each on its own chunk of the permuted rollout) is one graph, as in PPO (1 + 1 graph launches per update instead of 20 + 20)
                ne = self.num_learning_epochs
                mk = lambda fn, name: [CapturedStep((lambda: [fn(i) for _ in range(ne) for i in range(nmb)] and None), enabled=self._capture, warmup=1, name="CTS %s phase (%d epochs x %d steps)" % (name, ne, nmb))]
                self._steps = (mk(self._policy_step, "policy"), mk(self._student_step, "student"))
                self._step_reps = 1
            else:
                mk = lambda fn, name: [CapturedStep((lambda i=i: fn(i)), enabled=self._capture, warmup=3 if i == 0 else 1, name="CTS %s step %d" % (name, i)) for i in range(nmb)]
                self._steps = (mk(self._policy_step, "policy"), mk(self._student_step, "student"))
        # the rollout is gathered ONCE per update into mini-batch order ([teacher rows | student rows] per mini-batch; every epoch
        # reuses the same permutation, rollout_storage_cts.py:152-160), so each captured step reads a contiguous chunk
        if self._plan is not None and torch.randperm is _RANDPERM and hasattr(self.lib, "go2sim_cts_minibatch_indices"):
            self._head_step()          # keyed permutations on the device + gather + student latents: three launches (+ the MoE student encoder), one graph
        else:                          # (the goldens replace torch.randperm to replay the reference's draws; the autograd formulation)
            self._gather_update(torch.cat(st.mini_batch_indices(nmb)))
            if self._plan is not None:
                self._student_latents()
        for phase, steps in enumerate(self._steps):
            if phase == 1 and self._tlat_step is not None:
                self._tlat_step()          # the teacher latents of the student rows: constants of the student epochs, computed once (own kernels) instead of in each of the 20 steps
            for _ in range(self._step_reps or self.num_learning_epochs):
                for step in steps:
                    step()
        n = self.num_learning_epochs * nmb
        acc = self._acc
        if self._plan is not None:      # the own steps' running sums, in the autograd formulation's slots
            own, P = self._acc_own, self._NUM_POLICY_LOGS
            parts = [own[1:2], own[0:1], own[3:4], acc[3:3 + P], own[4:5] if self._own_student() else acc[3 + P:3 + P + 1], acc[3 + P + 1:]]
            acc = torch.cat(parts)
        out = torch.cat([acc / n, self._lr_t.reshape(1)]).tolist()          # ONE device -> host read per update
        self.learning_rate = float(out.pop())
        return self._ordered(tuple(out))

    def graphs_captured(self):
        """True iff every policy / student mini-batch step is being replayed from a HIP graph."""
        # (the update head and the teacher-latent step are `optional` CapturedSteps — a failed capture keeps them eager even under GO2_STRICT_GRAPHS — and
        #  count here once they exist and had their warm-up calls)
        heads = [h for h in (getattr(self, "_head_step", None), getattr(self, "_tlat_step", None)) if h is not None and h.calls > 0]          # (calls > 0: in use)
        return bool(self.use_graphs and self._capture and all_captured(self._steps) and all(h.graph is not None for h in heads))

    def update(self):
        self._pk_packed = False          # the optimizer steps below change the parameters: the next rollout re-packs
        self._img_cache = None
        out = self._update_graphs() if self.use_graphs else self._update_eager()
        self.storage.clear()
        return out
