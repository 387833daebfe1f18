"""PPO with clipped surrogate / clipped value loss / adaptive-KL learning rate (rsl_rl/rsl_rl/algorithms/ppo.py:38-187).

Multi-GPU (one process per GPU, torch.distributed backend 'nccl' = RCCL): each rank owns an env shard and a full
policy replica.  To train ONE coherent policy the gradients are averaged across ranks before the clip/step and the
mean KL is averaged before the learning-rate decision, so every rank takes the same branch; the
advantage statistics are all-reduced in RolloutStorage.compute_returns.  With world_size 1 none of this runs.
"""
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim

from ..storage import RolloutStorage


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


class PPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95, value_loss_coef=1.0,
                 entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01,
                 device="cpu", lib=None):
        self.device = device
        self.lib = lib
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic
        self.actor_critic.to(self.device)
        self.storage = None
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=learning_rate)
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        if _world() > 1:   # identical initial replicas
            for p in self.actor_critic.parameters():
                dist.broadcast(p.data, src=0)

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device, lib=self.lib)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    def act(self, obs, critic_obs):
        t = self.transition
        t.actions = self.actor_critic.act(obs).detach()
        t.values = self.actor_critic.evaluate(critic_obs).detach()
        t.actions_log_prob = self.actor_critic.get_actions_log_prob(t.actions).detach()
        t.action_mean = self.actor_critic.action_mean.detach()
        t.action_sigma = self.actor_critic.action_std.detach()
        t.observations, t.critic_observations = obs, critic_obs
        return t.actions

    def process_env_step(self, rewards, dones, infos):
        t = self.transition
        t.rewards = rewards.clone()
        t.dones = dones
        if "time_outs" in infos:   # bootstrap on time-outs (ppo.py:107-108)
            t.rewards += self.gamma * torch.squeeze(t.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    def update(self):
        mean_value_loss, mean_surrogate_loss = 0.0, 0.0
        world = _world()
        gen = self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs)
        for obs_b, cobs_b, act_b, tv_b, adv_b, ret_b, old_lp_b, old_mu_b, old_sig_b, _, _ in gen:
            self.actor_critic.act(obs_b)
            lp_b = self.actor_critic.get_actions_log_prob(act_b)
            val_b = self.actor_critic.evaluate(cobs_b)
            mu_b, sig_b, ent_b = self.actor_critic.action_mean, self.actor_critic.action_std, self.actor_critic.entropy
            if self.desired_kl is not None and self.schedule == "adaptive":
                with torch.inference_mode():
                    kl = torch.sum(torch.log(sig_b / old_sig_b + 1.0e-5) + (torch.square(old_sig_b) + torch.square(old_mu_b - mu_b)) / (2.0 * torch.square(sig_b)) - 0.5, axis=-1)
                    kl_mean = torch.mean(kl)
                    if world > 1:
                        dist.all_reduce(kl_mean, op=dist.ReduceOp.SUM)
                        kl_mean /= world
                    if kl_mean > self.desired_kl * 2.0:
                        self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                    elif kl_mean < self.desired_kl / 2.0 and kl_mean > 0.0:
                        self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                    for g in self.optimizer.param_groups:
                        g["lr"] = self.learning_rate
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
            self.optimizer.zero_grad()
            loss.backward()
            if world > 1:   # one flat bucket: 1.96 MB of fp32 gradients, latency-bound on xGMI
                grads = [p.grad for p in self.actor_critic.parameters() if p.grad is not None]
                flat = torch.cat([g.reshape(-1) for g in grads])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat /= world
                off = 0
                for g in grads:
                    n = g.numel(); g.copy_(flat[off:off + n].view_as(g)); off += n
            nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
            self.optimizer.step()
            mean_value_loss += value_loss.item()
            mean_surrogate_loss += surrogate_loss.item()
        n = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        return mean_value_loss / n, mean_surrogate_loss / n
