"""CTS with a multiplicative-compositional-policy actor (rsl_rl/rsl_rl/modules/actor_critic_mcp_cts.py:17-230): E primitives, each a
diagonal Gaussian (mu_e, sigma_e) from the command-free input [latent, obs without command], composed with sigmoid gates g_e from
the full input [latent, obs]:      1/sigma^2 = sum_e g_e / sigma_e^2 ,   mu = sigma^2 * sum_e g_e mu_e / sigma_e^2 .
The action std is therefore STATE-DEPENDENT and there is no `std` parameter; the algorithm takes the un-fused loss / rollout heads
(the fused kernels assume one std per action dimension).  Parameter names follow the reference (`actor_mcp.{gating_network,
experts_backbone,experts_hidden,experts_out}`, `teacher_encoder.N`, `student_encoder.N`, `critic.N`)."""
import torch
import torch.nn as nn

from .actor_critic import _mlp, get_activation
from .actor_critic_cts import ActorCriticCTS
from .fused import FusedSequential
from .utils import GroupedHeads


class ActorMCP(nn.Module):
    def __init__(self, input_dim, input_dim_no_goal, action_dim, hidden_dims=(512, 256), expert_num=8, expert_hidden_dim=256, activation="elu"):
        super().__init__()
        self.expert_num, self.action_dim = expert_num, action_dim
        layers, last = [], input_dim
        for h in hidden_dims:
            layers += [nn.Linear(last, h), get_activation(activation)]
            last = h
        layers += [nn.Linear(last, expert_num), nn.Sigmoid()]
        self.gating_network = nn.Sequential(*layers)
        layers, last = [], input_dim_no_goal
        for h in hidden_dims:
            layers += [nn.Linear(last, h), get_activation(activation)]
            last = h
        self.experts_backbone = FusedSequential(*layers)
        self.experts_hidden = FusedSequential(nn.Linear(last, expert_num * expert_hidden_dim), get_activation(activation))
        self.experts_out = GroupedHeads(expert_num, expert_hidden_dim, action_dim * 2)

    def forward(self, x, x_no_goal):
        weights = self.gating_network(x).unsqueeze(-1)                                                   # [B, E, 1]
        out = self.experts_out(self.experts_hidden(self.experts_backbone(x_no_goal)))                     # [B, E, 2A]
        mu, log_std = torch.chunk(out, 2, dim=-1)
        var = torch.exp(2 * torch.clamp(log_std, -5.0, 2.0)) + 1e-9
        var_total = 1.0 / (torch.sum(weights / var, dim=1) + 1e-9)
        return var_total * torch.sum(weights * mu / var, dim=1), torch.sqrt(var_total), weights.squeeze(-1)


class ActorCriticMCPCTS(ActorCriticCTS):
    state_dependent_std = True

    def __init__(self, num_obs, num_critic_obs, num_actions, num_envs, history_length, obs_no_goal_mask, actor_hidden_dims=(512, 256),
                 critic_hidden_dims=(512, 256, 128), teacher_encoder_hidden_dims=(512, 256), student_encoder_hidden_dims=(512, 256),
                 student_expert_num=8, activation="elu", latent_dim=32, norm_type="l2norm", **kwargs):
        self._mask_list, self._expert_num, self._latent_dim = list(obs_no_goal_mask), student_expert_num, latent_dim
        kwargs.pop("init_noise_std", None)
        super().__init__(num_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims, critic_hidden_dims,
                         teacher_encoder_hidden_dims, student_encoder_hidden_dims, activation, 1.0, latent_dim, norm_type, **kwargs)
        del self.std                                    # the std comes out of the actor
        self.register_buffer("obs_no_goal_mask", torch.tensor(self._mask_list, dtype=torch.bool), persistent=False)
        self.register_buffer("_no_goal_idx", torch.nonzero(self.obs_no_goal_mask).flatten(), persistent=False)

    def _build_heads(self, n_a, n_c, a_hidden, c_hidden, num_actions, activation):
        n_no_goal = sum(bool(m) for m in self._mask_list)
        self.actor_mcp = ActorMCP(n_a, self._latent_dim + n_no_goal, num_actions, list(a_hidden), self._expert_num, 256, activation)
        self.critic = _mlp(n_c, c_hidden, 1, activation)

    @property
    def actor(self):
        return self.actor_mcp

    def _actor_inputs(self, latent, obs):
        return torch.cat([latent, obs], dim=1), torch.cat([latent, obs.index_select(1, self._no_goal_idx)], dim=1)

    def policy_dist(self, latent, obs):
        mean, std, _ = self.actor_mcp(*self._actor_inputs(latent, obs))
        return mean, std

    def policy_mean(self, latent, obs):
        return self.policy_dist(latent, obs)[0]

    def policy_parameter_groups(self):
        return [list(self.teacher_encoder.parameters()), list(self.critic.parameters()), list(self.actor_mcp.parameters())]     # mcp_cts.py:72-76
