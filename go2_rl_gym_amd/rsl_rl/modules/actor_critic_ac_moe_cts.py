"""CTS variants with a mixture-of-experts ACTOR and expert CRITIC sharing the actor's gate
(rsl_rl/rsl_rl/modules/actor_critic_ac_moe_cts.py:20-146, actor_critic_dual_moe_cts.py:19-149):

  mean  = sum_e g_e(x_a) * actor_expert_e(x_a),            x_a = [latent, obs]
  value = sum_e g_e(x_a) * critic_expert_e(x_c),           x_c = [latent.detach(), privileged obs]

AC-MoE keeps the plain student encoder, Dual-MoE also uses the MoE student encoder of MoE-CTS.  Parameter names follow the
reference (`actor_moe.{experts,gating_network}`, `critic_experts.{backbone,experts}`, `teacher_encoder.0.network.N`,
`student_encoder.0.network.N` / `student_moe_encoder.moe...`, `std`)."""
import torch
import torch.nn as nn

from .actor_critic_cts import ActorCriticCTS
from .utils import MLP, Experts, MoE, StudentMoEEncoder, make_norm


class ActorCriticACMoECTS(ActorCriticCTS):
    heads_share_parameters = True       # the critic mixes its experts with the ACTOR's gate

    def __init__(self, num_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims=(512, 256, 128),
                 critic_hidden_dims=(512, 256, 128), teacher_encoder_hidden_dims=(512, 256), student_encoder_hidden_dims=(512, 256),
                 expert_num=8, activation="elu", init_noise_std=1.0, latent_dim=32, norm_type="l2norm", **kwargs):
        self._expert_num = expert_num
        super().__init__(num_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims, critic_hidden_dims,
                         teacher_encoder_hidden_dims, student_encoder_hidden_dims, activation, init_noise_std, latent_dim, norm_type, **kwargs)

    def _build_encoders(self, n_obs, n_priv, H, t_hidden, s_hidden, activation, latent_dim, norm_type, extra):
        self.teacher_encoder = nn.Sequential(MLP([n_priv, *t_hidden, latent_dim], activation), make_norm(norm_type))
        self.student_encoder = nn.Sequential(MLP([n_obs * H, *s_hidden, latent_dim], activation), make_norm(norm_type))

    def _build_heads(self, n_a, n_c, a_hidden, c_hidden, num_actions, activation):
        self.actor_moe = MoE(self._expert_num, n_a, list(a_hidden), num_actions, activation)
        self.critic_experts = Experts(self._expert_num, n_c, list(c_hidden[:-1]), c_hidden[-1], 1, activation)

    @property
    def actor(self):                       # the deployment exporter looks for `actor` / `actor_moe`
        return self.actor_moe

    def policy_mean(self, latent, obs):
        return self.actor_moe(torch.cat([latent, obs], dim=1))[0]

    def value(self, latent, obs, privileged_obs):
        weights = self.actor_moe.gating_network(torch.cat([latent, obs], dim=1))                   # [B, E]  (latent NOT detached here, :127-129)
        experts_value = self.critic_experts(torch.cat([latent.detach(), privileged_obs], dim=1))   # [B, E, 1]
        return torch.sum(weights.unsqueeze(-1) * experts_value, dim=1), weights

    def policy_parameter_groups(self):
        return [list(self.teacher_encoder.parameters()), list(self.critic_experts.parameters()), list(self.actor_moe.parameters()), [self.std]]

    def evaluate(self, obs, privileged_obs, history, is_teacher, **kwargs):
        latent = self.teacher_encoder(privileged_obs) if is_teacher else self.student_latent(history)[0]
        return self.value(latent, obs, privileged_obs)


class ActorCriticDualMoECTS(ActorCriticACMoECTS):
    def __init__(self, num_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims=(512, 256, 128),
                 critic_hidden_dims=(512, 256, 128), teacher_encoder_hidden_dims=(512, 256), student_encoder_hidden_dims=(512, 256, 256),
                 expert_num=8, activation="elu", init_noise_std=1.0, latent_dim=32, norm_type="l2norm", **kwargs):
        super().__init__(num_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims, critic_hidden_dims,
                         teacher_encoder_hidden_dims, student_encoder_hidden_dims, expert_num, activation, init_noise_std, latent_dim, norm_type, **kwargs)

    def _build_encoders(self, n_obs, n_priv, H, t_hidden, s_hidden, activation, latent_dim, norm_type, extra):
        self.teacher_encoder = nn.Sequential(MLP([n_priv, *t_hidden, latent_dim], activation), make_norm(norm_type))
        self.student_moe_encoder = StudentMoEEncoder(self._expert_num, n_obs * H, list(s_hidden), latent_dim, activation, norm_type)

    def student_parameters(self):
        return self.student_moe_encoder.parameters()

    def student_latent(self, history):
        return self.student_moe_encoder(history)
