"""CTS with a mixture-of-experts student encoder (rsl_rl/rsl_rl/modules/actor_critic_moe_cts.py:20-141): 8 experts over the
5-frame history, soft gate, load-balance regulariser in the algorithm.  Parameter names follow the reference
(`teacher_encoder.0.network.N`, `student_moe_encoder.moe.{experts.backbone,experts.experts,gating_network.0}`,
`actor.network.N`, `critic.network.N`, `std`)."""
import torch.nn as nn

from .actor_critic_cts import ActorCriticCTS
from .utils import MLP, StudentMoEEncoder, make_norm


class ActorCriticMoECTS(ActorCriticCTS):
    def __init__(self, num_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims=(512, 256, 128),
                 critic_hidden_dims=(512, 256, 128), teacher_encoder_hidden_dims=(512, 256), student_encoder_hidden_dims=(512, 256, 256),
                 expert_num=8, activation="elu", init_noise_std=1.0, latent_dim=32, norm_type="l2norm", **kwargs):
        self._expert_num = expert_num
        super().__init__(num_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims, critic_hidden_dims,
                         teacher_encoder_hidden_dims, student_encoder_hidden_dims, activation, init_noise_std, latent_dim, norm_type, **kwargs)

    def _build_encoders(self, n_obs, n_priv, H, t_hidden, s_hidden, activation, latent_dim, norm_type, extra):
        self.teacher_encoder = nn.Sequential(MLP([n_priv, *t_hidden, latent_dim], activation=activation), make_norm(norm_type))
        self.student_moe_encoder = StudentMoEEncoder(self._expert_num, n_obs * H, list(s_hidden), latent_dim, activation, norm_type)

    def _build_heads(self, n_a, n_c, a_hidden, c_hidden, num_actions, activation):
        self.actor = MLP([n_a, *a_hidden, num_actions], activation=activation)
        self.critic = MLP([n_c, *c_hidden, 1], activation=activation)

    def student_parameters(self):
        return self.student_moe_encoder.parameters()

    def student_latent(self, history):
        return self.student_moe_encoder(history)

    def student_moe_parts(self, history):
        return self.student_moe_encoder.moe.parts(history)
