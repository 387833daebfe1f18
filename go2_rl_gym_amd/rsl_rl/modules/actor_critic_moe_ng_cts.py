"""MoE-CTS whose experts do not see the command ("no goal"): the gate reads the full 5-frame history, the experts the history with
the 3 command entries of every frame masked out (rsl_rl/rsl_rl/modules/actor_critic_moe_ng_cts.py:17-240).  Parameter names
follow the reference (`teacher_encoder.N`, `student_moe_encoder.{experts_backbone,experts_hidden,experts_out,gating_network}`,
`actor.N`, `critic.N`, `std`); the expert heads run as one batched GEMM on Conv1d-shaped parameters (modules/utils.py)."""
import torch
import torch.nn as nn

from .actor_critic import get_activation
from .actor_critic_cts import ActorCriticCTS, _encoder
from .fused import FusedSequential
from .utils import GroupedHeads, make_norm


class StudentMoENoGoalEncoder(nn.Module):
    def __init__(self, expert_dim, gating_dim, hidden_dims=(512, 256), expert_num=8, expert_hidden_dim=256, latent_dim=32, activation="elu", norm_type="l2norm"):
        super().__init__()
        self.expert_num, self.latent_dim = expert_num, latent_dim
        self.norm_layer = make_norm(norm_type)
        layers, last = [], expert_dim
        for h in hidden_dims:
            layers += [nn.Linear(last, h), get_activation(activation)]
            last = h
        self.experts_backbone = FusedSequential(*layers)
        self.experts_hidden = FusedSequential(nn.Linear(last, expert_num * expert_hidden_dim), get_activation(activation))
        self.experts_out = GroupedHeads(expert_num, expert_hidden_dim, latent_dim)
        layers, last = [], gating_dim
        for h in hidden_dims:
            layers += [nn.Linear(last, h), get_activation(activation)]
            last = h
        layers += [nn.Linear(last, expert_num), nn.Softmax(dim=-1)]
        self.gating_network = nn.Sequential(*layers)

    def forward(self, obs, obs_no_goal):
        weights = self.gating_network(obs)                                            # [B, E]
        expert_latent = self.experts_out(self.experts_hidden(self.experts_backbone(obs_no_goal)))   # [B, E, latent]
        return self.norm_layer(torch.sum(weights.unsqueeze(-1) * expert_latent, dim=1)), weights


class ActorCriticMoENGCTS(ActorCriticCTS):
    def __init__(self, num_obs, num_critic_obs, num_actions, num_envs, history_length, obs_no_goal_mask, actor_hidden_dims=(512, 256, 128),
                 critic_hidden_dims=(512, 256, 128), teacher_encoder_hidden_dims=(512, 256), student_encoder_hidden_dims=(512, 256),
                 student_expert_num=8, activation="elu", init_noise_std=1.0, latent_dim=32, norm_type="l2norm", **kwargs):
        self._mask_list, self._expert_num = list(obs_no_goal_mask), student_expert_num
        super().__init__(num_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims, critic_hidden_dims,
                         teacher_encoder_hidden_dims, student_encoder_hidden_dims, activation, init_noise_std, latent_dim, norm_type, **kwargs)
        self.register_buffer("obs_no_goal_mask", torch.tensor(self._mask_list, dtype=torch.bool), persistent=False)
        # integer form of the mask: a boolean-mask index is a nonzero() + host sync, which a HIP-graph capture does not allow
        self.register_buffer("_no_goal_idx", torch.nonzero(self.obs_no_goal_mask).flatten(), persistent=False)

    def _build_encoders(self, n_obs, n_priv, H, t_hidden, s_hidden, activation, latent_dim, norm_type, extra):
        self.teacher_encoder = _encoder(n_priv, t_hidden, latent_dim, activation, norm_type)
        n_expert_in = sum(bool(m) for m in self._mask_list) * H
        self.student_moe_encoder = StudentMoENoGoalEncoder(n_expert_in, n_obs * H, list(s_hidden), self._expert_num, 256, latent_dim, activation)

    def student_parameters(self):
        return self.student_moe_encoder.parameters()

    def get_student_latent_and_weights(self, history):
        B = history.shape[0]
        no_goal = history.reshape(B, self.history_length, -1).index_select(2, self._no_goal_idx).reshape(B, -1)
        return self.student_moe_encoder(history, no_goal)

    def student_latent(self, history):
        return self.get_student_latent_and_weights(history)

    def student_moe_parts(self, history):
        """-> (gate logits, expert outputs [E, B, latent] before their bias, the heads' bias) ahead of the mixture (the fused loss head of the student step, fused_cts.moe_head_grads)"""
        B, enc = history.shape[0], self.student_moe_encoder
        no_goal = history.reshape(B, self.history_length, -1).index_select(2, self._no_goal_idx).reshape(B, -1)
        return enc.gating_network[:-1](history), enc.experts_out.expert_major(enc.experts_hidden(enc.experts_backbone(no_goal)), with_bias=False), enc.experts_out.bias
