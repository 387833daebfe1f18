"""Concurrent Teacher-Student actor-critic (rsl_rl/rsl_rl/modules/actor_critic_cts.py:18-180; CTS, arXiv 2405.10830).

  teacher_encoder : privileged obs [263]          -> latent [32], L2-normalised
  student_encoder : 5 stacked proprioceptive obs  -> latent [32], L2-normalised (trained to imitate the teacher's latent)
  actor           : [latent, obs]  -> action mean ;  critic : [latent.detach(), privileged obs] -> value

Parameter names match the reference (`teacher_encoder.0.weight`, `student_encoder.4.bias`, `actor.6.weight`, `std`) so its
checkpoints — e.g. deploy/pre_train/go2/go2_cts_150k.pt — load unchanged.

Besides the reference's per-group interface (`act/evaluate(..., is_teacher)`), `latents()` / `act_joint()` /
`evaluate_joint()` serve a whole batch laid out [teacher rows | student rows] with ONE actor and ONE critic pass (the rows
are independent, so the numbers are the same); that is what the algorithm uses on the GPU."""
import torch
import torch.nn as nn
from torch.distributions import Normal

from .actor_critic import _mlp
from .utils import make_norm


def _encoder(n_in, hidden, latent_dim, activation, norm_type):
    seq = _mlp(n_in, hidden, latent_dim, activation)
    seq.append(make_norm(norm_type))
    return seq


class ActorCriticCTS(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, num_envs, history_length, actor_hidden_dims=(512, 256, 128),
                 critic_hidden_dims=(512, 256, 128), teacher_encoder_hidden_dims=(512, 256), student_encoder_hidden_dims=(512, 256),
                 activation="elu", init_noise_std=1.0, latent_dim=32, norm_type="l2norm", **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs.keys())))
        super().__init__()
        self.num_actions, self.history_length, self.num_actor_obs = num_actions, history_length, num_actor_obs
        self._build_encoders(num_actor_obs, num_critic_obs, history_length, teacher_encoder_hidden_dims, student_encoder_hidden_dims,
                             activation, latent_dim, norm_type, kwargs)
        self._build_heads(latent_dim + num_actor_obs, latent_dim + num_critic_obs, actor_hidden_dims, critic_hidden_dims, num_actions, activation)
        # deployment-side history (act_inference); a non-persistent buffer so it follows .to(device) — the reference hard-codes 'cuda' (:48)
        self.register_buffer("history", torch.zeros(num_envs, history_length, num_actor_obs), persistent=False)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        Normal.set_default_validate_args = False

    # -- construction hooks (the MoE variant overrides the encoders / heads layout) --
    def _build_encoders(self, n_obs, n_priv, H, t_hidden, s_hidden, activation, latent_dim, norm_type, extra):
        self.teacher_encoder = _encoder(n_priv, t_hidden, latent_dim, activation, norm_type)
        self.student_encoder = _encoder(n_obs * H, s_hidden, latent_dim, activation, norm_type)

    def _build_heads(self, n_a, n_c, a_hidden, c_hidden, num_actions, activation):
        self.actor = _mlp(n_a, a_hidden, num_actions, activation)
        self.critic = _mlp(n_c, c_hidden, 1, activation)

    def student_parameters(self):
        return self.student_encoder.parameters()

    # -- the two heads as hooks (the AC-MoE variants replace them): action mean from [latent, obs]; value (+ auxiliary gate weights) --
    def policy_mean(self, latent, obs):
        return self.actor(torch.cat([latent, obs], dim=1))

    state_dependent_std = False
    heads_share_parameters = False      # True when value() reuses actor modules (AC-MoE gate): no two-stream split of the two heads then

    def policy_dist(self, latent, obs):
        mean = self.policy_mean(latent, obs)
        return mean, mean * 0.0 + self.std

    def value(self, latent, obs, privileged_obs):
        return self.critic(torch.cat([latent.detach(), privileged_obs], dim=1)), None

    def policy_parameter_groups(self):
        """optimizer1's param groups, in the reference's order (cts.py:72-77)."""
        return [list(self.teacher_encoder.parameters()), list(self.critic.parameters()), list(self.actor.parameters()), [self.std]]

    def student_latent(self, history):
        """-> (latent, gating weights or None)"""
        return self.student_encoder(history), None

    # -- reference surface ---------------------------------------------------------------------------------
    def reset(self, dones=None):
        self.history[dones > 0] = 0.0

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        # whatever comes with a checkpoint: the buffer is looked at once (one host read at load time) — a training checkpoint carries zeros, and a resumed run then
        # takes the same two launches per env step fewer that a fresh one does (resumed and fresh benchmarks differed silently)
        self._history_dirty = bool(self.history.abs().sum().item() != 0.0)
        return out

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def update_distribution(self, latent_and_obs):
        L = latent_and_obs.shape[1] - self.num_actor_obs
        mean, std = self.policy_dist(latent_and_obs[:, :L], latent_and_obs[:, L:])
        self.distribution = Normal(mean, std, validate_args=False)

    def _noise(self, like):
        return torch.randn_like(like)

    def _sample(self):
        d = self.distribution                    # mean + std * eps: same law as .sample(), capturable in a HIP graph
        return (d.mean + d.stddev * self._noise(d.mean)).detach()

    def act(self, obs, privileged_obs, history, is_teacher, **kwargs):
        latent = self.teacher_encoder(privileged_obs) if is_teacher else self.student_latent(history)[0].detach()
        self.update_distribution(torch.cat([latent, obs], dim=1))
        return self._sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    _history_dirty = False      # the deployment-side history holds something other than zeros (then CTS.process_env_step zeroes finished envs' rows, cts.py:163)

    def act_inference(self, obs):
        self._history_dirty = True
        self.history = torch.cat([self.history[:, 1:], obs.unsqueeze(1)], dim=1)
        latent = self.student_latent(self.history.flatten(1))[0]
        return self.policy_mean(latent, obs)

    def evaluate(self, privileged_obs, history, is_teacher, **kwargs):
        latent = self.teacher_encoder(privileged_obs) if is_teacher else self.student_latent(history)[0]
        return self.value(latent, None, privileged_obs)[0]

    # -- whole-batch surface: rows [0, n_teacher) are teacher rows, the rest student rows --------------------
    def latents(self, privileged_obs, history, n_teacher, pair=None):
        """pair (optional): callable (f, g) -> (f(), g()) that may run g on a second stream (the two encoders are independent chains)."""
        def student():
            with torch.no_grad():                                      # the policy loss never reaches the student encoder (:145), and its
                return self.student_latent(history[n_teacher:])[0]     # parameters are not in optimizer1: no activations kept for a backward
        teacher = lambda: self.teacher_encoder(privileged_obs[:n_teacher])
        lt, ls = pair(teacher, student) if pair is not None else (teacher(), student())
        return torch.cat([lt, ls], dim=0)

    def act_joint(self, obs, latent):
        self.update_distribution(torch.cat([latent, obs], dim=1))
        return self._sample()

    def evaluate_joint(self, privileged_obs, latent, obs=None):
        return self.value(latent, obs, privileged_obs)[0]
