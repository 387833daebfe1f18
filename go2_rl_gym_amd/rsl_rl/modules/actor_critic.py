"""MLP actor-critic with a state-independent Gaussian (rsl_rl/rsl_rl/modules/actor_critic.py:38-136).
Parameter names (`actor.N.weight`, `critic.N.bias`, `std`) match the reference so checkpoints interchange
(on_policy_runner.py:243-250).  The MLPs are FusedSequential modules: under autograd their layers run on the fp32-MFMA kernels of include/go2nn.h
(modules/fused.py; in PPO.update both networks as one node with grouped launches), in the rollout as one launch (go2nn_policy_act)."""
import torch
import torch.nn as nn
from torch.distributions import Normal

from .fused import FusedSequential

_ACT = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


def get_activation(name):
    if name not in _ACT:
        print("invalid activation function!")
        return None
    return _ACT[name]()


def _mlp(n_in, hidden, n_out, act):
    dims = [n_in] + list(hidden)
    layers = []
    for a, b in zip(dims[:-1], dims[1:]):
        layers += [nn.Linear(a, b), get_activation(act)]
    layers.append(nn.Linear(dims[-1], n_out))
    return FusedSequential(*layers)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(256, 256, 256), critic_hidden_dims=(256, 256, 256),
                 activation="elu", init_noise_std=1.0, **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs.keys())))
        super().__init__()
        self.actor = _mlp(num_actor_obs, actor_hidden_dims, num_actions, activation)
        self.critic = _mlp(num_critic_obs, critic_hidden_dims, 1, activation)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        Normal.set_default_validate_args = False

    def reset(self, dones=None):
        pass

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

    def update_distribution(self, observations):
        mean = self.actor(observations)
        # validate_args=False: the argument check is a host sync (torch._is_all_true) per call and cannot be captured in a HIP graph;
        # the reference intends the same (`Normal.set_default_validate_args = False`, actor_critic.py:92)
        self.distribution = Normal(mean, mean * 0.0 + self.std, validate_args=False)

    def _noise(self, like):
        return torch.randn_like(like)

    def act(self, observations, **kwargs):
        self.update_distribution(observations)
        # mean + std * eps  ==  Normal(mean, std).sample() in law (actor_critic.py:123-125); written out because torch.normal
        # checks std >= 0 with a host sync, which a HIP-graph capture of the rollout does not allow
        d = self.distribution
        return (d.mean + d.stddev * self._noise(d.mean)).detach()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)
