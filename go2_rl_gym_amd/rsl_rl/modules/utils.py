"""Building blocks of the CTS / MoE-CTS networks (rsl_rl/rsl_rl/modules/utils.py:1-152): MLP, latent normalisers, the
mixture-of-experts student encoder.  Module / parameter names follow the reference so its checkpoints load unchanged
(`...moe.experts.backbone.network.0.weight`, `...moe.experts.experts.weight` [E*out, hidden, 1], ...).

The reference runs the E expert heads as a grouped 1x1 `nn.Conv1d` on a [B, E*hidden, 1] tensor (utils.py:78-92).  Here the
same parameters (same shapes, same default initialisation) drive ONE batched GEMM (`torch.baddbmm` -> hipBLASLt strided
batched) — on ROCm the grouped-convolution route goes through MIOpen's generic grouped kernels instead."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .actor_critic import get_activation
from .fused import FusedSequential, chain_heads as _chain_heads


class L2Norm(nn.Module):
    def forward(self, x):
        return F.normalize(x, p=2.0, dim=-1)


class SimNorm(nn.Module):
    """Simplicial normalisation: softmax over consecutive groups of 8 features (utils.py:31-45)."""

    def __init__(self):
        super().__init__()
        self.dim = 8

    def forward(self, x):
        shp = x.shape
        return F.softmax(x.view(*shp[:-1], -1, self.dim), dim=-1).view(*shp)

    def __repr__(self):
        return f"SimNorm(dim={self.dim})"


def make_norm(norm_type):
    assert norm_type in ("l2norm", "simnorm"), f"Normalization type {norm_type} not supported!"
    return L2Norm() if norm_type == "l2norm" else SimNorm()


class MLP(nn.Module):
    """dims[0] -> ... -> dims[-1]; activation between layers and, optionally, after the last one (utils.py:47-62)."""

    def __init__(self, dims, activation="elu", last_activation=False):
        super().__init__()
        layers = []
        for a, b in zip(dims[:-2], dims[1:-1]):
            layers += [nn.Linear(a, b), get_activation(activation)]
        layers.append(nn.Linear(dims[-2], dims[-1]))
        if last_activation:
            layers.append(get_activation(activation))
        self.network = FusedSequential(*layers)

    def forward(self, x):
        return self.network(x)


class GroupedHeads(nn.Module):
    """E independent linear heads, hidden -> out each, with the parameter layout of nn.Conv1d(E*hidden, E*out, 1, groups=E)."""

    def __init__(self, groups, in_per_group, out_per_group):
        super().__init__()
        self.groups, self.cin, self.cout = groups, in_per_group, out_per_group
        self.weight = nn.Parameter(torch.empty(groups * out_per_group, in_per_group, 1))
        self.bias = nn.Parameter(torch.empty(groups * out_per_group))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))          # nn.Conv1d.reset_parameters
        bound = 1.0 / math.sqrt(in_per_group)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):                                              # x [B, E*hidden] -> [B, E, out]
        return self._heads(x, True).transpose(0, 1)

    def _heads(self, x, with_bias: bool):                              # -> [E, B, out], the batched GEMM's own output layout
        B = x.shape[0]
        xe = x.view(B, self.groups, self.cin).transpose(0, 1)          # [E, B, hidden]
        w = self.weight.view(self.groups, self.cout, self.cin).transpose(1, 2)   # [E, hidden, out]
        if not with_bias:                                               # (the fused loss head of the student step adds the bias itself: a plain batched product here)
            if self.cout < 32 and x.is_cuda:                            # (the same narrow-output trap as below — a latent of 4 / 8 / 16 columns)
                return torch.bmm(xe, F.pad(w, (0, 32 - self.cout)))[..., :self.cout].contiguous()          # (dense [E, B, out]: what the fused heads read)
            return torch.bmm(xe, w)
        b = self.bias.view(self.groups, 1, self.cout)
        if self.cout < 32 and x.is_cuda:
            # narrow heads (12 actions, 1 value): the strided-batched GEMM with N < 32 runs ~100x slower on ROCm 7 (58 ms vs 0.5 ms
            # per fwd+bwd at 24576 rows, tools/probe_heads.py) — pad the output to 32 zero columns and slice them off again
            pad = 32 - self.cout
            return torch.baddbmm(F.pad(b, (0, pad)), xe, F.pad(w, (0, pad)))[..., :self.cout]
        return torch.baddbmm(b, xe, w)

    @torch.jit.ignore
    def expert_major(self, x, with_bias=True):
        """[E, B, out] — no transposing view in front of the consumer (the fused loss head of the MoE student step, modules/fused_cts.py:moe_head_grads)"""
        return self._heads(x, with_bias)


class Experts(nn.Module):
    def __init__(self, expert_num, input_dim, backbone_hidden_dims, expert_hidden_dim, output_dim, activation="elu"):
        super().__init__()
        self.expert_num, self.output_dim = expert_num, output_dim
        self.backbone = MLP([input_dim, *backbone_hidden_dims, expert_num * expert_hidden_dim], activation, last_activation=True)
        self.experts = GroupedHeads(expert_num, expert_hidden_dim, output_dim)

    def forward(self, x):
        return self.experts(self.backbone(x))                          # [B, E, out]


class MoE(nn.Module):
    """Dense (soft) mixture: softmax gate over E experts, output = sum_e w_e * expert_e(x) (utils.py:96-126)."""

    def __init__(self, expert_num, input_dim, hidden_dims, output_dim, activation="elu"):
        super().__init__()
        self.experts = Experts(expert_num, input_dim, hidden_dims[:-1], hidden_dims[-1], output_dim, activation)
        self.gating_network = nn.Sequential(MLP([input_dim, *hidden_dims[:-1], expert_num], activation), nn.Softmax(dim=-1))

    def forward(self, x):
        weights = self.gating_network(x)                               # [B, E]
        outs = self.experts(x)                                         # [B, E, out]
        return torch.sum(weights.unsqueeze(-1) * outs, dim=1), weights                   # [B, out]

    def parts(self, x):
        """-> (gate logits [B, E] — the gating MLP before its softmax —, expert outputs [E, B, out] before their bias, expert-major as the batched GEMM leaves them,
        the heads' bias parameter [E * out]): what the fused loss head of the student step mixes itself"""
        heads = self.experts.experts
        outs = _chain_heads(self.experts.backbone.network, heads, x)          # the backbone and its heads as ONE autograd node on the library's kernels where they are covered
        if outs is None:
            outs = heads.expert_major(self.experts.backbone(x), with_bias=False)
        return self.gating_network[0](x), outs, heads.bias          # outs [E, B, out] WITHOUT the bias


class StudentMoEEncoder(nn.Module):
    def __init__(self, expert_num, input_dim, hidden_dims, output_dim, activation="elu", norm_type="l2norm"):
        super().__init__()
        self.norm_layer = make_norm(norm_type)
        self.expert_num = expert_num
        self.moe = MoE(expert_num, input_dim, hidden_dims, output_dim, activation)

    def forward(self, obs):
        latent, weights = self.moe(obs)
        return self.norm_layer(latent), weights
