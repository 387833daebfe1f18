"""Deployment export of a trained policy (legged_gym/utils/exporter.py:13-338): TorchScript (the format
deploy/pre_train/go2/*.pt and RoboGauge consume), a plain state-dict pickle, and ONNX.

Same I/O contract as the reference's exported modules:
  PPO      forward(obs[1,45])  -> action[1,12]
  CTS      forward(obs[1,45])  -> (action, (None, latent[1,32]))             history of the last 5 observations kept inside
  MoE-CTS  forward(obs[1,45])  -> (action, (gate weights[1,E], latent))      the module; reset() clears it
ONNX modules take the 5-frame observation stack laid out BY TERM (all ang_vel frames, all gravity frames, ...), as the
reference's deployment code feeds it (:232-250), and return the action (MoE: action, weights, latent).

One small scriptable module per policy family instead of one class that rebinds `forward` at construction time."""
import copy
import os
from typing import Optional, Tuple

import torch
from torch import nn

_TERM_DIMS = (3, 3, 3, 12, 12, 12)        # ang_vel, gravity, commands, dof_pos, dof_vel, actions (legged_robot.py:270-277)


class _ActorPolicy(nn.Module):
    def __init__(self, actor, normalizer):
        super().__init__()
        self.actor, self.normalizer = actor, normalizer

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.actor(self.normalizer(x))

    @torch.jit.export
    def reset(self):
        pass


class _CTSPolicy(nn.Module):
    def __init__(self, actor, student_encoder, history_length: int, num_obs: int, normalizer):
        super().__init__()
        self.actor, self.student_encoder, self.normalizer = actor, student_encoder, normalizer
        self.history = torch.zeros(1, history_length, num_obs)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[Optional[torch.Tensor], torch.Tensor]]:
        x = self.normalizer(x)
        self.history = torch.cat([self.history[:, 1:], x.unsqueeze(1)], dim=1)
        latent = self.student_encoder(self.history.flatten(1))
        none: Optional[torch.Tensor] = None
        return self.actor(torch.cat([latent, x], dim=1)), (none, latent)

    @torch.jit.export
    def reset(self):
        self.history = torch.zeros_like(self.history)


class _MoECTSPolicy(nn.Module):
    def __init__(self, actor, student_moe_encoder, history_length: int, num_obs: int, normalizer):
        super().__init__()
        self.actor, self.student_moe_encoder, self.normalizer = actor, student_moe_encoder, normalizer
        self.history = torch.zeros(1, history_length, num_obs)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        x = self.normalizer(x)
        self.history = torch.cat([self.history[:, 1:], x.unsqueeze(1)], dim=1)
        latent, weights = self.student_moe_encoder(self.history.flatten(1))
        return self.actor(torch.cat([latent, x], dim=1)), (weights, latent)

    @torch.jit.export
    def reset(self):
        self.history = torch.zeros_like(self.history)


def _plain(m):
    """The training-time containers route Linear->ELU pairs through a fused backward (modules/fused.py); a deployment module is plain
    nn.Sequential again (same children, same parameter names) so TorchScript / ONNX see only stock modules."""
    from ..rsl_rl.modules.fused import FusedSequential
    for name, child in list(m.named_children()):
        m._modules[name] = _plain(child)
    return nn.Sequential(*m.children()) if isinstance(m, FusedSequential) else m


class _MoENGCTSPolicy(nn.Module):
    def __init__(self, actor, student_moe_encoder, mask, history_length: int, num_obs: int, normalizer):
        super().__init__()
        self.actor, self.student_moe_encoder, self.normalizer = actor, student_moe_encoder, normalizer
        self.history_length = history_length
        self.obs_no_goal_mask = mask
        self.history = torch.zeros(1, history_length, num_obs)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        x = self.normalizer(x)
        self.history = torch.cat([self.history[:, 1:], x.unsqueeze(1)], dim=1)
        no_goal = self.history.reshape(1, self.history_length, -1)[:, :, self.obs_no_goal_mask].reshape(1, -1)
        latent, weights = self.student_moe_encoder(self.history.flatten(1), no_goal)
        return self.actor(torch.cat([latent, x], dim=1)), (weights, latent)

    @torch.jit.export
    def reset(self):
        self.history = torch.zeros_like(self.history)


class _MCPCTSPolicy(nn.Module):
    """forward(obs) -> (action mean, (gate weights, latent))   (exporter.py:153-163)"""

    def __init__(self, actor_mcp, student_encoder, mask, history_length: int, num_obs: int, normalizer):
        super().__init__()
        self.actor, self.student_encoder, self.normalizer = actor_mcp, student_encoder, normalizer
        self.obs_no_goal_mask = mask
        self.history = torch.zeros(1, history_length, num_obs)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        x = self.normalizer(x)
        self.history = torch.cat([self.history[:, 1:], x.unsqueeze(1)], dim=1)
        latent = self.student_encoder(self.history.flatten(1))
        mean, std, weights = self.actor(torch.cat([latent, x], dim=1), torch.cat([latent, x[:, self.obs_no_goal_mask]], dim=1))
        return mean, (weights, latent)

    @torch.jit.export
    def reset(self):
        self.history = torch.zeros_like(self.history)


class _ACMoECTSPolicy(nn.Module):
    """forward(obs) -> (action, (actor gate weights, latent))   (exporter.py:165-171)"""

    def __init__(self, actor_moe, student_encoder, history_length: int, num_obs: int, normalizer):
        super().__init__()
        self.actor, self.student_encoder, self.normalizer = actor_moe, student_encoder, normalizer
        self.history = torch.zeros(1, history_length, num_obs)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        x = self.normalizer(x)
        self.history = torch.cat([self.history[:, 1:], x.unsqueeze(1)], dim=1)
        latent = self.student_encoder(self.history.flatten(1))
        mean, weights = self.actor(torch.cat([latent, x], dim=1))
        return mean, (weights, latent)

    @torch.jit.export
    def reset(self):
        self.history = torch.zeros_like(self.history)


class _DualMoECTSPolicy(nn.Module):
    """forward(obs) -> (action, (student gate weights, actor gate weights, latent))   (exporter.py:173-180)"""

    def __init__(self, actor_moe, student_moe_encoder, history_length: int, num_obs: int, normalizer):
        super().__init__()
        self.actor, self.student_moe_encoder, self.normalizer = actor_moe, student_moe_encoder, normalizer
        self.history = torch.zeros(1, history_length, num_obs)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        x = self.normalizer(x)
        self.history = torch.cat([self.history[:, 1:], x.unsqueeze(1)], dim=1)
        latent, sw = self.student_moe_encoder(self.history.flatten(1))
        mean, aw = self.actor(torch.cat([latent, x], dim=1))
        return mean, (sw, aw, latent)

    @torch.jit.export
    def reset(self):
        self.history = torch.zeros_like(self.history)


def _cpu_copy(m):
    return _plain(copy.deepcopy(m).cpu())


def _deployment_module(policy, normalizer=None):
    if getattr(policy, "is_recurrent", False):
        raise NotImplementedError("recurrent policies are not part of the go2 tasks")
    norm = _cpu_copy(normalizer) if normalizer else nn.Identity()
    if not hasattr(policy, "actor"):
        raise ValueError("Policy does not have an actor/student module.")
    if hasattr(policy, "actor_mcp"):
        return _MCPCTSPolicy(_cpu_copy(policy.actor_mcp), _cpu_copy(policy.student_encoder), policy.obs_no_goal_mask.detach().clone().cpu(),
                             policy.history.shape[1], policy.history.shape[2], norm)
    if hasattr(policy, "actor_moe"):
        am = _cpu_copy(policy.actor_moe)
        if hasattr(policy, "student_moe_encoder"):
            return _DualMoECTSPolicy(am, _cpu_copy(policy.student_moe_encoder), policy.history.shape[1], policy.history.shape[2], norm)
        return _ACMoECTSPolicy(am, _cpu_copy(policy.student_encoder), policy.history.shape[1], policy.history.shape[2], norm)
    actor = _cpu_copy(policy.actor)
    if hasattr(policy, "student_moe_encoder") and hasattr(policy, "obs_no_goal_mask"):
        return _MoENGCTSPolicy(actor, _cpu_copy(policy.student_moe_encoder), policy.obs_no_goal_mask.detach().clone().cpu(), policy.history.shape[1], policy.history.shape[2], norm)
    if hasattr(policy, "student_moe_encoder"):
        return _MoECTSPolicy(actor, _cpu_copy(policy.student_moe_encoder), policy.history.shape[1], policy.history.shape[2], norm)
    if hasattr(policy, "student_encoder"):
        return _CTSPolicy(actor, _cpu_copy(policy.student_encoder), policy.history.shape[1], policy.history.shape[2], norm)
    return _ActorPolicy(actor, norm)


def export_policy_as_jit(policy, path, normalizer=None, filename="policy.pt"):
    """TorchScript file with forward(obs) and reset() (exporter.py:13-23,186-191)."""
    os.makedirs(path, exist_ok=True)
    mod = _deployment_module(policy, normalizer).to("cpu")
    torch.jit.script(mod).save(os.path.join(path, filename))
    return os.path.join(path, filename)


def export_policy_as_pkl(policy, path, filename="policy.pkl"):
    """state_dict pickle (exporter.py:42-57)."""
    os.makedirs(path, exist_ok=True)
    torch.save(policy.state_dict(), os.path.join(path, filename))
    return os.path.join(path, filename)


class _OnnxPolicy(nn.Module):
    """Input: the frame stack grouped by observation term; output as the reference's ONNX graphs (exporter.py:232-287)."""

    def __init__(self, policy, normalizer=None):
        super().__init__()
        self.normalizer = _cpu_copy(normalizer) if normalizer else nn.Identity()
        self.actor = _cpu_copy(policy.actor_mcp if hasattr(policy, "actor_mcp") else (policy.actor_moe if hasattr(policy, "actor_moe") else policy.actor))
        self.actor_is_moe = hasattr(policy, "actor_moe")
        self.actor_is_mcp = hasattr(policy, "actor_mcp")
        self.kind = "moe" if hasattr(policy, "student_moe_encoder") else ("cts" if hasattr(policy, "student_encoder") else "ppo")
        self.encoder = _cpu_copy(policy.student_moe_encoder) if self.kind == "moe" else (_cpu_copy(policy.student_encoder) if self.kind == "cts" else None)
        obs_dim = sum(_TERM_DIMS)
        self.frames = policy.history.shape[1] if self.kind != "ppo" else 1
        self.input_dim = obs_dim * self.frames
        self.no_goal_mask = policy.obs_no_goal_mask.detach().clone().cpu() if hasattr(policy, "obs_no_goal_mask") else None

    def by_frame(self, x):
        obs_dim = sum(_TERM_DIMS)
        if x.shape[1] % obs_dim != 0:
            raise ValueError(f"x.shape[1] ({x.shape[1]}) is not a multiple of obs_dim ({obs_dim})")
        frames = x.shape[1] // obs_dim
        chunks = torch.split(x, [d * frames for d in _TERM_DIMS], dim=1)
        terms = [c.view(-1, frames, d) for c, d in zip(chunks, _TERM_DIMS)]
        return torch.cat([torch.cat([t[:, i, :] for t in terms], dim=1) for i in range(frames)], dim=1), obs_dim

    def forward(self, x):
        history, obs_dim = self.by_frame(self.normalizer(x))
        last = history[:, -obs_dim:]
        if self.kind == "ppo":
            return self.actor(last)
        if self.actor_is_mcp:                  # MCP: (mean, weights) from the full and the command-free input (:279-287)
            latent = self.encoder(history)
            mean, std, weights = self.actor(torch.cat([latent, last], dim=1), torch.cat([latent, last[:, self.no_goal_mask]], dim=1))
            return mean, weights
        if self.actor_is_moe:                  # AC-MoE / Dual-MoE: the actor returns (mean, gate weights)
            latent = self.encoder(history) if self.kind == "cts" else self.encoder(history)[0]
            return self.actor(torch.cat([latent, last], dim=1))[0]
        if self.kind == "cts":
            return self.actor(torch.cat([self.encoder(history), last], dim=1))
        if self.no_goal_mask is not None:      # MoE-NG: the experts read the history without the command entries (:268-277)
            no_goal = history.view(-1, self.frames, obs_dim)[:, :, self.no_goal_mask].reshape(x.shape[0], -1)
            latent, weights = self.encoder(history, no_goal)
        else:
            latent, weights = self.encoder(history)
        return self.actor(torch.cat([latent, last], dim=1)), weights, latent


def export_policy_as_onnx(policy, path, normalizer=None, filename="policy.onnx", verbose=False):
    """ONNX graph, opset 11, static shapes (exporter.py:25-40,289-307)."""
    os.makedirs(path, exist_ok=True)
    mod = _OnnxPolicy(policy, normalizer).to("cpu")
    # output names follow what forward() returns for the policy class (exporter.py:289-307): MCP (mean, weights); AC-MoE / Dual-MoE mean only;
    # MoE / MoE-NG (mean, weights, latent); PPO / CTS mean only
    if mod.kind != "ppo" and mod.actor_is_mcp:
        names = ["actions", "weights"]
    elif mod.kind != "ppo" and mod.actor_is_moe:
        names = ["actions"]
    else:
        names = ["actions"] + (["weights", "latent"] if mod.kind == "moe" else [])
    torch.onnx.export(mod, torch.zeros(1, mod.input_dim), os.path.join(path, filename), export_params=True, opset_version=11, verbose=verbose,
                      input_names=["obs"], output_names=names, dynamic_axes={}, dynamo=False)
    return os.path.join(path, filename)
