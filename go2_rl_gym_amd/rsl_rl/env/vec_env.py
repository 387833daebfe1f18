"""The environment contract of the runners (the reference states it as an abstract class, rsl_rl/rsl_rl/env/vec_env.py:36-59).

Here it is a runtime-checkable protocol plus `missing_members(env)`: OnPolicyRunner accepts any object that has the members below — the
go2sim-backed LeggedRobot, or a scripted stand-in as in tests/test_cts_golden.py — and says which ones are missing instead of failing
somewhere inside the first iteration.
"""
from typing import Dict, Optional, Protocol, Sequence, Tuple, runtime_checkable

import torch

StepResult = Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor, torch.Tensor, dict]

# attribute -> what the runner / the algorithms read it for (file:line of the reference's use)
REQUIRED_ATTRIBUTES: Dict[str, str] = {
    "num_envs": "rollout storage shape (on_policy_runner.py:88)",
    "num_obs": "actor input width (:73)",
    "num_privileged_obs": "critic input width, None = critic sees the actor's observations (:74-77)",
    "num_actions": "policy output width (:80)",
    "max_episode_length": "init_at_random_ep_len draws episode clocks below it (:118)",
    "episode_length_buf": "int64 [num_envs], the runner overwrites it once (:118)",
}
REQUIRED_METHODS: Sequence[str] = ("step", "reset", "get_observations", "get_privileged_observations")


@runtime_checkable
class VecEnv(Protocol):
    num_envs: int
    num_obs: int
    num_privileged_obs: Optional[int]
    num_actions: int
    max_episode_length: int
    episode_length_buf: torch.Tensor

    def step(self, actions: torch.Tensor) -> StepResult:
        """actions [num_envs, num_actions] -> (obs, privileged obs or None, rewards [num_envs], dones [num_envs], infos); infos carries
        'time_outs' (bool [num_envs], read at ppo.py:108) and 'episode' (means of the episodes that ended, on_policy_runner.py:145)."""

    def reset(self, env_ids=None):
        """-> (obs, privileged obs or None)"""

    def get_observations(self) -> torch.Tensor:
        """the env's own [num_envs, num_obs] buffer (no copy)"""

    def get_privileged_observations(self) -> Optional[torch.Tensor]:
        """the env's own [num_envs, num_privileged_obs] buffer, or None"""


def missing_members(env) -> list:
    """Names of the contract's members `env` lacks (empty list = usable by the runners)."""
    return [n for n in REQUIRED_ATTRIBUTES if not hasattr(env, n)] + [n for n in REQUIRED_METHODS if not callable(getattr(env, n, None))]
