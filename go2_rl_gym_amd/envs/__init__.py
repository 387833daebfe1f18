"""Task registration (legged_gym/envs/__init__.py).  `go2` keeps the reference's terrain default (trimesh curriculum,
scope row f1/f4); `go2_flat` is the BASELINE workload: the same robot on a plane."""
from ..utils.task_registry import task_registry
from .base.legged_robot import LeggedRobot  # noqa: F401
from .go2.go2_config import GO2Cfg, GO2CfgACMoECTS, GO2CfgDualMoECTS, GO2CfgMCPCTS, GO2CfgCTS, GO2CfgMoECTS, GO2CfgMoENGCTS, GO2CfgPPO, GO2FlatCfg, GO2FlatCfgCTS, GO2FlatCfgMoECTS, GO2FlatCfgPPO
from .go2.go2_env import Go2Robot

task_registry.register("go2", Go2Robot, GO2Cfg(), GO2CfgPPO())
task_registry.register("go2_flat", Go2Robot, GO2FlatCfg(), GO2FlatCfgPPO())
task_registry.register("go2_cts", Go2Robot, GO2Cfg(), GO2CfgCTS())                   # legged_gym/envs/__init__.py:10-11
task_registry.register("go2_moe_cts", Go2Robot, GO2Cfg(), GO2CfgMoECTS())
task_registry.register("go2_flat_cts", Go2Robot, GO2FlatCfg(), GO2FlatCfgCTS())      # same algorithms on the BASELINE (flat) terrain
task_registry.register("go2_flat_moe_cts", Go2Robot, GO2FlatCfg(), GO2FlatCfgMoECTS())
task_registry.register("go2_moe_ng_cts", Go2Robot, GO2Cfg(), GO2CfgMoENGCTS())       # legged_gym/envs/__init__.py:12
task_registry.register("go2_ac_moe_cts", Go2Robot, GO2Cfg(), GO2CfgACMoECTS())       # legged_gym/envs/__init__.py:14-15
task_registry.register("go2_dual_moe_cts", Go2Robot, GO2Cfg(), GO2CfgDualMoECTS())
task_registry.register("go2_mcp_cts", Go2Robot, GO2Cfg(), GO2CfgMCPCTS())             # legged_gym/envs/__init__.py:13
