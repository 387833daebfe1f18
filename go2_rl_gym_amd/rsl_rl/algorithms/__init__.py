from .ppo import PPO
from .cts import CTS
from .moe_cts import ACMoECTS, DualMoECTS, MCPCTS, MoECTS, MoENGCTS
