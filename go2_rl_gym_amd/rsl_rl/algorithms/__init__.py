from .ppo import PPO
