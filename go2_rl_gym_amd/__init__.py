"""go2_rl_gym_amd — MI355X-native vectorised Go2 environment + PPO rollout (hot path of wty-yy/go2_rl_gym)."""
__version__ = "0.1.0"
