"""rsl_rl-compatible PPO stack (modules, algorithms, storage, runners, env) of the hot path."""
