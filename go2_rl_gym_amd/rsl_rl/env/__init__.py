from .vec_env import VecEnv, missing_members
