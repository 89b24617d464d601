"""id -> variant table, mirroring the reference's gymnasium registrations (src/balance_robot/__init__.py:5-52).

Registered: the north-star ids Env01-v1/v2, Env03-v1/v2 and the f3 variants Env01-v3, Env02-v1."""
from dataclasses import dataclass


@dataclass(frozen=True)
class EnvSpec:
    id: str
    variant: int
    max_episode_steps: int
    reward_threshold: float
    obs_noise: bool  # reference behaviour: only Env01-v2 overrides get_pitch() with noise (envs/env01_v2.py:16-20)


ENV_SPECS = {
    "Env01-v1": EnvSpec("Env01-v1", 0, 6000, 6000, False),
    "Env01-v2": EnvSpec("Env01-v2", 1, 6000, 6000, True),
    "Env03-v1": EnvSpec("Env03-v1", 2, 6000, 6000, False),
    "Env03-v2": EnvSpec("Env03-v2", 3, 1200, 6000, False),
    "Env01-v3": EnvSpec("Env01-v3", 4, 6000, 6000, False),
    "Env02-v1": EnvSpec("Env02-v1", 5, 6000, 6000, False),
}


def spec(env_id: str) -> EnvSpec:
    try:
        return ENV_SPECS[env_id]
    except KeyError:
        raise KeyError(f"unknown environment id {env_id!r}; registered: {sorted(ENV_SPECS)}") from None


def make_vec(env_id: str, num_envs: int, **kwargs):
    """batched counterpart of gym.make(env_id): returns a BalanceVecEnv of `num_envs` instances"""
    from .vec_env import BalanceVecEnv
    return BalanceVecEnv(env_id, num_envs, **kwargs)
