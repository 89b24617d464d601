"""MI355X-native batched simulator of the two-wheel balance-robot environments (Env01 / Env03 families).

The hot path -- Env.step()/reset() of lachlanhurst/balance-robot-mujoco-rl, i.e. 250 MuJoCo substeps plus reward /
observation / termination / block state machine per env step -- runs as hand-written HIP kernels behind the C ABI in
include/brs.h (libbrs_hip.so).  This package is the host-side mirror of the reference's env interface:

    registry.make_vec(id, num_envs, ...)   ~ gym.make(id)  (src/balance_robot/__init__.py:5-52 of the reference)
    BalanceVecEnv                          ~ the SB3 VecEnv that sb_rl.py's PPO consumes (sb_rl.py:63-71, 552)
    BatchedSim                             zero-copy torch-tensor interface to the kernels
    policy.DevicePolicy / DeviceRollout    SB3 MlpPolicy forward + sampling, time-limit bootstrap and GAE as HIP kernels
                                           (include/brs_policy.h): the rollout side of sb_rl.py:63-71, 552-556 on the GPU

There is no CPU fallback: creating a sim without a HIP device raises.
"""
from .registry import ENV_SPECS, make_vec, spec  # noqa: F401
from .sim import BatchedSim, BrsError  # noqa: F401
from .vec_env import BalanceVecEnv  # noqa: F401

__all__ = ["BatchedSim", "BalanceVecEnv", "BrsError", "ENV_SPECS", "make_vec", "spec"]
