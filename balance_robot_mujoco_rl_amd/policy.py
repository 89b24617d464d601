"""On-device rollout side of the path (include/brs_policy.h).

The reference hands its env to Stable-Baselines3's PPO("MlpPolicy") (src/sb_rl.py:63-71) and calls model.learn
(src/sb_rl.py:552-556); per env step SB3 runs the actor/critic forward, samples the diagonal Gaussian, clips the action,
steps the env, patches time-limit rewards and finally computes GAE -- all through numpy on the host.  `DevicePolicy` and
`DeviceRollout` do that arithmetic in HIP kernels of libbrs_hip.so on the simulator's own output tensors: a rollout of
65,536 envs never leaves the GPU and runs no per-env Python.  PyTorch only owns the buffers and the stream."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .sim import BrsError

# (name in an SB3 ActorCriticPolicy state_dict, shape) in the order of the flat parameter vector of brs_policy.h
SB3_LAYOUT = [("mlp_extractor.policy_net.0.weight", (64, 6)), ("mlp_extractor.policy_net.0.bias", (64,)),
              ("mlp_extractor.policy_net.2.weight", (64, 64)), ("mlp_extractor.policy_net.2.bias", (64,)),
              ("action_net.weight", (2, 64)), ("action_net.bias", (2,)),
              ("mlp_extractor.value_net.0.weight", (64, 6)), ("mlp_extractor.value_net.0.bias", (64,)),
              ("mlp_extractor.value_net.2.weight", (64, 64)), ("mlp_extractor.value_net.2.bias", (64,)),
              ("value_net.weight", (1, 64)), ("value_net.bias", (1,)), ("log_std", (2,))]
NPARAM = _lib.POLICY_NPARAM


def flatten_sb3_state_dict(sd):
    """SB3 MlpPolicy state_dict (tensors or arrays) -> flat float32 vector in brs_policy.h order"""
    parts = []
    for name, shape in SB3_LAYOUT:
        a = sd[name]
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        if tuple(a.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(a.shape)}")
        parts.append(a.astype(np.float32).ravel())
    flat = np.concatenate(parts)
    assert flat.size == NPARAM
    return flat


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need(t, name, dtype, shape, device):
    """the C ABI takes raw pointers: a sliced, float64 or host tensor would be read as garbage without any error"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device != device:
        raise ValueError(f"{name}: expected a tensor on {device}, got {getattr(t, 'device', type(t))}")
    if t.dtype != dtype or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected contiguous {dtype} of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)} "
                         f"(contiguous: {t.is_contiguous()})")
    return t


class DevicePolicy:
    """SB3's MlpPolicy (6-64-64 tanh actor and critic, state-independent log-std) evaluated by the HIP kernels"""

    def __init__(self, device=0, seed=0, env_index_base=0):
        if not torch.cuda.is_available():
            raise BrsError("no HIP device visible to PyTorch: the on-device policy has no CPU fallback")
        self.L = _lib.lib()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        h = C.c_void_p()
        rc = self.L.brs_policy_create(self.device.index, C.byref(h))
        if rc != 0:
            raise BrsError(f"brs_policy_create failed ({rc}): {self.L.brs_policy_last_error(None).decode()}")
        self.h = h
        self.seed, self.env_index_base = int(seed), int(env_index_base)
        self._dev_params = None

    def close(self):
        if getattr(self, "h", None):
            self.L.brs_policy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise BrsError(f"{what} failed ({rc}): {self.L.brs_policy_last_error(self.h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_weights(self, params):
        """flat float32 vector (host), or an SB3 state_dict"""
        flat = flatten_sb3_state_dict(params) if isinstance(params, dict) else np.ascontiguousarray(params, dtype=np.float32)
        if flat.size != NPARAM:
            raise ValueError(f"expected {NPARAM} parameters, got {flat.size}")
        self._check(self.L.brs_policy_set_weights(self.h, flat.ctypes.data_as(C.POINTER(C.c_float))), "brs_policy_set_weights")
        self._dev_params = None

    def use_device_weights(self, flat_param_tensor):
        """read the parameters from a device tensor the learner updates in place (no copies between optimiser steps)"""
        t = flat_param_tensor
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == NPARAM):
            raise ValueError("need a contiguous float32 CUDA tensor of NPARAM elements")
        self._dev_params = t  # keep it alive
        self._check(self.L.brs_policy_use_device_weights(self.h, _p(t)), "brs_policy_use_device_weights")

    def act(self, obs, step, deterministic=False, out=None, noise=None):
        """obs [n,6] f32 cuda -> (action [n,2] unclipped, action_clipped [n,2], logp [n], value [n]); `out` = the same four
        preallocated tensors (e.g. rows of a DeviceRollout)"""
        n = obs.shape[0]
        d, f32 = self.device, torch.float32
        _need(obs, "obs", f32, (n, 6), d)
        if out is None:
            out = (torch.empty((n, 2), dtype=f32, device=d), torch.empty((n, 2), dtype=f32, device=d),
                   torch.empty(n, dtype=f32, device=d), torch.empty(n, dtype=f32, device=d))
        a, ac, lp, v = out
        _need(a, "action", f32, (n, 2), d); _need(ac, "action_clipped", f32, (n, 2), d); _need(lp, "logp", f32, (n,), d); _need(v, "value", f32, (n,), d)
        if noise is not None:
            _need(noise, "noise", f32, (n, 2), d)
        self._check(self.L.brs_policy_act(self.h, n, _p(obs), self.seed, self.env_index_base, int(step) & 0xffffffff,
                                          int(bool(deterministic)), _p(a), _p(ac), _p(lp), _p(v), _p(noise), self._stream()),
                    "brs_policy_act")
        return out

    def value(self, obs, out=None):
        n = obs.shape[0]
        _need(obs, "obs", torch.float32, (n, 6), self.device)
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=self.device)
        _need(out, "value", torch.float32, (n,), self.device)
        self._check(self.L.brs_policy_value(self.h, n, _p(obs), _p(out), self._stream()), "brs_policy_value")
        return out

    def bootstrap(self, terminal_obs, terminated, truncated, gamma, reward):
        """reward += gamma * V(terminal_obs) where truncated and not terminated (in place)"""
        n, d = reward.shape[0], self.device
        _need(reward, "reward", torch.float32, (n,), d); _need(terminal_obs, "terminal_obs", torch.float32, (n, 6), d)
        _need(terminated, "terminated", torch.uint8, (n,), d); _need(truncated, "truncated", torch.uint8, (n,), d)
        self._check(self.L.brs_rollout_bootstrap(self.h, reward.shape[0], _p(terminal_obs), _p(terminated), _p(truncated),
                                                 float(gamma), _p(reward), self._stream()), "brs_rollout_bootstrap")
        return reward


def gae(reward, value, episode_start, last_value, last_done, gamma, lam, adv=None, ret=None):
    """GAE(lambda) on [T][N] device tensors (SB3 RolloutBuffer.compute_returns_and_advantage)"""
    T, N = reward.shape
    if adv is None:
        adv = torch.empty_like(reward)
    if ret is None:
        ret = torch.empty_like(reward)
    dev = reward.device
    rc = _lib.lib().brs_gae(dev.index, T, N, _p(reward), _p(value), _p(episode_start), _p(last_value), _p(last_done), float(gamma),
                            float(lam), _p(adv), _p(ret), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise BrsError(f"brs_gae failed ({rc})")
    return adv, ret


class DeviceRollout:
    """[T][N] rollout buffer resident in HBM (SB3's RolloutBuffer, without the host): collect() alternates
    brs_policy_act -> brs_step -> brs_rollout_bootstrap with no synchronisation and no allocation, finish() runs GAE."""

    def __init__(self, sim, policy, n_steps, gamma=0.99, gae_lambda=0.95):
        self.sim, self.policy, self.T, self.gamma, self.lam = sim, policy, int(n_steps), float(gamma), float(gae_lambda)
        n, d, T = sim.n, sim.device, self.T
        f = lambda *s: torch.zeros(s, dtype=torch.float32, device=d)
        self.obs, self.action, self.logp, self.value, self.reward = f(T, n, 6), f(T, n, 2), f(T, n), f(T, n), f(T, n)
        self.episode_start = torch.zeros((T, n), dtype=torch.uint8, device=d)
        self.adv, self.ret = f(T, n), f(T, n)
        self._clipped = f(n, 2)
        self._last_obs = None
        self._last_start = torch.ones(n, dtype=torch.uint8, device=d)
        self._last_value = f(n)
        self._step = 0

    def collect(self):
        sim, pol = self.sim, self.policy
        if self._last_obs is None:
            self._last_obs = sim.reset().clone()
        for t in range(self.T):
            self.obs[t].copy_(self._last_obs)
            self.episode_start[t].copy_(self._last_start)
            pol.act(self.obs[t], self._step, out=(self.action[t], self._clipped, self.logp[t], self.value[t]))
            self._step += 1
            obs, rew, term, trunc, tobs = sim.step(self._clipped)
            self.reward[t].copy_(rew)
            pol.bootstrap(tobs, term, trunc, self.gamma, self.reward[t])
            self._last_obs.copy_(obs)
            torch.bitwise_or(term, trunc, out=self._last_start)
        pol.value(self._last_obs, out=self._last_value)
        gae(self.reward, self.value, self.episode_start, self._last_value, self._last_start, self.gamma, self.lam, self.adv, self.ret)
        return self
