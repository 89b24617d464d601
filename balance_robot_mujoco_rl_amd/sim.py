"""BatchedSim: torch-tensor front end of the C ABI (include/brs.h).  PyTorch is plumbing here: it owns the device
buffers and the stream; all simulation work happens in libbrs_hip.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .registry import spec


class BrsError(RuntimeError):
    pass


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class BatchedSim:
    """N independent env instances of one variant on one GPU.

    step(actions) -> (obs[N,6] f32, reward[N] f32, terminated[N] bool, truncated[N] bool, terminal_obs[N,6] f32),
    all CUDA(=HIP) tensors owned by this object and overwritten by the next call (no allocation per step)."""

    def __init__(self, env_id, num_envs, device=0, seed=0, env_index_base=0, auto_reset=True, obs_noise=None,
                 max_episode_steps=0, substeps=0, timestep=0.0, block_threads=0, lane_grouping=True):
        self.spec = spec(env_id)
        if not torch.cuda.is_available():
            raise BrsError("no HIP device visible to PyTorch: the batched simulator has no CPU fallback")
        self.L = _lib.lib()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        flags = _lib.FLAG_AUTO_RESET if auto_reset else 0
        if obs_noise is True:
            flags |= _lib.FLAG_NOISE_ON
        elif obs_noise is False:
            flags |= _lib.FLAG_NOISE_OFF
        if not lane_grouping:
            flags |= _lib.FLAG_NO_LANE_GROUPING
        cfg = _lib.BrsConfig(self.spec.variant, int(num_envs), self.device.index, flags, int(seed), int(env_index_base),
                             int(max_episode_steps), int(substeps), float(timestep), int(block_threads), 0)
        h = C.c_void_p()
        rc = self.L.brs_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise BrsError(f"brs_create failed ({rc}): {self.L.brs_last_error(None).decode()}")
        self.h = h
        self.n = int(num_envs)
        nq, nv, no, na = (C.c_int32() for _ in range(4))
        self.L.brs_sizes(self.spec.variant, C.byref(nq), C.byref(nv), C.byref(no), C.byref(na))
        self.nq, self.nv = nq.value, nv.value
        self.max_episode_steps = max_episode_steps or self.spec.max_episode_steps
        d = self.device
        # step outputs live in ONE device allocation [obs 24n | terminal_obs 24n | reward 4n | terminated n | truncated n]
        # so that a host consumer (BalanceVecEnv) fetches them with a single D2H copy into a pinned mirror
        n = self.n
        self._packed = torch.zeros(54 * n, dtype=torch.uint8, device=d)
        self.obs = self._packed[0:24 * n].view(torch.float32).view(n, 6)
        self.terminal_obs = self._packed[24 * n:48 * n].view(torch.float32).view(n, 6)
        self.reward = self._packed[48 * n:52 * n].view(torch.float32)
        self.terminated = self._packed[52 * n:53 * n]
        self.truncated = self._packed[53 * n:54 * n]
        self._host = None  # pinned staging, allocated on first host-side use

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "h", None):
            self.L.brs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise BrsError(f"{what} failed ({rc}): {self.L.brs_last_error(self.h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ hot path
    def reset(self, mask=None):
        """reset all envs (or those where mask != 0); returns the obs tensor (rows of other envs untouched)"""
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mp = C.c_void_p(mask.data_ptr())
        self._check(self.L.brs_reset(self.h, mp, C.c_void_p(self.obs.data_ptr()), self._stream()), "brs_reset")
        return self.obs

    def step(self, actions):
        a = actions
        if not (isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()):
            a = torch.as_tensor(a, dtype=torch.float32, device=self.device).contiguous()
        if a.shape != (self.n, 2):
            raise ValueError(f"actions must have shape ({self.n}, 2), got {tuple(a.shape)}")
        self._check(self.L.brs_step(self.h, C.c_void_p(a.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                    C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.terminated.data_ptr()),
                                    C.c_void_p(self.truncated.data_ptr()), C.c_void_p(self.terminal_obs.data_ptr()),
                                    self._stream()), "brs_step")
        return self.obs, self.reward, self.terminated, self.truncated, self.terminal_obs

    # ------------------------------------------------------------------ host-side consumers (numpy in / numpy out)
    def _host_buffers(self):
        if self._host is None:
            n = self.n
            self._host = dict(out=torch.empty(54 * n, dtype=torch.uint8, pin_memory=True),
                              act=torch.empty((n, 2), dtype=torch.float32, pin_memory=True),
                              act_dev=torch.empty((n, 2), dtype=torch.float32, device=self.device),
                              done=torch.cuda.Event())
            o = self._host["out"].numpy()
            self._host["views"] = (o[0:24 * n].view(np.float32).reshape(n, 6), o[24 * n:48 * n].view(np.float32).reshape(n, 6),
                                   o[48 * n:52 * n].view(np.float32), o[52 * n:53 * n], o[53 * n:54 * n])
            self._host["act_np"] = self._host["act"].numpy()
        return self._host

    def step_host_async(self, actions_np):
        """numpy actions -> pinned staging -> async H2D, kernel, ONE async D2H of the packed outputs, all on the current
        stream; nothing blocks.  Pair with step_host_wait()."""
        hb = self._host_buffers()
        np.copyto(hb["act_np"], actions_np, casting="same_kind")
        hb["act_dev"].copy_(hb["act"], non_blocking=True)
        self.step(hb["act_dev"])
        hb["out"].copy_(self._packed, non_blocking=True)
        hb["done"].record(torch.cuda.current_stream(self.device))

    def step_host_wait(self):
        """-> (obs, terminal_obs, reward, terminated, truncated) numpy VIEWS of the pinned staging buffer (valid until the
        next step_host_async of this object)"""
        hb = self._host_buffers()
        hb["done"].synchronize()
        return hb["views"]

    def reset_host(self):
        hb = self._host_buffers()
        self.reset()
        hb["out"].copy_(self._packed, non_blocking=True)
        hb["done"].record(torch.cuda.current_stream(self.device))
        hb["done"].synchronize()
        return hb["views"][0]

    def physics(self, ctrl, nsub):
        c = torch.as_tensor(ctrl, dtype=torch.float32, device=self.device).contiguous()
        if c.shape != (self.n, 2):
            raise ValueError("ctrl must have shape (N, 2)")
        self._check(self.L.brs_physics(self.h, C.c_void_p(c.data_ptr()), int(nsub), self._stream()), "brs_physics")
        torch.cuda.current_stream(self.device).synchronize()  # keep `c` alive until the kernel has read it

    # ------------------------------------------------------------------ state access (host, synchronous)
    def get_state(self):
        qpos = np.zeros((self.n, self.nq)); qvel = np.zeros((self.n, self.nv))
        warm = np.zeros((self.n, self.nv)); time = np.zeros(self.n)
        self._check(self.L.brs_get_state(self.h, _dp(qpos), _dp(qvel), _dp(warm), _dp(time)), "brs_get_state")
        return qpos, qvel, warm, time

    def set_state(self, qpos=None, qvel=None, warm=None, time=None):
        c = lambda a, shape: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
        qpos, qvel = c(qpos, (self.n, self.nq)), c(qvel, (self.n, self.nv))
        warm, time = c(warm, (self.n, self.nv)), c(time, (self.n,))
        self._check(self.L.brs_set_state(self.h, _dp(qpos), _dp(qvel), _dp(warm), _dp(time)), "brs_set_state")

    def get_aux(self):
        aux = np.zeros((self.n, 14))
        self._check(self.L.brs_get_aux(self.h, _dp(aux)), "brs_get_aux")
        return aux

    def set_aux(self, aux):
        a = np.ascontiguousarray(np.asarray(aux, dtype=np.float64).reshape(self.n, 14))
        self._check(self.L.brs_set_aux(self.h, _dp(a)), "brs_set_aux")

    def get_xpose(self):
        xq = np.zeros((self.n, 4)); xp = np.zeros((self.n, 3))
        self._check(self.L.brs_get_xpose(self.h, _dp(xq), _dp(xp)), "brs_get_xpose")
        return xq, xp

    def set_xpose(self, xquat=None, xpos=None):
        c = lambda a, k: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(self.n, k))
        xquat, xpos = c(xquat, 4), c(xpos, 3)
        self._check(self.L.brs_set_xpose(self.h, _dp(xquat), _dp(xpos)), "brs_set_xpose")

    def step_bytes_per_env(self):
        return int(self.L.brs_step_bytes_per_env(self.h))

    def step_kernel_name(self):
        return self.L.brs_step_kernel_name(self.h).decode()
